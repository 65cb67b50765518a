#!/bin/bash
# A/B kernel-trace of the bench step under two environments (run on the GPU box from the repo root):
#   tools/ab_profile.sh TAG "ENV_A" "ENV_B"    e.g.  tools/ab_profile.sh fuse "SPN_FUSE_RESID=1" "SPN_FUSE_RESID=0"
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for E in "$2" "$3"; do
  i=$((i+1))
  [ -z "$E" ] && [ $i -gt 1 ] && continue
  export $E
  rocprofv3 --kernel-trace --stats -d $O/kt$i -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-packed --no-recall --no-extra-configs > $O/kt$i.log 2>&1
  python3 $R/tools/rocpd_stats.py $(ls $O/kt$i/*.db | head -1) $O/kernel_stats_$i.txt > /dev/null
  grep '^{"metric' $O/kt$i.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$E', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  head -16 $O/kernel_stats_$i.txt | cut -c1-40,90-160
  unset ${E%%=*}
  rm -rf $O/kt$i
done
