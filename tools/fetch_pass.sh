#!/bin/bash
# FETCH_SIZE PMC pass of the bench step under an environment (run on the GPU box from the repo root): tools/fetch_pass.sh TAG "ENV=.."
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export $2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-packed --no-recall --no-extra-configs --no-prof > $O/pf.log 2>&1
python3 $R/tools/pmc_stats.py $O/pf/p_counter_collection.csv $O/pmc_fetch.txt > /dev/null
rm -rf $O/pf
grep "gemm_nt2_kernel" $O/pmc_fetch.txt | cut -c1-170
