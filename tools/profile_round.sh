#!/bin/bash
# Round profile bundle (run on the GPU box from the repo root): bench JSON, rocprofv3 kernel stats of the same
# command, and the two HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-packed --no-recall --no-extra-configs > $O/kt.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $O/kt/*.db | head -1) $O/kernel_stats.txt --from-first eot_argmax_kernel > /dev/null
python3 $R/tools/rocpd_stats.py $(ls $O/kt/*.db | head -1) $O/kernel_stats_whole_process.txt > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-packed --no-recall --no-extra-configs --no-prof > $O/pf.log 2>&1
python3 $R/tools/pmc_stats.py $O/pf/p_counter_collection.csv $O/pmc_fetch.txt > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-packed --no-recall --no-extra-configs --no-prof > $O/pw.log 2>&1
python3 $R/tools/pmc_stats.py $O/pw/p_counter_collection.csv $O/pmc_write.txt > /dev/null
rm -rf $O/pf $O/pw $O/kt
tail -c 600 $O/bench_n1.json; head -30 $O/kernel_stats.txt; head -12 $O/pmc_fetch.txt | cut -c1-170; head -12 $O/pmc_write.txt | cut -c1-170
