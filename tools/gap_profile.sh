R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/gaps -o g -- python3 $R/bench.py --no-cpu-baseline --no-packed --no-prof --steps 10 --warmup 3 > $R/gpurun_out/gaps.log 2>&1
python3 $R/tools/gap_stats.py $(ls $R/gpurun_out/gaps/*.db | head -1)
rm -rf $R/gpurun_out/gaps
