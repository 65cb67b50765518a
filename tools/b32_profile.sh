# rocprofv3 kernel table of the dense config-2 step at 32 triplets per GPU (config 3's strong-scaling split: 2 464 rows), step-only
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/b32kt -o kt -- python3 $R/bench.py --batch-per-gpu 32 --steps 40 --warmup 5 --no-cpu-baseline --no-packed --no-recall --no-extra-configs --no-prof > $R/gpurun_out/b32kt.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/b32kt/*.db | head -1) $R/gpurun_out/r06_b32_kernel_stats.txt --from-first eot_argmax_kernel > /dev/null
rm -rf $R/gpurun_out/b32kt
tail -1 $R/gpurun_out/b32kt.log | cut -c1-200; head -32 $R/gpurun_out/r06_b32_kernel_stats.txt | cut -c1-64,92-150; tail -1 $R/gpurun_out/r06_b32_kernel_stats.txt
