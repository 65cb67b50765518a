R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tgkt -o kt -- python3 $R/tools/tgcir_bench.py 256 > $R/gpurun_out/tgkt.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/tgkt/*.db | head -1) $R/gpurun_out/tgcir_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/tgkt
head -24 $R/gpurun_out/tgcir_kernel_stats.txt | cut -c1-70,90-150; tail -1 $R/gpurun_out/tgcir_kernel_stats.txt
