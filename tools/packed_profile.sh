R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pk -o kt -- python3 $R/tools/packed_step.py 10 > $R/gpurun_out/pk.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/pk/*.db | head -1) $R/gpurun_out/packed_kernel_stats.txt --from-first build_row_map > /dev/null
rm -rf $R/gpurun_out/pk
tail -1 $R/gpurun_out/pk.log; head -24 $R/gpurun_out/packed_kernel_stats.txt | cut -c1-52,90-150; tail -1 $R/gpurun_out/packed_kernel_stats.txt
