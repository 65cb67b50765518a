R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/blipkt -o kt -- python3 $R/tools/blip_bench.py --images 2000 --steps 5 > $R/gpurun_out/blipkt.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/blipkt/*.db | head -1) $R/gpurun_out/blip_kernel_stats.txt --from-first fusion_mask_kernel > /dev/null
rm -rf $R/gpurun_out/blipkt
head -22 $R/gpurun_out/blip_kernel_stats.txt | cut -c1-60,90-150; tail -1 $R/gpurun_out/blip_kernel_stats.txt
