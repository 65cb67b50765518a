R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/jpegkt -o kt -- python3 $R/tools/jpeg_bench.py 256 > $R/gpurun_out/jpegkt.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/jpegkt/*.db | head -1) $R/gpurun_out/jpeg_kernel_stats.txt --from-first jpeg_huffman > /dev/null
rm -rf $R/gpurun_out/jpegkt
cat $R/gpurun_out/jpegkt.log | grep -v amdgpu; head -8 $R/gpurun_out/jpeg_kernel_stats.txt | cut -c1-50,90-150
