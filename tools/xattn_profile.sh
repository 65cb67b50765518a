# Per-kernel times of the absorbed cross-attention (tools/xattn_bench.py) across (B, L): one rocprofv3 kernel-trace run per shape.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "128 32" "64 32" "32 32" "128 16" "128 8" "128 1" "256 32"; do
  rm -rf $R/gpurun_out/xb
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/xb -o kt -- python3 $R/tools/xattn_bench.py $cfg > $R/gpurun_out/xb.log 2>&1
  echo "== B L = $cfg"
  python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/xb/*.db | head -1) $R/gpurun_out/xb_stats.txt --from-first xattn_head_expand_kernel > /dev/null
  grep -E "xattn" $R/gpurun_out/xb_stats.txt | cut -c1-50,90-140
done
rm -rf $R/gpurun_out/xb
