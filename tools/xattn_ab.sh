# Per-kernel times of the absorbed cross-attention at config 4's shape, second generation (csrc/xattn2.hip) against the first
# (SPN_XATTN_V2=0): one rocprofv3 kernel-trace run per variant and shape.   usage: tools/xattn_ab.sh ["B L" ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- "128 32"
for cfg in "$@"; do
  for v2 in 1 0; do
    export SPN_XATTN_V2=$v2
    rm -rf $R/gpurun_out/xb
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/xb -o kt -- python3 $R/tools/xattn_bench.py $cfg > $R/gpurun_out/xb.log 2>&1
    echo "== B L = $cfg   SPN_XATTN_V2=$v2"
    python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/xb/*.db | head -1) $R/gpurun_out/xb_stats_v$v2.txt --from-first xattn_head_expand_kernel > /dev/null
    grep -E "xattn" $R/gpurun_out/xb_stats_v$v2.txt | cut -c1-60,90-140
  done
done
rm -rf $R/gpurun_out/xb
