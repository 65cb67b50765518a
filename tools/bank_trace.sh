#!/bin/bash
# rocprofv3 kernel tables of the bank InfoNCE pair per case -> gpurun_out/bank_trace.txt (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/bank_trace.txt
: > $OUT
for c in "32 40000 768 bf16" "256 40000 768 bf16" "32 100000 768 bf16" "32 100000 768 fp8" "16 400000 768 bf16" "16 400000 768 fp8"; do
  rm -rf $R/gpurun_out/bt
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bt -o kt -- python3 $R/tools/bank_trace.py $c > $R/gpurun_out/bt.log 2>&1
  echo "== B M D bank = $c  (28 forward/backward pairs, 4 rotating bank copies)" >> $OUT
  python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/bt/*.db | head -1) $R/gpurun_out/bt_stats.txt > /dev/null
  grep -E "kernel  |bank_|gemm_|splitk|fold" $R/gpurun_out/bt_stats.txt | cut -c1-70,90-150 >> $OUT
done
rm -rf $R/gpurun_out/bt
cat $OUT
