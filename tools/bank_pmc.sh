#!/bin/bash
# HBM-side traffic of the bank pair (PMC FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes;
# FETCH_SIZE x2 on gfx950) -> gpurun_out/bank_pmc.txt.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/bank_pmc.txt
: > $OUT
for c in "32 40000 768 bf16" "16 400000 768 bf16" "16 400000 768 fp8" "256 40000 768 bf16"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/bp
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/bp -o p -- python3 $R/tools/bank_trace.py $c > /tmp/bp.log 2>&1
    echo "== B M D bank = $c   $ctr (KB per dispatch)" >> $OUT
    python3 $R/tools/pmc_stats.py /tmp/bp/p_counter_collection.csv /tmp/bp_stats.txt > /dev/null
    grep -E "bank_|gemm_|splitk" /tmp/bp_stats.txt | cut -c1-60,100-170 >> $OUT
  done
done
cat $OUT
