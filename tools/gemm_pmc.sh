#!/bin/bash
# SQ-block PMC pass (one pass, no trace domains) of the default NT / TN GEMM kernels on a step shape:
# MFMA pipe busy cycles, LDS bank conflicts, wave-cycle split.   bash tools/gemm_pmc.sh  (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gemm_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kind in nt tn tng; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --kernel-trace --output-format csv -d $O/$kind -o p -- python3 $R/tools/gemm_only.py 19712 3072 768 $kind > $O/$kind.log 2>&1
  python3 $R/tools/pmc_stats.py $O/$kind/p_counter_collection.csv $O/pmc_$kind.txt > /dev/null
  grep -E "gemm_(nt|tn)2_(group_)?kernel" $O/pmc_$kind.txt | cut -c1-60,100-170
done
