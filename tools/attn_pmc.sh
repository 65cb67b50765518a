#!/bin/bash
# SQ PMC passes over tools/attn_bench.py (whole-head attention at the text tower's shape): where the waves' cycles go.
#   tools/attn_pmc.sh TAG [ENV=..]     run on the GPU box from the repo root
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/attnpmc_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export $2
[ -n "$3" ] && export $3
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/attn_bench.py > $O/log$i.txt 2>&1
  python3 $R/tools/pmc_stats.py $O/p$i/p_counter_collection.csv $O/pmc$i.txt > /dev/null
  rm -rf $O/p$i
done
cat $O/pmc1.txt $O/pmc2.txt | grep "attention_small_bwd" | cut -c1-60,100-170
