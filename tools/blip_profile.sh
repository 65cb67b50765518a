# rocprofv3 kernel table of the BLIP4CIR stage-2 step (tools/blip_bench.py; BLIP_ARGS="--dense" for the padded rows); step-only
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${BLIP_OUT:-blip_kernel_stats.txt}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/blipkt -o kt -- python3 $R/tools/blip_bench.py --images 2000 --steps 5 $BLIP_ARGS > $R/gpurun_out/blipkt.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/blipkt/*.db | head -1) $R/gpurun_out/$OUT --from-first gather_bank_rows_bf16_kernel > /dev/null
rm -rf $R/gpurun_out/blipkt
tail -1 $R/gpurun_out/blipkt.log | cut -c1-40; head -${BLIP_HEAD:-24} $R/gpurun_out/$OUT | cut -c1-60,90-150; tail -1 $R/gpurun_out/$OUT
