#!/bin/bash
# PMC passes over tools/xattn_bench.py (args: "B L"): SQ wait/active/MFMA-busy, LDS counters, HBM-side fetch / write bytes, L2 hit rate.
# Each counter group is its own rocprofv3 run (counters only with --kernel-trace, MI355X_MICROARCH.md).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/xpmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFG=${1:-128 32}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/xattn_bench.py $CFG > $O/log$i.txt 2>&1
done
python3 - <<PY
import csv, collections, re, glob
csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
for i in range(1, 7):
    for f in glob.glob("$O/p%d/*counter_collection.csv" % i):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"])[:58]
            if "xattn" not in k:
                continue
            c = r["Counter_Name"] + ("_2" if i == 2 and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES") else "")
            agg[k][c] += float(r["Counter_Value"])
            n[k][c] += 1
out = ["%-58s %5s %9s %8s %5s %5s %6s %9s %8s %7s %9s %9s %9s %6s %8s" % ("kernel", "calls", "cyc/call", "waves/CU", "wait", "stall", "active", "mfma_busy", "wait_lds", "act_lds", "bank_conf", "fetchMB*2", "writeMB", "L2hit", "gui_cyc")]
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = max(d.get("SQ_WAVE_CYCLES", 0), 1)
    calls = max(n[k]["SQ_WAVES"], 1)
    busy = max(d.get("SQ_BUSY_CYCLES", 0) / 32.0, 1)
    wc2, busy2 = max(d.get("SQ_WAVE_CYCLES_2", 0), 1), max(d.get("SQ_BUSY_CYCLES_2", 0) / 32.0, 1)
    hit, miss = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    out.append("%-58s %5d %9.0f %8.1f %5.2f %5.2f %6.2f %9.2f %8.2f %7.2f %9.2f %9.1f %9.1f %6.2f %8.0f" % (
        k, calls, busy / calls, 4 * wc / busy / 256.0, d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy / 1024.0, d.get("SQ_WAIT_INST_LDS", 0) / wc2, d.get("SQ_ACTIVE_INST_LDS", 0) / wc2,
        d.get("SQ_LDS_BANK_CONFLICT", 0) / busy2 / 256.0, 2 * d.get("FETCH_SIZE", 0) / max(n[k]["FETCH_SIZE"], 1) / 1024.0, d.get("WRITE_SIZE", 0) / max(n[k]["WRITE_SIZE"], 1) / 1024.0,
        hit / max(hit + miss, 1), d.get("GRBM_GUI_ACTIVE", 0) / max(n[k]["GRBM_GUI_ACTIVE"], 1)))
open("$R/gpurun_out/xattn_pmc_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -2 $O/log1.txt | cut -c1-200
rm -rf $O
