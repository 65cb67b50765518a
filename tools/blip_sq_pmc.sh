#!/bin/bash
# SQ PMC pass over the BLIP step (tools/blip_bench.py $BLIP_ARGS): per-kernel wave cycles, wait / active split, MFMA-busy and LDS counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/blipsq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES"
C2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES"
rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $R/tools/blip_bench.py --images 2000 --steps 2 --warmup 1 $BLIP_ARGS > $O/log1.txt 2>&1
rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/p2 -o p -- python3 $R/tools/blip_bench.py --images 2000 --steps 2 --warmup 1 $BLIP_ARGS > $O/log2.txt 2>&1
python3 - <<PY
import csv, collections, re
csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in ("$O/p1/p_counter_collection.csv", "$O/p2/p_counter_collection.csv"):
    try:
        rd = csv.DictReader(open(f))
    except OSError:
        continue
    first = f.endswith("p1/p_counter_collection.csv")
    for r in rd:
        k = re.sub(r"^void ", "", r["Kernel_Name"])[:60]
        c = r["Counter_Name"]
        if not first and c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"):
            c += "_2"
        agg[k][c] += float(r["Counter_Value"])
        if c == "SQ_WAVES":
            n[k] += 1
rows = []
for k, d in agg.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0:
        continue
    busy = d.get("SQ_BUSY_CYCLES", 0) / 32.0
    wc2, busy2 = max(d.get("SQ_WAVE_CYCLES_2", 0), 1), max(d.get("SQ_BUSY_CYCLES_2", 0) / 32.0, 1)
    rows.append((wc, k, n[k], busy / max(n[k], 1), 4 * wc / max(busy, 1) / 256.0, d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc,
                 d.get("SQ_ACTIVE_INST_ANY", 0) / wc, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(busy, 1) / 1024.0,
                 d.get("SQ_WAIT_INST_LDS", 0) / wc2, d.get("SQ_ACTIVE_INST_LDS", 0) / wc2,
                 d.get("SQ_LDS_BANK_CONFLICT", 0) / busy2 / 256.0, d.get("SQ_INSTS_LDS", 0)))
out = ["%-60s %6s %9s %8s %5s %5s %6s %9s %8s %7s %9s" % ("kernel", "calls", "cyc/call", "waves/CU", "wait", "stall", "active", "mfma_busy", "wait_lds", "act_lds", "bank_conf")]
for wc, k, c, cyc, occ, w, s, a, m, wl, al, bc, il in sorted(rows, reverse=True)[:18]:
    out.append("%-60s %6d %9.0f %8.1f %5.2f %5.2f %6.2f %9.2f %8.2f %7.2f %9.2f" % (k, c, cyc, occ, w, s, a, m, wl, al, bc))
open("$R/gpurun_out/blip_sq_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -3 $O/log2.txt | cut -c1-300
rm -rf $O/p1 $O/p2
