#!/bin/bash
# SQ PMC pass over 3 bench steps: per-kernel wave cycles, wait / active split and MFMA-busy cycles.   tools/step_sq_pmc.sh TAG [ENV=..]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sq_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export $2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-packed --no-recall --no-extra-configs --no-prof > $O/log.txt 2>&1
python3 - <<PY
import csv, collections, re
csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open("$O/p/p_counter_collection.csv")):
    k = re.sub(r"^void ", "", r["Kernel_Name"])[:86]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
rows = []
for k, d in agg.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0:
        continue
    busy = d.get("SQ_BUSY_CYCLES", 0) / 32.0                      # summed over 32 shader engines -> kernel cycles
    rows.append((wc, k, n[k], busy / max(n[k], 1), 4 * wc / max(busy, 1) / 256.0, d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc,
                 d.get("SQ_ACTIVE_INST_ANY", 0) / wc, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(busy, 1) / 1024.0))
out = ["%-86s %6s %10s %8s %6s %6s %6s %8s" % ("kernel", "calls", "cyc/call", "waves/CU", "wait", "stall", "active", "mfma_busy")]
for wc, k, c, cyc, occ, w, s, a, m in sorted(rows, reverse=True)[:24]:
    out.append("%-86s %6d %10.0f %8.1f %6.2f %6.2f %6.2f %8.2f" % (k, c, cyc, occ, w, s, a, m))
open("$O/sq_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $O/p
