#!/bin/bash
# LDS bank-conflict ratio of every kernel of the step: one SQ PMC pass over 3 bench steps.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ldsc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-packed --no-prof > $O/log.txt 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/p/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:16]:
    a, c = d.get("SQ_LDS_IDX_ACTIVE", 0), d.get("SQ_LDS_BANK_CONFLICT", 0)
    w, wa = d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_WAIT_ANY", 0)
    print(f"{k:70s} lds_active {a:12.4e} conflict {c:12.4e} ratio {c/a if a else 0:6.3f}  wait_any/wave {wa/w if w else 0:5.2f}")
PY
