"""Idle time between consecutive kernel dispatches in a rocprofv3 kernel-trace database (rocpd sqlite)."""
import re, sqlite3, sys
from collections import defaultdict

def main(db, skip_ms=0.0):
    c = sqlite3.connect(db)
    rows = list(c.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
                             join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
    # keep the second half of the run (steady-state steps)
    rows = rows[len(rows) // 2:]
    span = rows[-1][2] - rows[0][1]
    busy = sum(e - s for _, s, e in rows)
    gaps = defaultdict(lambda: [0, 0])
    tot_gap = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g > 0:
            k = (re.sub(r"\(.*", "", n0)[:48], re.sub(r"\(.*", "", n1)[:48])
            gaps[k][0] += 1; gaps[k][1] += g
            tot_gap += g
    print(f"dispatches {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  positive gaps {tot_gap/1e6:.2f} ms ({100*tot_gap/span:.1f} %)")
    for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"{t/1e3:9.1f} us total  {t/n/1e3:6.2f} us avg x{n:5d}   {a}  ->  {b}")

if __name__ == "__main__":
    main(sys.argv[1])
