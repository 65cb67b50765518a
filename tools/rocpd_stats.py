"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table."""
import sqlite3, sys, re

def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    lines = [f"{'kernel':<90} {'calls':>6} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}"]
    for name, n, tot, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)[:90]
        lines.append(f"{short:<90} {n:>6} {tot/1e6:>10.3f} {avg/1e3:>9.1f} {mn/1e3:>9.1f} {mx/1e3:>9.1f} {100*tot/total:>6.2f}")
    lines.append(f"TOTAL kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
