"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table.
    python tools/rocpd_stats.py trace.db [out.txt] [--from-first KERNEL_SUBSTRING]
--from-first: count only the dispatches from the first launch of a kernel whose name contains the substring on - a bench's set-up
(random fills of synthetic banks, uploads) then stays out of the table of the STEP; the header line says so."""
import sqlite3, sys, re

def main(db, out=None, from_first=None):
    c = sqlite3.connect(db)
    where, note = "", ""
    if from_first:
        r = c.execute("""select min(d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                         where s.kernel_name like ?""", (f"%{from_first}%",)).fetchone()
        if r and r[0] is not None:
            where = f"where d.start >= {int(r[0])}"
            note = f"  (dispatches from the first `{from_first}` launch on: the bench's set-up is excluded)"
    q = f"""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id {where}
           group by s.kernel_name order by 3 desc"""
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    lines = [f"{'kernel':<90} {'calls':>6} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}"]
    for name, n, tot, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)[:90]
        lines.append(f"{short:<90} {n:>6} {tot/1e6:>10.3f} {avg/1e3:>9.1f} {mn/1e3:>9.1f} {mx/1e3:>9.1f} {100*tot/total:>6.2f}")
    lines.append(f"TOTAL kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches{note}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")

if __name__ == "__main__":
    args = sys.argv[1:]
    ff = None
    if "--from-first" in args:
        i = args.index("--from-first")
        ff = args[i + 1]
        del args[i:i + 2]
    main(args[0], args[1] if len(args) > 1 else None, ff)
