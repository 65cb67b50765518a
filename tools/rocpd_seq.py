"""Per-dispatch view of a rocprofv3 rocpd database: GEMM dispatches of the LAST bench step in launch order, and per
(kernel, grid) statistics - separates the shapes that share one kernel instantiation."""
import re, sqlite3, sys
from collections import defaultdict

def main(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gcol = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    wcol = "workgroup_size_x" if "workgroup_size_x" in cols else None
    sel = f"d.{gcol}" if gcol else "0"
    sel2 = f"d.{wcol}" if wcol else "1"
    rows = list(c.execute(f"""select s.kernel_name, {sel}, {sel2}, d.start, d.end from rocpd_kernel_dispatch d
                              join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
    def short(n):
        m = re.match(r".*?(gemm_\w+?)_kernel<(.*?)>", n)
        if m:
            return f"{m.group(1)}<{m.group(2).replace(' ', '')}>"
        return re.sub(r"\(.*", "", n)[:40]
    stats = defaultdict(list)
    for n, g, w, s, e in rows:
        if "gemm" in n:
            stats[(short(n), g // max(w, 1))].append((e - s) / 1e3)
    print(f"{'kernel':<60} {'blocks':>7} {'calls':>6} {'avg_us':>8} {'min_us':>8} {'max_us':>8}")
    for (n, g), v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:<60} {g:>7} {len(v):>6} {sum(v)/len(v):>8.1f} {min(v):>8.1f} {max(v):>8.1f}")
    # the last 98 + 37 gemm dispatches = one step: print in order
    gem = [(short(n), g // max(w, 1), (e - s) / 1e3) for n, g, w, s, e in rows if "gemm_nt" in n]
    print("\nNT GEMMs of the last layer-forward + layer-backward (launch order):")
    for n, g, us in gem[-98:-98 + 4] + gem[-8:]:
        print(f"  {n:<60} blocks {g:>5}  {us:7.1f} us")

if __name__ == "__main__":
    main(sys.argv[1])
