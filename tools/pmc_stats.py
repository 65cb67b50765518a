"""Per-kernel averages of a rocprofv3 --pmc run (CSV output, *_counter_collection.csv)."""
import csv, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"((?:[\w:]|<[^<>]*(?:<[^<>]*>[^<>]*)*>)+)", name)
    return (m.group(1) if m else name)[:100]


def main(path, out=None):
    csv.field_size_limit(1 << 30)
    acc = defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    lines = [f"{'kernel':<100} {'counter':<14} {'dispatches':>10} {'mean/dispatch':>16} {'total':>16}"]
    for (k, c), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:<100} {c:<14} {n:>10} {tot / n:>16.1f} {tot:>16.1f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
