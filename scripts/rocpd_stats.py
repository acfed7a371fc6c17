"""Kernel statistics from a rocprofv3 rocpd sqlite database (the --stats view).

   python scripts/rocpd_stats.py <results.db> [out.md|-] [skip_first_n_dispatches]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[skip:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"igemm_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\d+)>", n)
    if m:
        return f"igemm_kernel<{m.group(1)}x{m.group(2)},A{m.group(3)},B{m.group(4)}>"
    return n[:90]


agg = {}
for name, s, e in rows:
    k = short(name)
    a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
    d = (e - s) / 1e3
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
span = (rows[-1][2] - rows[0][1]) / 1e3
lines = [f"kernel dispatches: {len(rows)}; summed kernel time {total/1e3:.3f} ms; "
         f"first-start..last-end span {span/1e3:.3f} ms", "",
         "| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"| {k} | {a[0]} | {a[1]:.1f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | "
                 f"{100*a[1]/total:.1f} |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    open(sys.argv[2], "w").write(txt + "\n")
