"""Kernel breakdown of ONE optimizer step out of a rocprofv3 --kernel-trace database: the launches
between the last two optimizer (multi_tensor / Adam) launch groups.

    python scripts/rocpd_one_step.py <results.db>
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if "multi_tensor" in n or "adam" in n.lower()]
ends = []
for i in idx:
    if not ends or i - ends[-1] > 50:
        ends.append(i)
    else:
        ends[-1] = i
print("step delimiters", len(ends))
a, b = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
seg = rows[a:b]
span = (seg[-1][2] - seg[0][1]) / 1e3
busy = sum(e - s for _, s, e in seg) / 1e3
print(f"one update: {len(seg)} launches, span {span:.1f} us, summed kernel time {busy:.1f} us")
agg = {}
for n, s, e in seg:
    k = re.sub(r"\(anonymous namespace\)::", "", n)[:80]
    v = agg.setdefault(k, [0, 0.0])
    v[0] += 1
    v[1] += (e - s) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[0]:6d} {v[1]:10.1f} us  {k}")
gap = 0.0
last = seg[0][2]
for n, s, e in seg[1:]:
    if s > last:
        gap += (s - last) / 1e3
    last = max(last, e)
print(f"idle time between launches: {gap:.1f} us")
