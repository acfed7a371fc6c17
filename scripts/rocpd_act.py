"""Per-call view of scripts/act_profile.py under rocprofv3 --kernel-trace: launches, kernel time
per stream and the idle time of the LAST call (calls are delimited by the one-element
`marker.add_` launch the script issues before each).

    python scripts/rocpd_act.py <results.db> [list]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
marks = [i for i, r in enumerate(rows)
         if "CUDAFunctorOnSelf_add" in r[0] or ("vectorized_elementwise" in r[0] and "add" in r[0].lower()
                                                and r[2] - r[1] < 4000)]
# the delimiter is the add_ right before a call: keep marks that are followed by > 50 launches
marks = [m for k, m in enumerate(marks) if (marks[k + 1] if k + 1 < len(marks) else len(rows)) - m > 50]
a = marks[-2] if len(marks) >= 2 else 0
b = marks[-1] if len(marks) >= 2 else len(rows)
seg = rows[a + 1:b]
t0, t1 = seg[0][1], max(r[2] for r in seg)
print(f"one act(): {len(seg)} launches, first start .. last end {(t1 - t0) / 1e3:.1f} us, "
      f"summed kernel time {sum(r[2] - r[1] for r in seg) / 1e3:.1f} us")
per = {}
for n, s, e, st in seg:
    v = per.setdefault(st, [0, 0.0, s, e])
    v[0] += 1
    v[1] += (e - s) / 1e3
    v[2], v[3] = min(v[2], s), max(v[3], e)
for st, v in per.items():
    print(f"  stream {st}: {v[0]} launches, kernel time {v[1]:.1f} us, active "
          f"{(v[2] - t0) / 1e3:.1f} .. {(v[3] - t0) / 1e3:.1f} us")
agg = {}
for n, s, e, st in seg:
    k = re.sub(r"\(anonymous namespace\)::", "", n)[:70]
    v = agg.setdefault(k, [0, 0.0])
    v[0] += 1
    v[1] += (e - s) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[0]:5d} {v[1]:9.1f} us  {k}")
if len(sys.argv) > 2:
    for n, s, e, st in seg:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} s{st} {re.sub(r'[(].*', '', n)[:60]}")
