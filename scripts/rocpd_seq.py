"""Every kernel of the last-but-one step of a rocprofv3 rocpd database in start order:
offset (us), duration (us), gap to the previous kernel's end on ANY stream, stream, name.
python scripts/rocpd_seq.py <db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
adam = [i for i, r in enumerate(rows) if "TensorListScalarListMetadata" in r[0]]
clusters = []
for i in adam:
    if clusters and i - clusters[-1][-1] < 200:
        clusters[-1].append(i)
    else:
        clusters.append([i])
a, b = clusters[-3][-1] + 1, clusters[-2][-1] + 1
step = rows[a:b]
t0 = step[0][1]
last_end = t0
print(f"{len(step)} kernels, span {(max(r[2] for r in step) - t0) / 1e3:.1f} us")
for name, s, e, st, q in step:
    n = re.sub(r"\(anonymous namespace\)::|^void |at::native::|vlnce_detail::", "", name)[:90]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {(s - last_end) / 1e3:7.1f}  s{st}q{q}  {n}")
    last_end = max(last_end, e)
