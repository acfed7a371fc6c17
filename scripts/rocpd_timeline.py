"""Timeline view of the last bench step in a rocprofv3 rocpd database: busy-time union,
idle gaps, per-stream totals.  python scripts/rocpd_timeline.py <db> [n_steps_in_run]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# steps are delimited by the Adam multi_tensor kernels; take the region between the last two
adam = [i for i, r in enumerate(rows) if "TensorListScalarListMetadata" in r[0]]
# group consecutive adam kernels into clusters
clusters = []
for i in adam:
    if clusters and i - clusters[-1][-1] < 200:
        clusters[-1].append(i)
    else:
        clusters.append([i])
skip_last = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # attribution pass is the last step
a, b = clusters[-2 - skip_last][-1] + 1, clusters[-1 - skip_last][-1] + 1
step = rows[a:b]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"step: {len(step)} kernels, span {(t1 - t0)/1e6:.3f} ms")
iv = sorted((r[1], r[2]) for r in step)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (union) {busy/1e6:.3f} ms, idle {((t1 - t0) - busy)/1e6:.3f} ms, "
      f"sum of kernel durations {sum(r[2]-r[1] for r in step)/1e6:.3f} ms")
gaps.sort(reverse=True)
print("largest gaps (us @ offset ms):", [(round(g / 1e3, 1), round(o / 1e6, 2)) for g, o in gaps[:12]])
per = {}
for r in step:
    k = (r[3], r[4])
    per.setdefault(k, [0, 0, r[1], r[2]])
    per[k][0] += 1
    per[k][1] += r[2] - r[1]
    per[k][3] = max(per[k][3], r[2])
for k, v in per.items():
    print(f"stream/queue {k}: {v[0]} kernels, {v[1]/1e6:.3f} ms, active {(v[2]-t0)/1e6:.2f}..{(v[3]-t0)/1e6:.2f} ms")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:60]


agg = {}
for r in step:
    a_ = agg.setdefault(short(r[0]), [0, 0])
    a_[0] += 1
    a_[1] += r[2] - r[1]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {v[1]/1e3:9.1f} us  x{v[0]:4d}  {k}")

# per-stream run-length view: consecutive launches of the same kernel collapsed
print("\nper-stream sequence (offset ms, duration us incl. gaps inside the run, kernel x count):")
for key in per:
    seq = [r for r in step if (r[3], r[4]) == key]
    print(f"-- stream/queue {key}")
    i = 0
    out = []
    while i < len(seq):
        j = i
        while j + 1 < len(seq) and short(seq[j + 1][0]) == short(seq[i][0]):
            j += 1
        out.append(((seq[i][1] - t0) / 1e6, (seq[j][2] - seq[i][1]) / 1e3, short(seq[i][0]), j - i + 1))
        i = j + 1
    # merge into coarse 0.25 ms buckets to keep the listing short
    bucket = {}
    for off, dur, name, cnt in out:
        b = int(off / 0.5)
        d = bucket.setdefault(b, {})
        d[name] = d.get(name, 0) + dur
    for b in sorted(bucket):
        top = sorted(bucket[b].items(), key=lambda kv: -kv[1])[:3]
        print(f"  {b*0.5:5.1f} ms: " + "; ".join(f"{n[:44]} {d:.0f}us" for n, d in top))
