#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_25
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/trainable_step.py 5 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
grep "ms/step" $O/kt.log
python - <<P
import sqlite3, glob, re
db = sqlite3.connect(glob.glob('$O/kt/**/*.db', recursive=True)[0])
rows = list(db.execute("select name, start, end from kernels order by start"))
n = len(rows)
# the last 5/8 of the dispatches are the 5 timed steps (3 warm-up steps in front)
rows = rows[int(n * 3 / 8):]
agg = {}
for name, s, e in rows:
    k = re.sub(r"\(anonymous namespace\)::|vlnce_detail::|^void ", "", name)[:70]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"dispatches in 5 steps: {len(rows)}; kernel time per step {tot/5/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[1]/5:9.1f} us/step {v[0]/5:7.1f} calls/step  {k}")
P
rm -rf $O/kt
