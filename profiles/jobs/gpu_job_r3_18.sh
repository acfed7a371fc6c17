#!/bin/bash
# round 3, job 18: conv_x3 with the round-to-nearest split (kernel tests, cross-kernel test, accuracy file);
# kernel breakdown of ONE cached-feature CMA update (what the rollout recurrence costs)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03r
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_policy_sizes_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed" $O/gpu_tests.log | tail -2
VLNCE_P3=0 VLNCE_U3=0 timeout 300 python scripts/conv_accuracy.py --n 16 > $O/conv_accuracy_x3_rne.txt 2>&1
grep -v amdgpu $O/conv_accuracy_x3_rne.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/bench_data_path.py --update-only --iters 6 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $O/kt.log
db=$(find $O/kt -name "*.db" | head -1)
python - "$db" > $O/cached_update_one_step.txt <<'EOF'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
# the last update = dispatches after the last Adam launch before the final one: take the last 1/10 by count
names = [r[0] for r in rows]
# find Adam (multi_tensor) launches as step delimiters
idx = [i for i, n in enumerate(names) if "multi_tensor" in n or "adam" in n.lower()]
ends = []
for i in idx:
    if not ends or i - ends[-1] > 50:
        ends.append(i)
    else:
        ends[-1] = i
print("step delimiters", len(ends))
if len(ends) >= 2:
    a, b = ends[-2] + 1, ends[-1] + 1
else:
    a, b = 0, len(rows)
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
# gaps
gap = 0.0
for (n0, s0, e0), (n1, s1, e1) in zip(seg[:-1], seg[1:]):
    if s1 > e0:
        gap += (s1 - e0) / 1e3
print(f"idle gaps between consecutive launches: {gap:.1f} us")
EOF
rm -rf $O/kt
cat $O/cached_update_one_step.txt | cut -c1-140
