#!/bin/bash
# last check: default bench command (bound rank, ascending cpu sweep), smoke, policy GPU tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_57
mkdir -p $O
cd $GRAFT_REPO_ROOT
/usr/bin/time -f "bench wall %e s" timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
grep -E "bound|cpu_baseline|wall" $O/bench.err | head -12
python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5_57/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["steps_per_sec_by_threads"], d["config"]["host_threads"])
PY
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 280 python -m pytest tests/test_policy_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_policy.txt
