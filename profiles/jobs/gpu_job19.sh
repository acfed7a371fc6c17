#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02r
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed" $O/gpu_tests.log | tail -2
if grep -q "failed\|error" $O/gpu_tests.log; then tail -120 $O/gpu_tests.log | head -100; fi
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["no_pipeline_ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_per_step"), d["config"]["act_latency_ms_by_num_envs"])
PY
timeout 300 python scripts/trunkbench.py 2>&1 | tail -3
