#!/bin/bash
# last check: default bench command (bound rank, ascending cpu sweep)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_58
mkdir -p $O
cd $GRAFT_REPO_ROOT
T0=$(date +%s)
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - T0 )) s rc=$?" | tee $O/wall.txt
grep -E "bound|cpu_baseline" $O/bench.err | head -12
python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5_58/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["steps_per_sec_by_threads"], d["config"]["host_threads"])
PY
