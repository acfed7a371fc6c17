#!/bin/bash
# last check of the committed tree: full GPU tier, smoke, default bench command timed as the driver runs it
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_51
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
SECONDS=0
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "default bench command: ${SECONDS}s wall"
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
P
