#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_06
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_dagger_hooks_gpu.py -q 2>&1 | tail -60 > $O/pytest_a.txt
tail -12 $O/pytest_a.txt
timeout 1500 python -m pytest tests/test_policy_gpu.py tests/test_policy_sizes_gpu.py -q 2>&1 | tail -120 > $O/pytest_b.txt
tail -12 $O/pytest_b.txt
timeout 300 python scripts/convbench.py --mode train --pro --backlog --only 3x3 > $O/convbench_3x3.txt 2>&1
cat $O/convbench_3x3.txt
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], r['bf16_pipe']['frac'])
P
