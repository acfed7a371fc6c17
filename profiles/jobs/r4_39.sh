#!/bin/bash
# fused categorical action head: kernel tests, policy parity tests, bench line, tail step phases
O=$GRAFT_REPO_ROOT/gpurun_out/r04_39
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "action_head" -p no:cacheprovider 2>&1 | tail -5
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'act', d['config']['act_latency_ms_by_num_envs'])
P
timeout 300 python scripts/tail_probe.py > $O/tail_probe.txt 2>/dev/null; cat $O/tail_probe.txt
