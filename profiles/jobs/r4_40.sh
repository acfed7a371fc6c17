#!/bin/bash
# A/B of the fused action head on one box: bench plain loop, 3 alternations
O=$GRAFT_REPO_ROOT/gpurun_out/r04_40
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in 0 1; do
VLNCE_ACTION_HEAD=$v timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench_$v.json 2> $O/bench.err || tail -3 $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench_$v.json').read().strip().split('\n')[-1])
print('ACTION_HEAD=$v', d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'act', d['config']['act_latency_ms_by_num_envs'])
P
done
done 2>&1 | tee $O/ab.txt
