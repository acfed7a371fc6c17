#!/bin/bash
# conv_m3 in the default dispatch: full GPU test tier, bench line, per-launch conv times, act latency
O=$GRAFT_REPO_ROOT/gpurun_out/r04_48
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed\|Error" $O/gpu_tests.log | tail -4
for v in 0 1; do
VLNCE_M3=$v timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench_m3_$v.json 2> $O/bench.err || tail -3 $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench_m3_$v.json').read().strip().split('\n')[-1]); r=d['roofline']
print('M3=$v', d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', r['kernel_ms_per_step'], r['frac'], r['bf16_pipe']['frac'], 'act', d['config']['act_latency_ms_by_num_envs'])
P
done
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -3 $O/conv_launch_times.txt
