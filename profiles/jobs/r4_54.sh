#!/bin/bash
# how the measured step time depends on the number of timed / warm-up steps (same box)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_54
mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "30 4" "100 10" "30 4" "300 20" "100 10"; do
set -- $cfg
timeout 600 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-f32-compare > $O/b.json 2> $O/bench.err || tail -3 $O/bench.err
python - <<P
import json
d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('steps $1 warmup $2:', d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'])
P
done 2>&1 | tee $O/steps_sweep.txt
