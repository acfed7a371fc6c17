#!/bin/bash
# A/B of conv_m3 in the whole step, alternating on one box
O=$GRAFT_REPO_ROOT/gpurun_out/r04_49
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 1 0; do
VLNCE_M3=$v timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench_m3_$v.json 2> $O/bench.err || tail -3 $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench_m3_$v.json').read().strip().split('\n')[-1]); r=d['roofline']
print('M3=$v', d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', r['kernel_ms_per_step'], 'eager trunks', r['eager_single_stream_trunks_ms'], 'act', d['config']['act_latency_ms_by_num_envs'])
P
done
done 2>&1 | tee $O/ab.txt
for v in 1 0; do VLNCE_M3=$v timeout 300 python scripts/trunkbench.py --n 64 --iters 20 2>/dev/null | tee -a $O/ab.txt; done
