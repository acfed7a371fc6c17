#!/bin/bash
# round 6, job 17: final state (HEAD): GPU tier, smoke, the default bench line, kernel stats of the bench command
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_17
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'])
print('f32 only', d['config'].get('fp32_mfma_only'), 'cpu', d.get('cpu_baseline', {}).get('value'), 'act', d['config']['act_latency_ms_by_num_envs'], d['config']['act_fwd_only_eval_steps_per_sec_per_gpu'])
P
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
find $O/kt -name "*stats*" | head -5
rm -rf $O/kt
head -12 $O/bench_kernel_stats.md | cut -c1-150
timeout 600 python bench.py --policy waypoint --steps 10 --warmup 3 > $O/bench_waypoint.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_waypoint.json | sed 's/^/waypoint /'
