#!/bin/bash
# round-5 evidence on the final build: full GPU test tier, smoke, the default bench line, rocprofv3 kernel
# stats of the bench command, PMC HBM traffic, per-launch conv times, step phase probe, secondary workloads
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_40
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], 'fp32peak', r['fp32_mfma_peak']['frac'], 'bf16', r['bf16_pipe']['frac'], 'traffic', r['traffic'])
print('f32 only', d['config'].get('fp32_mfma_only'), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'), 'act', d['config']['act_latency_ms_by_num_envs'], d['config']['act_fwd_only_eval_steps_per_sec_per_gpu'])
P
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --pmc-step > $O/pmc_$c.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_pmc.py "$(find $O/pmc_$c -name '*.db' | head -1)" > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
python scripts/pmc_traffic_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/r05_pmc_traffic.json "profiles/r05_zz_pmc_fetch_size.txt, r05_zz_pmc_write_size.txt" | cut -c1-200
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -2 $O/conv_launch_times.txt
timeout 300 python scripts/backward_phase_probe.py 2>/dev/null > $O/host_issue_vs_gpu_time_per_phase.txt; tail -16 $O/host_issue_vs_gpu_time_per_phase.txt
timeout 400 python bench.py --policy waypoint --steps 10 --warmup 3 > $O/bench_waypoint.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_waypoint.json | sed 's/^/waypoint /'
timeout 400 python bench.py --policy seq2seq --steps 20 > $O/bench_seq2seq.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_seq2seq.json | sed 's/^/seq2seq /'
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err; tail -c 300 $O/bench_data_path.json; echo
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_trainable.json | sed 's/^/trainable /'
