#!/bin/bash
# round-4 evidence on the final kernels: depth trunk per layer with / without conv_m3, PMC HBM traffic,
# rocprofv3 kernel stats of the bench command, the default bench line, per-launch conv times, secondary configs
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04zz2
mkdir -p $O
cd $GRAFT_REPO_ROOT
for opt in "m3=0" "m3=1"; do
  echo "== options '$opt'"
  timeout 300 python scripts/convbench.py --set depth --mode train --backlog --iters 30 --opt "$opt" 2>/dev/null
done > $O/depth_convbench_m3.txt
grep "trunk total" $O/depth_convbench_m3.txt
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --pmc-step > $O/pmc_$c.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_pmc.py "$(find $O/pmc_$c -name '*.db' | head -1)" > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
python scripts/pmc_traffic_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt profiles/archive/r04_pmc_traffic.json "profiles/archive/r04_zz_pmc_fetch_size.txt, r04_zz_pmc_write_size.txt" | cut -c1-200
cp profiles/archive/r04_pmc_traffic.json $O/r04_pmc_traffic.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', r['kernel_ms_per_step'], r['frac'], 'bf16', r['bf16_pipe']['frac'], 'floor', r['per_launch_floor']['frac'], 'traffic', r['traffic'], r['launches_per_step'])
print('f32 only', d['config'].get('fp32_mfma_only'), 'cpu', d.get('cpu_baseline', {}).get('value'), 'act', d['config']['act_latency_ms_by_num_envs'], d['config']['act_fwd_only_eval_steps_per_sec_per_gpu'])
print(json.dumps(r['bf16_pipe']['by_kernel']), json.dumps(r['fp32_mfma'])[:200])
P
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -2 $O/conv_launch_times.txt; grep -c " m3 " $O/conv_launch_times.txt
timeout 400 python scripts/bench_policies.py > $O/bench_other_policies.jsonl 2> $O/bench_other_policies.err
python - <<P
import json
for l in open('$O/bench_other_policies.jsonl'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d.get('config',{}).get('workload', d.get('policy')), d.get('value'), d.get('ms_per_step'), d.get('config',{}).get('encode_ahead_ms_per_step'))
P
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err
tail -c 300 $O/bench_data_path.json; echo
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_trainable.json').read().strip().split('\n')[-1]); print('trainable', d['value'], d['ms_per_step'])"
