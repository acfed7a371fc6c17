#!/bin/bash
# round 6, job 11: evidence set on commit 73ff0f2: full GPU tier, smoke, the default bench line (cpu baseline,
# fp32-MFMA comparison), rocprofv3 kernel stats of the bench command, PMC HBM traffic passes, per-launch conv
# times, MFMA-busy per layer, act() eager vs whole-call graph, secondary workloads
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_11
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], 'fp32peak', r['fp32_mfma_peak']['frac'], 'pipe', r['bf16_pipe']['frac'], 'traffic', r['traffic'])
print('f32 only', d['config'].get('fp32_mfma_only'), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'), 'act', d['config']['act_latency_ms_by_num_envs'], d['config']['act_fwd_only_eval_steps_per_sec_per_gpu'])
P
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --pmc-step > $O/pmc_$c.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_pmc.py "$(find $O/pmc_$c -name '*.db' | head -1)" > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
python scripts/pmc_traffic_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/r06_pmc_traffic.json "profiles/r06_zz_pmc_fetch_size.txt, r06_zz_pmc_write_size.txt" | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -30 $O/bench_kernel_stats.md | cut -c1-150
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -2 $O/conv_launch_times.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_busy -- python $GRAFT_REPO_ROOT/scripts/convbench.py --set r50 --mode train --pro --iters 3 > $O/convbench_pmc.txt 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_pmc_layers.py "$(find $O/pmc_busy -name '*.db' | head -1)" 6 conv_ > $O/pmc_mfma_busy_per_layer.txt 2>&1
rm -rf $O/pmc_busy
head -40 $O/pmc_mfma_busy_per_layer.txt | cut -c1-160
python - > $O/act_eager_vs_graph.txt 2>&1 <<P
import os, sys, time, torch
sys.path.insert(0, '.')
import bench, vlnce_amd
dev = torch.device('cuda:0')
for mode in ('0', '1'):
    os.environ['VLNCE_ACT_GRAPH'] = mode
    torch.manual_seed(0)
    pol = vlnce_amd.build_model(vlnce_amd.make_config('CMAPolicy'), *vlnce_amd.make_spaces(256, 256)).to(dev)
    pol.eval()
    b = bench.synth_batch(8, 256, 80, dev)
    print('VLNCE_ACT_GRAPH=' + mode, bench.act_latency(pol, b, dev, sizes=(1, 4, 8), iters=50))
P
cat $O/act_eager_vs_graph.txt
timeout 400 python bench.py --policy seq2seq --steps 20 > $O/bench_seq2seq.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_seq2seq.json | sed 's/^/seq2seq /'
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err; tail -c 300 $O/bench_data_path.json; echo
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_trainable.json | sed 's/^/trainable /'
timeout 300 python bench.py --force-dist --steps 20 --no-cpu-baseline --no-f32-compare --no-pipeline > $O/bench_force_dist.json 2>/dev/null; grep -o '"allreduce_ms": [0-9.]*\|"ms_per_step": [0-9.]*\|"allreduce_hidden_frac": [0-9.]*' $O/bench_force_dist.json | tr '\n' ' '; echo
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
