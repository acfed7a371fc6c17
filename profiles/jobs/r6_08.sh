#!/bin/bash
# round 6, job 8: conv_p3 KxK producers back on twelve offset registers; prologue vectors in LDS (main
# build) against per-chunk global loads (variant vecglobal); tests; bench; then where the time is
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_08
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in main vecglobal; do
  L=$GRAFT_REPO_ROOT/vln-ce_amd/libvlnce_hip.so; [ $v = vecglobal ] && L=$GRAFT_REPO_ROOT/build/variants/libvlnce_vecglobal.so
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode train --pro --backlog --only 3x3 > $O/convbench_3x3_$v.txt 2>&1
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode eval --backlog --set r18 --n 416 > $O/convbench_r18_$v.txt 2>&1
  cat $O/convbench_3x3_$v.txt $O/convbench_r18_$v.txt | grep -v amdgpu
done
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], r['bf16_pipe']['frac'], 'act', d['config']['act_latency_ms_by_num_envs'], d['config']['act_fwd_only_eval_steps_per_sec_per_gpu'])
P
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
timeout 300 python scripts/backward_phase_probe.py 2>/dev/null > $O/phase_probe.txt; tail -18 $O/phase_probe.txt
timeout 600 python bench.py --policy waypoint --steps 10 --warmup 3 > $O/bench_waypoint.json 2>$O/bench_waypoint.err; tail -c 1500 $O/bench_waypoint.json; echo
timeout 600 python bench.py --policy seq2seq --steps 20 --no-cpu-baseline > $O/bench_seq2seq.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_seq2seq.json | sed 's/^/seq2seq /'
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_trainable.json | sed 's/^/trainable /'
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt_w -- python $GRAFT_REPO_ROOT/bench.py --policy waypoint --steps 5 --warmup 3 --no-cpu-baseline > $O/kt_w.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt_w -name '*.db' | head -1)" $O/waypoint_kernel_stats.md 900 > /dev/null
rm -rf $O/kt_w
head -45 $O/waypoint_kernel_stats.md
