#!/bin/bash
# round 4, job 30: residual in the bf16-plane kernels' epilogues: conv tests, eval-mode numbers
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04zd; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "conv" > $out/tests_conv.txt 2>&1
echo "conv tests rc=$?"; tail -2 $out/tests_conv.txt
timeout 600 python scripts/bench_policies.py --which waypoint > $out/bench_waypoint.jsonl 2>/dev/null; cut -c1-330 $out/bench_waypoint.jsonl
timeout 300 python bench.py --bn eval --no-cpu-baseline --no-f32-compare 2>/dev/null | tail -1 > $out/bench_bn_eval.json
python -c "
import json; d=json.load(open('$out/bench_bn_eval.json')); r=d['roofline']
print('bn=eval: ms/step', d['ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'bf16 launches', r['bf16_pipe']['launches'], 'f32 launches', r['fp32_mfma']['launches'], 'act fwd steps/s', d['config']['act_fwd_only_eval_steps_per_sec_per_gpu'], 'act latency', d['config']['act_latency_ms_by_num_envs'])"
