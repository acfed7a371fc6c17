#!/bin/bash
# round 4, job 26: full GPU suite + smoke + bench line on the fused-statistics build
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04z; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/gpu_suite.txt 2>&1
echo "gpu suite rc=$?"; tail -2 $out/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 300 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | tail -1 > $out/bench.json
python -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']
print('ms/step', d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'bf16 frac', r['bf16_pipe']['frac'], 'floor frac', r['per_launch_floor']['frac'], 'act', d['config']['act_latency_ms_by_num_envs'])"
