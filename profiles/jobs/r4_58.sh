#!/bin/bash
# end-of-round verification: full GPU test tier, smoke, the default bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_final; mkdir -p $out
( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $out/gpu_suite.txt 2>&1
echo "gpu suite rc=$?"; grep -n "passed\|failed" $out/gpu_suite.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 600 python bench.py 2>$out/bench.err | tail -1 > $out/bench.json
python -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']
print('ms/step', d['ms_per_step'], 'value', d['value'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'frac', r['frac'], 'bf16 frac', r['bf16_pipe']['frac'], 'floor frac', r['per_launch_floor']['frac'], 'traffic', r['traffic'], 'f32', d['config'].get('fp32_mfma_only',{}).get('ms_per_step'), 'cpu', d.get('cpu_baseline',{}).get('value'), 'act', d['config']['act_latency_ms_by_num_envs'])"
