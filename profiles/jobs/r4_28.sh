#!/bin/bash
# round 4, job 28: full GPU suite, smoke, bench line, secondary configs with their roofline blocks
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04zb; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/gpu_suite.txt 2>&1
echo "gpu suite rc=$?"; tail -2 $out/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 400 python bench.py 2>/dev/null | tail -1 > $out/bench.json
python -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']
print('ms/step', d['ms_per_step'], 'value', d['value'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'frac', r['frac'], 'bf16 frac', r['bf16_pipe']['frac'], 'floor frac', r['per_launch_floor']['frac'], 'f32', d['config'].get('fp32_mfma_only',{}).get('ms_per_step'), 'cpu', d.get('cpu_baseline',{}).get('value'))"
timeout 600 python scripts/bench_policies.py > $out/bench_other_policies.jsonl 2>/dev/null; cut -c1-400 $out/bench_other_policies.jsonl
