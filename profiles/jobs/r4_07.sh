#!/bin/bash
# round 4, job 7: options ABI + s3 (pipelined epilogue) as the default: kernel tests, bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04g; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider > $out/kernel_tests.txt 2>&1
echo "kernel tests rc=$?"; tail -3 $out/kernel_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | tail -1 > $out/bench.json
python -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']
print('ms/step', d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'bf16 frac', r['bf16_pipe']['frac'], 'floor frac', r['per_launch_floor']['frac'])"
