#!/bin/bash
# round 5, first call: the N=64 reference golden on the GPU, the bench line of the round-4 kernels with the
# reference's read-backs in the loop, host/GPU phase probe
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_01
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "n64 or cma_update_64" 2>&1 | tail -5 | tee $O/golden_n64.txt
timeout 600 python bench.py --no-f32-compare > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', r['kernel_ms_per_step'], r['frac'], 'bf16', r['bf16_pipe']['frac'], 'cpu', d.get('cpu_baseline'))
P
timeout 300 python scripts/host_vs_gpu_probe.py > $O/host_vs_gpu.txt 2>&1
cat $O/host_vs_gpu.txt | tail -12
