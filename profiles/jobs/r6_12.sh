#!/bin/bash
# round 6, job 12: BatchNorm vectors finished in the consumer's prologue (vlnce_bn_pending) -- kernel test,
# policy tests, bench A/B against VLNCE_BN_PROLOGUE=0 (one finalize launch behind every convolution) on one box
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_12
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "pending or bn or conv2d_fwd" 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_policy_gpu.py tests/test_policy_sizes_gpu.py tests/test_dagger_hooks_gpu.py -q -x 2>&1 | tail -12
for rep in 1 2; do
for m in 1 0; do
  VLNCE_BN_PROLOGUE=$m timeout 600 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline > $O/bench_$m.json 2> $O/bench_$m.err
  python - <<P
import json
d=json.loads(open('$O/bench_$m.json').read().strip().split('\n')[-1]); r=d['roofline']
print('VLNCE_BN_PROLOGUE=$m', d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], 'trunks eager', r['eager_single_stream_trunks_ms'])
P
done; done | tee $O/bench_ab.txt
