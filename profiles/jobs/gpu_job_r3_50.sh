#!/bin/bash
# round 3, job 50: conv_u3 with the stride-1 case as a compile-time variant (no scratch in its chunk loop)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03aq
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "u3" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_policy_sizes_gpu.py -m gpu -x -q -p no:cacheprovider -k "bench_geometry" 2>&1 | tail -1
ONLY=l2_1x1s2_256_512,l3_1x1_256_1024,l3_1x1_512_256,l3_1x1s2_512_1024,l4_1x1_1024_512,l4_1x1s2_1024_2048,l4_1x1_512_2048
timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only $ONLY > $O/convbench_u3.txt 2>&1
grep "^l[1-4]_" $O/convbench_u3.txt
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('value', d['value'], d['ms_per_step'], d['config'].get('encode_ahead_ms_per_step'), 'conv', r['kernel_ms_per_step'], r['bf16_pipe']['frac'], r['bf16_pipe']['by_kernel'])"
