#!/bin/bash
# round 3, job 51: conv_s3 K=128 variant without scratch (prologue vectors in LDS)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "s3 or p3_1x1 or u3_scale" 2>&1 | tail -1
timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only l1_1x1_64_256,l2_1x1_128_512 2>&1 | grep "^l[12]_"
timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('value', d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], r['bf16_pipe']['frac'], r['per_launch_floor']['frac'])"
