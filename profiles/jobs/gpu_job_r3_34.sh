#!/bin/bash
# round 3, job 34: conv_s3 (short-K wide 1x1): parity (default + forced), per-layer times, bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ah
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "s3 or u3_scale or p3_1x1" 2>&1 | tail -4
ONLY=l1_1x1_64_256,l2_1x1_128_512
for v in 0 1; do
  VLNCE_S3=$v timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only $ONLY > $O/convbench_s3_$v.txt 2>&1
  grep "^l[12]_" $O/convbench_s3_$v.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('value', d['value'], d['ms_per_step'], d['config'].get('encode_ahead_ms_per_step'), 'conv', r['kernel_ms_per_step'], r['bf16_pipe']['frac'], r['per_launch_floor']['frac'], r['per_launch_floor']['hbm_bound'])"
