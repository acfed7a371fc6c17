#!/bin/bash
# round 3, job 37: conv_s3 with LDS-turned 16-byte row stores
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ak
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "s3 or u3_scale or p3_1x1" 2>&1 | tail -2
timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only l1_1x1_64_256,l2_1x1_128_512 2>&1 | grep "^l[12]_"
