#!/bin/bash
# round 3, job 30: in-kernel timers of conv_u3 on the store-bound expansion layers
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ad
mkdir -p $O
cd $GRAFT_REPO_ROOT
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_p3time.so timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 1 --rounds 1 --only l1_1x1_64_256,l2_1x1_128_512,l3_1x1_256_1024 > $O/u3time.txt 2>&1
grep "u3 wave" $O/u3time.txt | sort | uniq -c | sort -rn | head -12 | cut -c1-330
