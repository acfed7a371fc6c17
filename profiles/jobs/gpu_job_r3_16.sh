#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $O
cd $GRAFT_REPO_ROOT
SEL=l3_1x1_256_1024,l4_1x1_1024_512
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_p3time.so timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 1 --rounds 1 --only $SEL > $O/u3time.txt 2>&1
grep "^u3" $O/u3time.txt | awk '!seen[$0]++' | cut -c1-300 | head -12
