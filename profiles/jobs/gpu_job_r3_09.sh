#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03i
mkdir -p $O
cd $GRAFT_REPO_ROOT
SEL=l1_1x1_64_256,l2_1x1_128_512,l3_1x1_256_1024,l3_1x1_1024_256,l4_1x1_1024_512,l4_1x1_512_2048
VLNCE_U3=1 VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_p3time.so timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 1 --rounds 1 --only $SEL > $O/u3time.txt 2>&1
grep "^u3\|^l[1-4]" $O/u3time.txt | awk '!seen[$0]++' | cut -c1-300 | head -60
