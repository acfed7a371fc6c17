#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03d
mkdir -p $O
cd $GRAFT_REPO_ROOT
SEL=l1_3x3,l2_3x3_,l3_3x3_
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_p3time.so timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 1 --only $SEL > $O/p3time.txt 2>&1
grep "^p3 prod\|^l[1-4]" $O/p3time.txt | awk '!seen[$0]++' | cut -c1-300 | head -40
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_p3time.so timeout 200 python scripts/convbench.py --mode eval --set r50 --iters 1 --only $SEL > $O/p3time_eval.txt 2>&1
grep "^p3 prod\|^l[1-4]" $O/p3time_eval.txt | awk '!seen[$0]++' | cut -c1-300 | head -40
