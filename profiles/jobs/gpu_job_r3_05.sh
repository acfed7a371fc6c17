#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03e
mkdir -p $O
cd $GRAFT_REPO_ROOT
SEL=l1_3x3,l2_3x3_,l3_3x3_,l4_3x3_,l1_1x1_64_256,l2_1x1_128_512,l3_1x1_256_1024,l3_1x1_1024_256,l4_1x1_512_2048
for v in base pr_m0p0 pr_m1p2 pr_m3p3 pr_m2p3; do
  lib=$GRAFT_REPO_ROOT/build/variants/libvlnce_$v.so
  [ $v = base ] && lib=$GRAFT_REPO_ROOT/vln-ce_amd/libvlnce_hip.so
  VLNCE_HIP_LIB=$lib timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --only $SEL > $O/cb_$v.txt 2>&1
done
paste <(grep "^l[1-4]" $O/cb_base.txt | awk '{printf "%-20s %8s\n", $1,$5}') <(grep "^l[1-4]" $O/cb_pr_m0p0.txt | awk '{print $5}') <(grep "^l[1-4]" $O/cb_pr_m1p2.txt | awk '{print $5}') <(grep "^l[1-4]" $O/cb_pr_m3p3.txt | awk '{print $5}') <(grep "^l[1-4]" $O/cb_pr_m2p3.txt | awk '{print $5}')
