#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02q
mkdir -p $O
cd $GRAFT_REPO_ROOT
L=l2_3x3,l2_1x1_512_128,l3_1x1_512_256,l4_1x1_1024_512,l4_1x1s2,l1_1x1_64_256,l2_1x1_128_512
for v in $VARIANTS; do
  if [ $v = prod ]; then unset VLNCE_HIP_LIB; else export VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_$v.so; fi
  echo "== $v"
  timeout 600 python scripts/convbench.py --n 64 --iters 20 --only $L 2>&1 | grep -v amdgpu.ids | awk '{print $1, $3, $(NF-2), $(NF-1)}'
done
