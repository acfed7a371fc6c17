#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02o
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in base split; do
  export VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_$v.so
  echo "== $v"
  timeout 300 python scripts/conv_accuracy.py > $O/acc_$v.log 2>&1; tail -8 $O/acc_$v.log
  timeout 600 python scripts/convbench.py --n 64 --iters 20 > $O/conv_$v.log 2>&1
  tail -26 $O/conv_$v.log | awk '{print $1, $(NF-2), $(NF-1)}' | column -t
  timeout 300 python scripts/trunkbench.py > $O/trunk_$v.log 2>&1
  tail -3 $O/trunk_$v.log
done
export VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_split.so
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv or igemm or block or trunk" ) > $O/kt.log 2>&1
tail -5 $O/kt.log
