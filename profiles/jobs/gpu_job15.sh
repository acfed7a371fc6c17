#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02n
mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv or igemm or block or trunk" ) > $O/kt.log 2>&1
tail -3 $O/kt.log
for v in base rot; do
  if [ $v = base ]; then export VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_base.so; else unset VLNCE_HIP_LIB; fi
  timeout 600 python scripts/convbench.py --n 64 --iters 20 > $O/conv_$v.log 2>&1
  tail -4 $O/conv_$v.log
  timeout 300 python scripts/trunkbench.py > $O/trunk_$v.log 2>&1
  tail -3 $O/trunk_$v.log
done
