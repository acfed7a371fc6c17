#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02p
mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv or igemm or block or trunk" ) > $O/kt.log 2>&1
tail -15 $O/kt.log
timeout 300 python scripts/conv_accuracy.py --n 64 > $O/acc.log 2>&1; tail -8 $O/acc.log
for v in f32 x3; do
  export VLNCE_CONV_MATH=$v
  timeout 600 python scripts/convbench.py --n 64 --iters 20 > $O/conv_$v.log 2>&1
  timeout 600 python scripts/convbench.py --n 64 --iters 20 --mode train > $O/convtrain_$v.log 2>&1
  tail -1 $O/convtrain_$v.log
  timeout 300 python scripts/trunkbench.py > $O/trunk_$v.log 2>&1
  tail -3 $O/trunk_$v.log
done
paste <(awk '{print $1, $2,$3,$4, $(NF-2), $(NF-1)}' $O/conv_f32.log) <(awk '{print $(NF-2), $(NF-1)}' $O/conv_x3.log) | tail -26
