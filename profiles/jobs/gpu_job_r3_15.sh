#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03o
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "(conv_p3 or conv2d_fwd or bottleneck or block) and not every_tile" -p no:cacheprovider > $O/u3_tests.log 2>&1
echo "u3 tests rc=$?"; tail -3 $O/u3_tests.log
for u in 0 1; do
  VLNCE_U3=$u timeout 300 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only 1x1 > $O/cb_train_u3_$u.txt 2>&1
done
paste <(grep "^l[1-4]" $O/cb_train_u3_0.txt | awk '{printf "%-22s %8s %6s %5s %9s\n", $1,$2,$3,$4,$5}') <(grep "^l[1-4]" $O/cb_train_u3_1.txt | awk '{printf "%9s %7s\n", $5,$6}') | tee $O/ab_u3.txt
