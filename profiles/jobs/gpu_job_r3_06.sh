#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_p3 and not every_tile" -p no:cacheprovider > $O/p3_tests.log 2>&1
echo "p3 tests rc=$?"; tail -3 $O/p3_tests.log
for p3 in 0 1 2; do
  VLNCE_P3=$p3 timeout 300 python scripts/convbench.py --mode train --pro --set r50,depth,r18 --iters 10 --rounds 3 > $O/cb_train_p3_$p3.txt 2>&1
  VLNCE_P3=$p3 timeout 300 python scripts/convbench.py --mode eval --set r50,depth,r18 --iters 10 --rounds 3 > $O/cb_eval_p3_$p3.txt 2>&1
done
for m in train eval; do
paste <(grep -v amdgpu $O/cb_${m}_p3_0.txt | awk '{printf "%-22s %8s %6s %5s %9s\n", $1,$2,$3,$4,$5}') <(grep -v amdgpu $O/cb_${m}_p3_1.txt | awk '{printf "%9s %7s\n", $5,$6}') <(grep -v amdgpu $O/cb_${m}_p3_2.txt | awk '{printf "%9s\n", $5}') | tee $O/ab_$m.txt
done
