#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q ) > $O/kernel_tests.log 2>&1
tail -3 $O/kernel_tests.log
if ! grep -q " passed" $O/kernel_tests.log || grep -q "failed" $O/kernel_tests.log; then echo KERNEL TESTS FAILED; tail -80 $O/kernel_tests.log | head -70; fi
for cfg in "VLNCE_IGEMM_NO_DMA=1" "VLNCE_PK_TILES=1" "VLNCE_PK_TILES=4" "VLNCE_PK_TILES=100000"; do
  echo "== $cfg"
  env $cfg timeout 300 python scripts/convbench.py --mode train > $O/convbench_$cfg.txt 2>&1
  tail -27 $O/convbench_$cfg.txt | awk '{printf "%s %s %s | ", $1, $5, $6} END {print ""}'
done
( time timeout 900 python -m pytest tests/test_policy_gpu.py -x -q ) > $O/policy_tests.log 2>&1
tail -3 $O/policy_tests.log
if grep -q "failed" $O/policy_tests.log; then tail -80 $O/policy_tests.log | head -70; fi
