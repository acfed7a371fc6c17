#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03l
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/cross_kernel_floor.py > $O/cross_kernel_floor.txt 2>&1
cat $O/cross_kernel_floor.txt | grep -v amdgpu
timeout 900 python -m pytest tests/test_policy_sizes_gpu.py -x -q -m gpu -k "planes_vs or seq2seq" -p no:cacheprovider > $O/new_tests.log 2>&1
echo "new tests rc=$?"; tail -15 $O/new_tests.log | cut -c1-300
