#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_07
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q 2>&1 | tail -15 | tee $O/test_policy_gpu.txt
