#!/bin/bash
# round 3, job 53: conv_u3 with the dual kind as a template parameter (identity skip: no in-loop scratch): parity
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03dual
timeout 110 python -m pytest tests/test_kernels_gpu.py tests/test_policy_sizes_gpu.py -x -q -p no:cacheprovider \
  -k "dual or u3_forced or bench_geometry or bottleneck" > gpurun_out/r03dual/tests.txt 2>&1
echo "rc=$?"; tail -5 gpurun_out/r03dual/tests.txt
