#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_23
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear" 2>&1 | tail -4
timeout 300 python scripts/linear_rows_bench.py 2>/dev/null | tee $O/linear_rows_bench.txt
