#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_22
mkdir -p $O
cd $GRAFT_REPO_ROOT
PYTHONFAULTHANDLER=1 timeout 300 python scripts/linear_rows_bench.py > $O/linear_rows_bench.txt 2> $O/err.txt
cat $O/linear_rows_bench.txt; grep -v "Warn\|warn\|amdgpu" $O/err.txt | tail -25
