#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_02
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/golden_errors.py cma_update_n64_256 2>/dev/null | tee $O/golden_n64_errors.txt
