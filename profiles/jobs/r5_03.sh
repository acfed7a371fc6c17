#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_03
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/n64_grad_diff.py cma_update_n64_256 2>/dev/null | tee $O/n64_grad_diff.txt
timeout 300 python scripts/n64_grad_diff.py cma_update_n64_256 8 2>/dev/null | tee $O/n8_grad_diff.txt
