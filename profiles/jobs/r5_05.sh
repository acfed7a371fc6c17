#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_05
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TOP=40
echo "== default"; timeout 300 python scripts/n64_grad_diff.py cma_update_n64_256 2>/dev/null | tee $O/default.txt | head -44
echo "== VLNCE_SIDE_STREAMS=0"; VLNCE_SIDE_STREAMS=0 timeout 300 python scripts/n64_grad_diff.py cma_update_n64_256 2>/dev/null | tee $O/no_side.txt | head -12
echo "== VLNCE_LINEAR_PLANES=0"; VLNCE_LINEAR_PLANES=0 timeout 300 python scripts/n64_grad_diff.py cma_update_n64_256 2>/dev/null | tee $O/no_planes.txt | head -12
echo "== VLNCE_CONV_MATH=f32"; VLNCE_CONV_MATH=f32 timeout 300 python scripts/n64_grad_diff.py cma_update_n64_256 2>/dev/null | tee $O/f32.txt | head -12
