#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_06
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/n64_rowwise.py 2>&1 | grep -v "Warn\|warn\|super()\|amdgpu.ids" | tee $O/rowwise.txt | tail -30
echo "== all masks one"
ALL_MASKS_ONE=1 timeout 300 python scripts/n64_rowwise.py 2>&1 | grep -v "Warn\|warn\|super()\|amdgpu.ids" | tee $O/rowwise_masks1.txt | tail -12
