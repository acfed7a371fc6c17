#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_04
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/tail_grad_bisect.py 64 2>&1 | grep -v Warning | tee $O/tail_grad_bisect_64.txt | tail -70
