#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_30
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/backward_phase_probe.py 2>/dev/null | tee $O/phase_probe.txt
