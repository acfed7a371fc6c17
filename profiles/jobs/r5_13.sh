#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_13
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/tail_lib_call_times.py 2>&1 | grep -v "Warn\|warn\|super()\|amdgpu.ids" > $O/tail_lib_call_times.txt
head -45 $O/tail_lib_call_times.txt
