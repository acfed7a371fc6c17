#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_09
mkdir -p $O
cd $GRAFT_REPO_ROOT
VLNCE_TAIL_SPLIT=1 PYTHONFAULTHANDLER=1 timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 3 --warmup 4 > $O/out.txt 2> $O/err.txt
echo rc=$?
grep -v "Warning\|warn" $O/err.txt | tail -60
