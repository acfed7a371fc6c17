#!/bin/bash
# round 3, job 41: the bench line of the final build (with the committed PMC traffic figure)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03zzz
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json
