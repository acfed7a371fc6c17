#!/bin/bash
# round 3, job 46: the committed final state: full GPU tests, smoke, bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03final
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed" $O/gpu_tests.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json
