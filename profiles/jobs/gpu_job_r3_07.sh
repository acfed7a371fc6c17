#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "full gpu suite rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
cat $O/bench.json | cut -c1-1800
grep "plain loop\|conv attribution\|timed region" $O/bench.err
