#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
if grep -q "failed\|error" $O/gpu_tests.log; then tail -120 $O/gpu_tests.log | head -100; fi
timeout 600 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err
tail -3 $O/bench_data_path.json | cut -c1-1500
tail -5 $O/bench_data_path.err
timeout 600 python scripts/bench_policies.py --steps 10 > $O/bench_policies.txt 2>&1
tail -3 $O/bench_policies.txt
