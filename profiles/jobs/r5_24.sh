#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_24
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do
  VLNCE_LINEAR_ROWS=$v timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/linear_rows=$v rep $rep /"
done; done | tee $O/bench_ab.txt
for v in 0 1; do
  echo "== VLNCE_LINEAR_ROWS=$v"
  VLNCE_LINEAR_ROWS=$v timeout 300 python scripts/tail_graph_time.py 2>/dev/null | head -1
done | tee $O/tail_graphs.txt
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden" 2>&1 | tail -2
