#!/bin/bash
# skinny-linear kernels: parity, then A/B of the plain loop and the tail graphs alone
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_21
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden" 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
  VLNCE_LINEAR_ROWS=$v timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/linear_rows=$v rep $rep /"
done; done
for v in 0 1; do
  echo "== VLNCE_LINEAR_ROWS=$v"
  VLNCE_LINEAR_ROWS=$v timeout 300 python scripts/tail_graph_time.py 2>/dev/null | head -2
  VLNCE_LINEAR_ROWS=$v timeout 300 python scripts/backward_phase_probe.py 2>/dev/null | tee $O/phase_probe_linear_rows$v.txt | grep -i "replay\|grad ready\|backward\|ms/step\|Adam"
done
