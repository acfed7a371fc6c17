#!/bin/bash
# parameter-gradient launches of the tail's linear layers as parallel branches of the backward graph: A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_18
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do
  VLNCE_GRAPH_BRANCHES=$v timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>$O/err_$v.txt | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/graph_branches=$v rep $rep /"
done; done
for v in 0 1; do
  echo "== VLNCE_GRAPH_BRANCHES=$v"
  VLNCE_GRAPH_BRANCHES=$v timeout 300 python scripts/backward_phase_probe.py 2>/dev/null | tee $O/phase_probe_branches$v.txt | grep -i "grad ready\|backward\|ms/step"
done
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or determin or twice" 2>&1 | tail -3
grep -v "Warn\|warn\|super()\|amdgpu" $O/err_1.txt | tail -5
