#!/bin/bash
# tail as two graph sections captured together (VLNCE_TAIL_SPLIT) A/B: plain-loop ms/step and phases
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_11
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sp in 0 1; do
  VLNCE_TAIL_SPLIT=$sp timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 > $O/bench_split${sp}_$rep.json 2>$O/err_split${sp}_$rep.txt
  echo "split $sp rep $rep: rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_split${sp}_$rep.json)"
done; done
for sp in 0 1; do
  echo "== VLNCE_TAIL_SPLIT=$sp"
  VLNCE_TAIL_SPLIT=$sp timeout 300 python scripts/host_vs_gpu_probe.py 2>/dev/null | tee $O/host_vs_gpu_split$sp.txt
done
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or pipeline or determin or twice" 2>&1 | tail -3
