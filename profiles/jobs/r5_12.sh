#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_12
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/tail_graph_time.py 2>&1 | grep -v "Warn\|warn\|super()\|amdgpu.ids" | tee $O/tail_graph_time.txt
for sk in 0 1; do
  VLNCE_IGEMM_NO_SPLITK=$sk timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/no_splitk=$sk /"
done
for sk in 0 1; do
  VLNCE_IGEMM_NO_SPLITK=$sk timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/no_splitk=$sk /"
done
