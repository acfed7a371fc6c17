#!/bin/bash
# side-stream priorities with the depth trunk on its own stream (stream order: instruction, run-ahead RGB, depth)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_52
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2 3 4; do
  echo "default -1,-1,-1   $(run)"
  echo "depth low -1,-1,0  $(VLNCE_SIDE_PRIORITY=-1,-1,0 run)"
  echo "all normal 0,0,0   $(VLNCE_SIDE_PRIORITY=0,0,0 run)"
done | tee $O/priorities.txt
