#!/bin/bash
# side-stream priority and the depth trunk on its own stream, with the one-node instruction layer
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_20
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2; do
  echo "default            $(run)"
  echo "side prio 0,0,0    $(VLNCE_SIDE_PRIORITY=0,0,0 run)"
  echo "side prio 1,1,1    $(VLNCE_SIDE_PRIORITY=1,1,1 run)"
  echo "depth own stream   $(VLNCE_DEPTH_OWN_STREAM=1 run)"
  echo "depth own, prio 0  $(VLNCE_DEPTH_OWN_STREAM=1 VLNCE_SIDE_PRIORITY=0,0,0 run)"
done | tee $O/side_stream_variants.txt
