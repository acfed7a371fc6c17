#!/bin/bash
# order / stream of the depth trunk in a training step: 4 alternating repetitions
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_28
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2 3 4; do
  echo "default            $(run)"
  echo "depth own stream   $(VLNCE_DEPTH_OWN_STREAM=1 run)"
  echo "depth first        $(VLNCE_TRAIN_ORDER=depth_first run)"
done | tee $O/depth_order.txt
VLNCE_TRAIN_ORDER=depth_first timeout 300 python scripts/backward_phase_probe.py 2>/dev/null | head -8
