#!/bin/bash
# the two directions' recurrent-layer parameter gradients on two streams: parity + A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_29
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn or instruction" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "golden" 2>&1 | tail -2
run() { timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2 3 4; do
  echo "one stream   $(VLNCE_RNN_WGRAD_STREAMS=0 run)"
  echo "two streams  $(run)"
done | tee $O/wgrad_streams.txt
