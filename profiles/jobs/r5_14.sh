#!/bin/bash
# RNNLayerFn (instruction encoder layer as one autograd node, 2-call backward): parity + A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_14
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn or instruction" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "golden" 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
  VLNCE_RNN_LAYER=$v timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/rnn_layer=$v rep $rep /"
done; done
for v in 0 1; do
  echo "== VLNCE_RNN_LAYER=$v"
  VLNCE_RNN_LAYER=$v timeout 300 python scripts/host_vs_gpu_probe.py 2>/dev/null | tee $O/host_vs_gpu_rnn_layer$v.txt
done
