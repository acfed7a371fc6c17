#!/bin/bash
# round 6, job 21: wgrad_x6_kernel (weight gradient on the 16-bit pipe): kernel tests, trainable-encoder tests,
# the trainable-encoder step with it and with the fp32-MFMA kernel (VLNCE_WGRAD_TILE=1), alternating
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_21
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad or conv_backward" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -5
for rep in 1 2; do for m in 64 1; do
  VLNCE_WGRAD_TILE=$m timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable_$m.json 2>/dev/null
  echo "VLNCE_WGRAD_TILE=$m $(grep -o '"ms_per_step": [0-9.]*' $O/bench_trainable_$m.json)"
done; done | tee $O/trainable_ab.txt
