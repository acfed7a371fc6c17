#!/bin/bash
# one-launch weight preparation for trainable trunks: kernel test, trainable tests, A/B bench
O=gpurun_out/r6_27; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "prepare_weights or conv_backward or wgrad" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  VLNCE_WEIGHT_PREP=$v timeout 600 python bench.py --trainable-encoders --steps 10 --warmup 3 2>/dev/null | tee $O/bench_trainable_prep$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prep=$v', d['ms_per_step'])"
done
