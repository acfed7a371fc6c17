#!/bin/bash
# one zeroed arena for a trunk's weight gradients (vlnce_conv2d_wgrad accumulate): tests, A/B
O=gpurun_out/r6_32; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad or conv_backward" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  VLNCE_DW_ARENA=$v timeout 600 python bench.py --trainable-encoders --steps 10 --warmup 3 2>/dev/null | tee $O/bench_trainable_arena$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dw_arena=$v', d['ms_per_step'])"
done
