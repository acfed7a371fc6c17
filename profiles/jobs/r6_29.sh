#!/bin/bash
# data gradient on fp16 planes with the norm-backward power of two: tests, A/B
O=gpurun_out/r6_29; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "power_of_two or bn_bwd or gn_bwd or conv_backward" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -5
for v in f16 bf16 f16 bf16; do
  VLNCE_GRAD_PLANES=$v timeout 600 python bench.py --trainable-encoders --steps 10 --warmup 3 2>/dev/null | tee $O/bench_trainable_grad_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grad_planes=$v', d['ms_per_step'])"
done
