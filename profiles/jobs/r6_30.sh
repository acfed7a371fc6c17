#!/bin/bash
# weight gradient on fp16 planes: tests, per-layer, A/B
O=gpurun_out/r6_30; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad or conv_backward or power_of_two" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -5
timeout 300 python scripts/wgradbench.py > $O/wgrad_bf16.txt 2>&1; tail -1 $O/wgrad_bf16.txt
timeout 300 python scripts/wgradbench.py --f16 > $O/wgrad_f16.txt 2>&1; tail -1 $O/wgrad_f16.txt
for v in f16 bf16 f16 bf16; do
  VLNCE_GRAD_PLANES=$v timeout 600 python bench.py --trainable-encoders --steps 10 --warmup 3 2>/dev/null | tee $O/bench_trainable_grad_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grad_planes=$v', d['ms_per_step'])"
done
