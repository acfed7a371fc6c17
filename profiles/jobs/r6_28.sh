#!/bin/bash
# block-input gradient sum in the dgrad epilogue + vectorised max-pool backward: tests, A/B
O=gpurun_out/r6_28; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "maxpool or conv_backward or bn_bwd or gn_bwd" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  VLNCE_DGRAD_ADD=$v timeout 600 python bench.py --trainable-encoders --steps 10 --warmup 3 2>/dev/null | tee $O/bench_trainable_add$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dgrad_add=$v', d['ms_per_step'])"
done
