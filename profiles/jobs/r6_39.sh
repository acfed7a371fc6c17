#!/bin/bash
# trainable trunks: the RGB stem's forward through stem7 (default) vs the fp32-MFMA convolution (VLNCE_STEM7=0)
O=gpurun_out/r6_39; mkdir -p $O
timeout 600 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  VLNCE_STEM7=$v timeout 600 python bench.py --trainable-encoders --steps 10 --warmup 3 2>/dev/null | tee $O/bench_trainable_stem7_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stem7=$v', d['ms_per_step'])"
done
