#!/bin/bash
# vectorised BN/GN backward: kernel tests, per-layer rates, trainable step
O=gpurun_out/r6_26; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bn_bwd or gn_bwd" 2>&1 | tail -3
timeout 300 python scripts/normbwdbench.py 2>&1 | tee $O/normbwd.txt
timeout 600 python -m pytest tests/test_trainable_encoders.py -q -x 2>&1 | tail -3
timeout 600 python bench.py --trainable-encoders --steps 10 --warmup 3 2>/dev/null | tee $O/bench_trainable.json | cut -c1-300
