#!/bin/bash
# full GPU tier + smoke + default bench on the state after the trainable-encoder work
O=gpurun_out/r6_31; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py 2>/dev/null | tee $O/bench_default.json | cut -c1-400
timeout 600 python bench.py --trainable-encoders 2>/dev/null | tee $O/bench_trainable.json | cut -c1-300
