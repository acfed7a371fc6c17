#!/bin/bash
# sanity of the final library: conv kernel tests, default bench line
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -n 2
timeout 900 python bench.py 2>/dev/null | tee gpurun_out/r6_42_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
