#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "option or conv or forced or s3 or u3 or m3 or p3" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
