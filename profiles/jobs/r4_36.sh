#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04zj; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "stem7" 2>&1 | tail -2
timeout 200 python scripts/stem_time.py 2>&1 | grep '^stem7' | tee $out/stem7_times.txt
