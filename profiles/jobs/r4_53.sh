#!/bin/bash
# stem7 with the previous tile's stores under the current tile's MFMAs: tests + times
O=$GRAFT_REPO_ROOT/gpurun_out/r04_53
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "stem7" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python scripts/stem_time.py 2>/dev/null | tee $O/stem7_times.txt
