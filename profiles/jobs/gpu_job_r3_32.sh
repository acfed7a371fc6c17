#!/bin/bash
# round 3, job 32: final build: forced conv_u3 tests, the bench line with the per-launch floor block
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03af
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "u3 or conv_p3" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 900 $O/bench.json
