#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02s
mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q ) > $O/kt.log 2>&1
tail -5 $O/kt.log
timeout 600 python scripts/convbench.py --n 64 --iters 20 > $O/conv_x3.log 2>&1
timeout 600 python scripts/convbench.py --n 64 --iters 20 --mode train > $O/convtrain_x3.log 2>&1
paste <(awk '{print $1, $2,$3,$4, $(NF-2), $(NF-1)}' $O/conv_x3.log) <(awk '{print $(NF-2), $(NF-1)}' $O/convtrain_x3.log) | tail -26
timeout 300 python scripts/trunkbench.py 2>&1 | tail -3
