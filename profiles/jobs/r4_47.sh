#!/bin/bash
# conv_m3_kernel: kernel tests, then the depth trunk per layer (GPU-paced) with it off / default rule / forced
O=$GRAFT_REPO_ROOT/gpurun_out/r04_47
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "m3 or bn_sums_every or options" -p no:cacheprovider 2>&1 | tail -15
for opt in "m3=0" "m3=1" "m3=2"; do
  echo "== options '$opt'"
  timeout 300 python scripts/convbench.py --set depth --mode train --backlog --iters 30 --opt "$opt" 2>/dev/null
done > $O/depth_convbench_m3.txt
cat $O/depth_convbench_m3.txt
