#!/bin/bash
# conv_m3: ring of 3 vs 4 slabs on the depth trunk's layers (GPU-paced); m3 tests on the restructured loop
O=$GRAFT_REPO_ROOT/gpurun_out/r04_60
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "m3" -p no:cacheprovider 2>&1 | tail -2
for opt in "m3=1" "m3=4" "m3=1" "m3=4"; do
  echo "== options '$opt'"
  timeout 300 python scripts/convbench.py --set depth --mode train --backlog --iters 30 --opt "$opt" 2>/dev/null | grep -v "^layer"
done > $O/depth_m3_depth4.txt
grep "==\|trunk total\|3x3" $O/depth_m3_depth4.txt | cut -c1-90
