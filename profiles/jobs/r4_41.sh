#!/bin/bash
# depth trunk conv shapes, GPU-paced, per kernel family
O=$GRAFT_REPO_ROOT/gpurun_out/r04_41
mkdir -p $O
cd $GRAFT_REPO_ROOT
for opt in "" "conv_math=0" "p3=1" "p3=0,u3=0,s3=0"; do
  echo "== options '$opt'"
  timeout 300 python scripts/convbench.py --set depth --mode train --pro --backlog --iters 30 --opt "$opt" 2>/dev/null
done > $O/depth_convbench.txt
cat $O/depth_convbench.txt
