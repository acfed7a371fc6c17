#!/bin/bash
# per-layer sweep of tile / kernel choices on the RGB trunk's layers (train-mode prologue, GPU-paced)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_59
mkdir -p $O
cd $GRAFT_REPO_ROOT
{
for opt in "" "u3=2" "u3=3" "s3=0" "s3=2"; do
  echo "== 1x1 layers, options '$opt'"
  timeout 200 python scripts/convbench.py --set r50 --mode train --pro --backlog --iters 20 --only 1x1 --opt "$opt" 2>/dev/null | grep -v "^layer"
done
for t in 0 1 2 3 4 5 6; do
  echo "== 3x3 layers, p3_tile=$t"
  timeout 200 python scripts/convbench.py --set r50 --mode train --pro --backlog --iters 20 --only 3x3 --opt "p3_tile=$t" 2>/dev/null | grep -v "^layer"
done
} > $O/sweep.txt
wc -l $O/sweep.txt
