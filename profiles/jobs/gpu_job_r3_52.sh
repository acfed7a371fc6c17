#!/bin/bash
# round 3, job 52: conv_p3 tile 256x128 (spills, a scratch reload per chunk in the matrix loop) vs 128x128 on the 128-channel 3x3
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for t in 0 3 4; do
  echo "P3_TILE=$t: $(VLNCE_P3_TILE=$t timeout 100 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only l2_3x3_,l1_3x3 2>&1 | grep '^l[12]_' | awk '{printf "%s %s us  ", $1, $5}')"
done
