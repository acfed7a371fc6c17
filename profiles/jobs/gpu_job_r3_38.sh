#!/bin/bash
# round 3, job 38: conv_s3 on fewer CUs: is the store phase bound per CU or by the chip's HBM?
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for g in 256 192 128 64; do
  echo "grid $g: $(VLNCE_S3_GRID=$g timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only l1_1x1_64_256,l2_1x1_128_512 2>&1 | grep '^l[12]_' | awk '{printf "%s %s us   ", $1, $5}')"
done
