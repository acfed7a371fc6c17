#!/bin/bash
# round 3, job 36: conv_s3 with a staggered start of every second workgroup
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03aj
mkdir -p $O
cd $GRAFT_REPO_ROOT
ONLY=l1_1x1_64_256,l2_1x1_128_512
for st in 0 30 60 90 130 200; do
  VLNCE_S3_STAGGER=$st timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only $ONLY > $O/convbench_stagger_$st.txt 2>&1
  echo "stagger $st: $(grep '^l[12]_' $O/convbench_stagger_$st.txt | awk '{printf "%s %s us   ", $1, $5}')"
done
