#!/bin/bash
# round 6, job 10: dispatch experiments on the dual block ends (options only, one build)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_10
mkdir -p $O
cd $GRAFT_REPO_ROOT
L="l4_1x1_2048_512,l2_1x1_512_128,l1_1x1_256_64,l2_1x1_256_128,l3_1x1_1024_256,l3_1x1_512_256,l4_1x1_1024_512,l1_1x1_64_64"
for d in identity bn; do
for o in "" "u3=2" "p3=1" "p3=1,u3=0" "x3_tile=1" "x3_tile=2" "x3_tile=3" "x3_tile=4" "u3=0"; do
  echo "== dual $d opt [$o]"
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual $d --only $L ${o:+--opt $o} 2>&1 | grep -v "amdgpu\|^layer\|trunk total"
done; done > $O/dual_dispatch_experiments.txt 2>&1
cat $O/dual_dispatch_experiments.txt
for o in "" "u3=2" "u3=3" "p3=1,u3=0" "u3=0"; do
  echo "== single opt [$o]"
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --only 1x1 ${o:+--opt $o} 2>&1 | grep -v "amdgpu\|^layer"
done > $O/single_dispatch_experiments.txt 2>&1
cat $O/single_dispatch_experiments.txt
