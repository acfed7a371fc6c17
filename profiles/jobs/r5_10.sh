#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_10
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "2 " "3 " "1 1" "2 1" "3 1"; do
  set -- $v
  VLNCE_TAIL_SPLIT=$1 VLNCE_TAIL_CLONE=$2 timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 20 --warmup 4 > $O/out_$1_$2.txt 2> $O/err_$1_$2.txt
  echo "split=$1 clone=$2 rc=$? $(tail -c 300 $O/out_$1_$2.txt | grep -o '"ms_per_step": [0-9.]*')"
done
