#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "VLNCE_IGEMM_STAGGER=0" "VLNCE_IGEMM_STAGGER=4" "VLNCE_IGEMM_STAGGER=8" "VLNCE_X=default" "VLNCE_IGEMM_STAGGER=16"; do
  echo "== $cfg"
  env $cfg timeout 300 python scripts/convbench.py --mode train > $O/convbench_$cfg.txt 2>&1
  tail -27 $O/convbench_$cfg.txt | awk '{printf "%s %s %s | ", $1, $5, $6} END {print ""}'
done
