#!/bin/bash
# round 4, job 10: phase stamps with and without the validation sync
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04j; mkdir -p $out
for v in "" 1; do
  echo "== VLNCE_EXP_NOVALIDATE='$v'"
  VLNCE_EXP_NOVALIDATE=$v timeout 300 python scripts/tail_probe.py 2>&1 | grep -E "ms/step|phases"
done | tee $out/tail_probe_novalidate.txt
