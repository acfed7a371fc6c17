#!/bin/bash
# round 4, job 12: host blocked until the trunks are done (before the tail is enqueued)?
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04l; mkdir -p $out
for v in "" 1 "" 1; do
  echo "== VLNCE_EXP_SYNC_AFTER_TRUNK='$v'"
  VLNCE_EXP_SYNC_AFTER_TRUNK=$v timeout 300 python scripts/tail_probe.py 2>&1 | grep -E "full step|phases"
done | tee $out/sync_after_trunk.txt
