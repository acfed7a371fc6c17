#!/bin/bash
# round 4, job 9: step time vs where the host synchronises
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04i; mkdir -p $out
timeout 300 python scripts/sync_probe.py > $out/sync_probe_full.txt 2>&1
grep "ms/step" $out/sync_probe_full.txt | tee $out/sync_probe.txt
tail -5 $out/sync_probe_full.txt
