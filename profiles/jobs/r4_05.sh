#!/bin/bash
# round 4, job 5: the step outside the trunks (tail_probe) + kernel stats of the cached-trunk step
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04e; mkdir -p $out
timeout 300 python scripts/tail_probe.py 2>&1 | grep -E "ms/step|phases" | tee $out/tail_probe.txt
