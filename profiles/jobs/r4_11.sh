#!/bin/bash
# round 4, job 11: does work enqueued behind the RGB trunk slow it down
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04k; mkdir -p $out
timeout 300 python scripts/overlap_probe3.py 2>&1 | grep "RGB trunk" | tee $out/overlap_probe3.txt
