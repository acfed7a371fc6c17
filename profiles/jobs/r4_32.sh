#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04zf; mkdir -p $out
timeout 200 python scripts/stem_time.py 2>&1 | grep "^stem7" | tee $out/stem7_times.txt
