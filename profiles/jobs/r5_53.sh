#!/bin/bash
# where the run-to-run spread of the bench comes from: slow steps or slow processes
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_53
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
  timeout 200 python scripts/step_jitter.py 200 2>/dev/null
done | tee $O/step_jitter.txt
