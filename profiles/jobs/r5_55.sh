#!/bin/bash
# issuing thread on the GPU's socket / the other socket / wherever the scheduler puts it
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_55
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  for m in none local remote; do
    BIND=$m timeout 200 python scripts/step_jitter.py 100 2>>$O/err.txt
  done
done | tee $O/step_jitter_socket.txt
tail -3 $O/err.txt
