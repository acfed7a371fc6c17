#!/bin/bash
# rank threads on the GPU's socket vs on ONE L3 domain (CCD) of that socket
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_59
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for m in local l3; do
    BIND=$m timeout 100 python scripts/step_jitter.py 100 2>>$O/err.txt
  done
done | tee $O/step_jitter_l3.txt
tail -3 $O/err.txt | cut -c1-300
