#!/bin/bash
# the two trunks on disjoint CUs (CU-masked streams) vs shared
O=$GRAFT_REPO_ROOT/gpurun_out/r04_51
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python scripts/cumask_overlap_probe.py 2>$O/err.txt | tee $O/cumask_overlap.txt
tail -5 $O/err.txt
