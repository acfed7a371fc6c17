#!/bin/bash
# per-launch conv times of both trunks with the BatchNorm-sums convolutions and the stem hooked
O=$GRAFT_REPO_ROOT/gpurun_out/r04zz
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>$O/clt.err
cat $O/conv_launch_times.txt
tail -3 $O/clt.err
