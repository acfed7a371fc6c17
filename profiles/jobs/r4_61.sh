#!/bin/bash
# host issue time vs GPU time per phase of the plain-loop step
O=$GRAFT_REPO_ROOT/gpurun_out/r04_61
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/host_vs_gpu_probe.py 2>/dev/null | tee $O/host_vs_gpu.txt
