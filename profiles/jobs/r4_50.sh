#!/bin/bash
# CU-masked streams: where do the workgroups land?
O=$GRAFT_REPO_ROOT/gpurun_out/r04_50
mkdir -p $O
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -shared -fPIC scripts/xcc_probe.hip -o /tmp/xcc_probe.so 2>&1 | tail -3
timeout 120 python scripts/cumask_probe.py 2>&1 | tee $O/cumask_probe.txt | tail -12
