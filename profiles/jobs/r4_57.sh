#!/bin/bash
# kernel statistics of the Waypoint WDDPPO update on the current build
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_57
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/bench_policies.py --which waypoint --steps 10 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/waypoint_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -40 $O/waypoint_kernel_stats.md | cut -c1-170
