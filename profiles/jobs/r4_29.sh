#!/bin/bash
# round 4, job 29: kernel statistics of the Waypoint WDDPPO minibatch update (416 frames)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04zc; mkdir -p $out
O=$GRAFT_REPO_ROOT/$out
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/bench_policies.py --which waypoint --steps 6 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/waypoint_kernel_stats.md > /dev/null 2>&1
rm -rf $O/trace
head -45 $O/waypoint_kernel_stats.md | cut -c1-150
