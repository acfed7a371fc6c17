#!/bin/bash
# fused WDDPPO loss: parity + the Waypoint update
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_44
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "ppo" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "ppo or waypoint" 2>&1 | tail -2
for rep in 1 2; do timeout 400 python bench.py --policy waypoint --steps 10 --warmup 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/waypoint /'; done | tee $O/waypoint.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --policy waypoint --steps 6 --warmup 3 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/waypoint_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -40 $O/waypoint_kernel_stats.md | cut -c1-150
