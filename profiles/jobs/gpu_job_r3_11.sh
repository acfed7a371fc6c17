#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03k
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/cross_kernel_floor.py > $O/cross_kernel_floor.txt 2>&1
cat $O/cross_kernel_floor.txt | grep -v amdgpu
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/step_profile.py --steps 12 --warmup 6 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/step_kernel_stats.md > /dev/null 2>&1
python scripts/rocpd_timeline.py $db 0 > $O/step_timeline.txt 2>&1
rm -rf $O/trace
head -45 $O/step_kernel_stats.md | cut -c1-160
cat $O/step_timeline.txt | cut -c1-400
