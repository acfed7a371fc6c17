#!/bin/bash
# round 4, job 2: where is the host while the RGB trunk runs (scripts/overlap_probe.py), the
# branch issue orders, and a timeline of the side_first order
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04b; mkdir -p $out
timeout 300 python scripts/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/overlap_probe.txt
O=$GRAFT_REPO_ROOT/$out
cd /tmp
VLNCE_TRAIN_ORDER=side_first timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/step_profile.py --steps 12 --warmup 6 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_timeline.py $db 0 > $O/step_timeline_side_first.txt 2>&1
rm -rf $O/trace
cut -c1-200 $O/step_timeline_side_first.txt | head -12
