#!/bin/bash
# kernel sequence of one plain-loop step (rocprofv3 kernel trace) after the skinny-linear kernels
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_26
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/step_profile.py --steps 10 --warmup 6 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_seq.py $db > $O/step_kernel_sequence.txt 2>&1
python scripts/rocpd_timeline.py $db 0 > $O/step_timeline.txt 2>&1
rm -rf $O/trace
head -3 $O/step_kernel_sequence.txt; head -8 $O/step_timeline.txt
