#!/bin/bash
# round 4, job 15: kernel sequence of a trunk-less step with the graphed instruction recurrence
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04o; mkdir -p $out
O=$GRAFT_REPO_ROOT/$out
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/tail_profile.py > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_seq.py $db > $O/tail_step_sequence_instr_graph.txt 2>&1
rm -rf $O/trace
head -3 $O/tail_step_sequence_instr_graph.txt
