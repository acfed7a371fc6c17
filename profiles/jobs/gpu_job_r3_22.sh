#!/bin/bash
# round 3, job 22: kernel breakdown of ONE trainable-encoder step (plain loop)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/step_profile.py --trainable-encoders --steps 4 --warmup 3 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $O/kt.log
python scripts/rocpd_one_step.py "$(find $O/kt -name '*.db' | head -1)" > $O/trainable_one_step.txt
rm -rf $O/kt
head -45 $O/trainable_one_step.txt | cut -c1-150
