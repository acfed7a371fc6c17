#!/bin/bash
# trainable-encoder step: kernel stats
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_45
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --trainable-encoders --steps 5 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/trainable_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -34 $O/trainable_kernel_stats.md | cut -c1-160
