#!/bin/bash
# kernel statistics of the bench command and per-launch conv times on the round's last commit
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_66
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -12 $O/bench_kernel_stats.md | cut -c1-150
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -3 $O/conv_launch_times.txt
