#!/bin/bash
# final state: kernel stats of the bench command (rocprofv3 --kernel-trace --stats), conv launch table, policy lines
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_37
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -14 $O/bench_kernel_stats.md | cut -c1-150
timeout 600 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>&1; head -3 $O/conv_launch_times.txt
timeout 600 python bench.py --policy waypoint --steps 10 --warmup 3 > $O/bench_waypoint.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_waypoint.json | sed 's/^/waypoint /'
timeout 600 python bench.py --policy seq2seq --steps 10 --warmup 3 > $O/bench_seq2seq.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_seq2seq.json | sed 's/^/seq2seq /'
