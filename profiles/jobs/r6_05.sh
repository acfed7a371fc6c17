#!/bin/bash
# round 6, job 5: where the time is after the fp16-plane change: phase probe of the CMA step, kernel
# stats of the Waypoint update and of the trainable-encoder step, seq2seq line, act profile
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_05
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/backward_phase_probe.py 2>/dev/null > $O/phase_probe.txt; tail -18 $O/phase_probe.txt
timeout 400 python bench.py --policy waypoint --steps 10 --warmup 3 > $O/bench_waypoint.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_waypoint.json | sed 's/^/waypoint /'
timeout 400 python bench.py --policy seq2seq --steps 20 > $O/bench_seq2seq.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_seq2seq.json | sed 's/^/seq2seq /'
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
grep -o '"ms_per_step": [0-9.]*' $O/bench_trainable.json | sed 's/^/trainable /'
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt_w -- python $GRAFT_REPO_ROOT/bench.py --policy waypoint --steps 5 --warmup 3 > $O/kt_w.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt_w -name '*.db' | head -1)" $O/waypoint_kernel_stats.md 900 > /dev/null
rm -rf $O/kt_w
head -40 $O/waypoint_kernel_stats.md
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt_t -- python $GRAFT_REPO_ROOT/bench.py --trainable-encoders --steps 5 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/kt_t.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt_t -name '*.db' | head -1)" $O/trainable_kernel_stats.md 900 > /dev/null
rm -rf $O/kt_t
head -40 $O/trainable_kernel_stats.md
