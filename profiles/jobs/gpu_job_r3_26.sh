#!/bin/bash
# round-3 final evidence run: full GPU tests, bench line (+cpu baseline, fp32-only comparison), kernel-trace stats,
# PMC traffic passes, per-launch conv times, secondary configs, cached-feature data path, trainable encoders
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03z
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed" $O/gpu_tests.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -2 $O/conv_launch_times.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/kt -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/bench_kernel_stats.md 900 > /dev/null
head -12 $O/bench_kernel_stats.md | cut -c1-150
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --pmc-step > $O/pmc_$c.log 2>&1
  cd $GRAFT_REPO_ROOT
  db=$(find $O/pmc_$c -name "*.db" | head -1)
  python scripts/rocpd_pmc.py $db > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
  grep -n "segment" $O/pmc_$c.txt | head -6 | cut -c1-170
done
timeout 400 python scripts/bench_policies.py > $O/bench_other_policies.jsonl 2> $O/bench_other_policies.err
tail -3 $O/bench_other_policies.jsonl | cut -c1-250
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err
tail -1 $O/bench_data_path.json | cut -c1-400
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/ktu -- python $GRAFT_REPO_ROOT/scripts/bench_data_path.py --update-only --iters 6 > $O/ktu.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_one_step.py "$(find $O/ktu -name '*.db' | head -1)" > $O/cached_update_one_step.txt
rm -rf $O/ktu
head -8 $O/cached_update_one_step.txt | cut -c1-120
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
tail -c 2500 $O/bench_trainable.json | head -c 300
for n in 1 4 8; do timeout 200 python scripts/act_profile.py --num-envs $n --iters 30 2>/dev/null | tail -1; done | tee $O/act.txt
