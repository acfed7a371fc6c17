#!/bin/bash
# round-2 evidence run: full GPU tests, bench line, kernel-trace stats of bench, PMC traffic passes
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02zz
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed" $O/gpu_tests.log | tail -2
if grep -q "failed\|error" $O/gpu_tests.log; then tail -120 $O/gpu_tests.log | head -100; fi
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
# kernel trace of the same command (short)
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
db=$(find $O/kt -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/bench_kernel_stats.md 900 > /dev/null
head -24 $O/bench_kernel_stats.md | cut -c1-160
rm -rf $O/kt
# HBM traffic: separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python bench.py --pmc-step > $O/pmc_$c.log 2>&1
  db=$(find $O/pmc_$c -name "*.db" | head -1)
  python scripts/rocpd_pmc.py $db > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
  grep -n "segment\|conv_x3\|igemm" $O/pmc_$c.txt | head -30 | cut -c1-170
done
# other BASELINE.json configurations and the cached-feature data path
timeout 400 python scripts/bench_policies.py > $O/bench_other_policies.jsonl 2> $O/bench_other_policies.err
tail -3 $O/bench_other_policies.jsonl | cut -c1-250
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err
tail -1 $O/bench_data_path.json | cut -c1-600
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
tail -c 600 $O/bench_trainable.json | head -c 300
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -3 $O/conv_launch_times.txt
timeout 300 python scripts/trunkbench.py 2>/dev/null | tail -6
