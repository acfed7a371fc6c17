#!/bin/bash
# round-3 final evidence run on the final build: full GPU tests, smoke, bench line, kernel stats, PMC traffic,
# per-launch conv times, secondary configs
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03zz
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed" $O/gpu_tests.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --pmc-step > $O/pmc_$c.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_pmc.py "$(find $O/pmc_$c -name '*.db' | head -1)" > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
python scripts/pmc_traffic_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt profiles/r03_pmc_traffic.json "profiles/r03_zz_pmc_fetch_size.txt, r03_zz_pmc_write_size.txt" > /dev/null
cp profiles/r03_pmc_traffic.json $O/r03_pmc_traffic.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -2 $O/conv_launch_times.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
timeout 400 python scripts/bench_policies.py > $O/bench_other_policies.jsonl 2> $O/bench_other_policies.err
tail -2 $O/bench_other_policies.jsonl | cut -c1-200
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err
tail -1 $O/bench_data_path.json | cut -c-1 > /dev/null; tail -c 200 $O/bench_data_path.json
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_trainable.json').read().strip().split('\n')[-1]); print('trainable', d['value'], d['ms_per_step'])"
