#!/bin/bash
# round 3, job 19: one-launch GRU rollout: parity tests, cached-feature update A/B, kernel breakdown
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03s
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "rollout or rnn or lstm or gru or instruction" ) > $O/rollout_tests.log 2>&1
grep -n "passed\|failed\|Error" $O/rollout_tests.log | tail -5
( time timeout 900 python -m pytest tests/test_policy_gpu.py tests/test_policy_sizes_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/policy_tests.log 2>&1
grep -n "passed\|failed\|Error" $O/policy_tests.log | tail -5
for v in 1 0; do
  VLNCE_GRU_ROLLOUT=$v timeout 300 python scripts/bench_data_path.py --update-only --iters 30 > $O/update_rollout_$v.json 2> $O/update_rollout_$v.err
  echo "VLNCE_GRU_ROLLOUT=$v: $(tail -1 $O/update_rollout_$v.json)"
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/bench_data_path.py --update-only --iters 6 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/kt -name "*.db" | head -1)
python scripts/rocpd_one_step.py "$db" > $O/cached_update_one_step.txt
rm -rf $O/kt
head -30 $O/cached_update_one_step.txt | cut -c1-140
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err
tail -1 $O/bench_data_path.json | cut -c1-500
