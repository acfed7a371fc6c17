#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn or rollout or gru or lstm" ) > $O/rnn_tests.log 2>&1
tail -3 $O/rnn_tests.log
if grep -q "failed\|error" $O/rnn_tests.log; then tail -100 $O/rnn_tests.log | head -80; fi
for cfg in "VLNCE_RNN_STEP_FUSED=0 VLNCE_INSTR_DEDUP=1" "VLNCE_RNN_STEP_FUSED=1 VLNCE_INSTR_DEDUP=1"; do
  env $cfg timeout 300 python scripts/bench_data_path.py --update-only 2>/dev/null | tail -1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/bench_data_path.py --update-only --iters 5 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/kt -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/kernel_stats_cached_update.md 2000 > /dev/null
head -24 $O/kernel_stats_cached_update.md | cut -c1-150
rm -rf $O/kt
