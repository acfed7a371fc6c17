#!/bin/bash
# round 3, job 20: GRU rollout with tagged-pair exchange: tests, per-step cost (also all workgroups on one XCD,
# instruction LSTM with 2-step prefetch), cached-feature update, launch list of one act() at num_envs=1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03t
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "rollout or rnn or lstm or gru or instruction" ) > $O/rollout_tests.log 2>&1
grep -n "passed\|failed\|Error" $O/rollout_tests.log | tail -5
timeout 300 python scripts/seqbench.py > $O/seqbench.txt 2>&1; grep -v amdgpu $O/seqbench.txt
VLNCE_ROLLOUT_ONE_XCD=1 timeout 300 python scripts/seqbench.py > $O/seqbench_one_xcd.txt 2>&1; grep "GRU" $O/seqbench_one_xcd.txt
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_pf2.so timeout 300 python scripts/seqbench.py > $O/seqbench_pf2.txt 2>&1; grep "rnn_seq" $O/seqbench_pf2.txt
for x in 0 1; do
  VLNCE_ROLLOUT_ONE_XCD=$x timeout 300 python scripts/bench_data_path.py --update-only --iters 30 > $O/update_one_xcd_$x.json 2> $O/update_one_xcd_$x.err
  echo "ONE_XCD=$x: $(tail -1 $O/update_one_xcd_$x.json)"
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/bench_data_path.py --update-only --iters 6 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_one_step.py "$(find $O/kt -name '*.db' | head -1)" > $O/cached_update_one_step.txt
rm -rf $O/kt
head -12 $O/cached_update_one_step.txt | cut -c1-140; tail -1 $O/cached_update_one_step.txt
timeout 200 python scripts/act_profile.py --num-envs 1 --iters 30 > $O/act_n1.txt 2>&1; tail -1 $O/act_n1.txt
timeout 200 python scripts/act_profile.py --num-envs 1 --iters 30 --sync > $O/act_n1_sync.txt 2>&1; tail -1 $O/act_n1_sync.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kta -- python $GRAFT_REPO_ROOT/scripts/act_profile.py --num-envs 1 --iters 8 --sync > $O/kta.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_act.py "$(find $O/kta -name '*.db' | head -1)" list > $O/act_one_call.txt 2>&1
rm -rf $O/kta
head -40 $O/act_one_call.txt | cut -c1-130
