#!/bin/bash
# round 3, job 21: hardware exp2/rcp gate math in the instruction RNN, depth trunk first in act():
# full GPU tests, per-step cost + shader clock of the sequence kernels, cached-feature update, act(), bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03u
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed\|Error" $O/gpu_tests.log | tail -5
timeout 300 python scripts/seqbench.py > $O/seqbench.txt 2>&1; grep -v amdgpu $O/seqbench.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc -- python $GRAFT_REPO_ROOT/scripts/seqbench.py --reps 3 > $O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/pmc -name "*.db" | head -1)
python scripts/rocpd_pmc_layers.py $db 6 rnn_seq > $O/pmc_rnn_seq.txt 2>&1
python scripts/rocpd_pmc_layers.py $db 6 gru_rollout > $O/pmc_gru_rollout.txt 2>&1
rm -rf $O/pmc
cut -c1-200 $O/pmc_rnn_seq.txt | head -12; cut -c1-200 $O/pmc_gru_rollout.txt | head -12
timeout 300 python scripts/bench_data_path.py --update-only --iters 30 > $O/update.json 2> $O/update.err; tail -1 $O/update.json
for n in 1 4 8; do timeout 200 python scripts/act_profile.py --num-envs $n --iters 30 2>/dev/null | tail -1; done | tee $O/act.txt
timeout 200 python scripts/act_profile.py --num-envs 1 --iters 30 --sync 2>/dev/null | tail -1 | tee -a $O/act.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kta -- python $GRAFT_REPO_ROOT/scripts/act_profile.py --num-envs 1 --iters 8 --sync > $O/kta.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_act.py "$(find $O/kta -name '*.db' | head -1)" > $O/act_one_call.txt 2>&1
rm -rf $O/kta
head -12 $O/act_one_call.txt | cut -c1-130
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03u/bench.json").read().strip().split("\n")[-1])
c=d["config"]; r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"ahead",c.get("encode_ahead_ms_per_step"),"act",c.get("act_latency_ms_by_num_envs"))
print("conv ms",r["kernel_ms_per_step"],"frac",r["frac"],"bf16",r["bf16_pipe"]["frac"],r["bf16_pipe"]["by_kernel"])
PY
