#!/bin/bash
# round 3, job 23: instruction RNN with the transposed accumulator assignment (16-byte accesses), one-sync
# instruction de-duplication, 128x128 weight-gradient tiles: tests, per-step cost, update benches, bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03w
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed\|Error" $O/gpu_tests.log | tail -5
timeout 300 python scripts/seqbench.py > $O/seqbench.txt 2>&1; grep "rnn_seq" $O/seqbench.txt
timeout 300 python scripts/bench_data_path.py --update-only --iters 30 > $O/update.json 2> $O/update.err; tail -1 $O/update.json
for t in 64 128; do
  VLNCE_WGRAD_TILE=$t timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable_$t.json 2>/dev/null
  python -c "
import json,sys
d=json.loads(open('$O/bench_trainable_$t.json').read().strip().split('\n')[-1]); print('wgrad tile $t: trainable', d['value'], d['ms_per_step'])"
done
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03w/bench.json").read().strip().split("\n")[-1])
c=d["config"]; r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"ahead",c.get("encode_ahead_ms_per_step"),"act",c.get("act_latency_ms_by_num_envs"))
print("conv ms",r["kernel_ms_per_step"],"frac",r["frac"],"bf16",r["bf16_pipe"]["frac"])
PY
