#!/bin/bash
# round-2 job 2: persistent conv kernel -- correctness, then convbench / bench A-B over the knobs
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q ) > $O/kernel_tests.log 2>&1
tail -5 $O/kernel_tests.log
if ! grep -q " passed" $O/kernel_tests.log || grep -q "failed" $O/kernel_tests.log; then echo KERNEL TESTS FAILED; tail -60 $O/kernel_tests.log; fi
for cfg in "VLNCE_IGEMM_NO_PERSIST=1" "VLNCE_PK_TILES=1" "VLNCE_PK_TILES=2" "VLNCE_PK_TILES=4" "VLNCE_PK_TILES=8" "VLNCE_PK_TILES=100000"; do
  echo "== $cfg"
  env $cfg timeout 300 python scripts/convbench.py --mode train > $O/convbench_$cfg.txt 2>&1
  tail -27 $O/convbench_$cfg.txt | awk '{printf "%s %s %s | ", $1, $5, $6} END {print ""}'
done
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
for cfg in "VLNCE_IGEMM_NO_PERSIST=1" "VLNCE_PK_TILES=4" "VLNCE_PK_TILES=8" "VLNCE_PK_TILES=100000"; do
  echo "== bench $cfg"
  env $cfg timeout 600 python bench.py --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$cfg.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["config"]["no_pipeline_ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["config"]["act_latency_ms_by_num_envs"])
except Exception as e:
    print("bench failed", e)
PY
done
