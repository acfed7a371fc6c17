#!/bin/bash
# round 4, job 14: instruction encoder's recurrence as graphed fwd+bwd: tests + A/B + rccl check
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04n; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_policy_gpu.py -x -q -p no:cacheprovider -k "instruction or golden or rccl or graph or cma or seq2seq or waypoint or policy" > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $out/tests.txt
for v in 0 1 0 1; do
  echo "INSTR_GRAPH=$v: $(VLNCE_INSTR_GRAPH=$v timeout 200 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done | tee $out/instr_graph_ab.txt
VLNCE_INSTR_GRAPH=1 timeout 300 python scripts/tail_probe.py 2>&1 | grep -E "ms/step|phases" | tee $out/tail_probe.txt
