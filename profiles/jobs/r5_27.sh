#!/bin/bash
# fused-KV attention (AttnKVFn): parity tests + plain loop
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_27
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden" 2>&1 | tail -2
for rep in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/attn_kv rep $rep /"
done | tee $O/bench.txt
timeout 300 python scripts/tail_graph_time.py 2>/dev/null | head -1 | tee -a $O/bench.txt
timeout 300 python scripts/backward_phase_probe.py 2>/dev/null | tee $O/phase_probe.txt | tail -14
