#!/bin/bash
# rgb_kv of the 64-environment step on the bf16-plane kernels (threshold 1024 rows): A/B of the step
O=$GRAFT_REPO_ROOT/gpurun_out/r04_64
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or oracle" -p no:cacheprovider 2>&1 | tail -1
for v in 0 1 0 1; do
echo "== VLNCE_LINEAR_PLANES=$v"
VLNCE_LINEAR_PLANES=$v timeout 300 python scripts/host_vs_gpu_probe.py 2>/dev/null | grep "ms/step\|build_distribution\|backward"
done 2>&1 | tee $O/ab.txt
