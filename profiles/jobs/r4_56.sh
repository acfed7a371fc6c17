#!/bin/bash
# large linears on the bf16-plane kernels: test, cached-feature update and Waypoint update with / without
O=$GRAFT_REPO_ROOT/gpurun_out/r04_56
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "large_linear or linear" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or rollout or distinct" -p no:cacheprovider 2>&1 | tail -2
for v in 0 1 0 1; do
echo "== VLNCE_LINEAR_PLANES=$v"
VLNCE_LINEAR_PLANES=$v timeout 300 python scripts/bench_data_path.py --update-only 2>/dev/null | tail -1
VLNCE_LINEAR_PLANES=$v timeout 400 python scripts/bench_policies.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:40] if isinstance(d['config'],dict) else '', d.get('ms_per_step'))"
done 2>&1 | tee $O/ab.txt
