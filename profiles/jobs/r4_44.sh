#!/bin/bash
# distinct-instruction path: kernel + policy tests, cached-feature update time, gemm log
O=$GRAFT_REPO_ROOT/gpurun_out/r04_44
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or segment or action_head" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "distinct or rollout or golden" -p no:cacheprovider 2>&1 | tail -3
for v in 0 1; do
VLNCE_INSTR_DEDUP=$v timeout 300 python scripts/bench_data_path.py --update-only 2>/dev/null | tail -1
done
timeout 300 python scripts/bench_data_path.py --update-only 2>/dev/null | tail -1
timeout 300 python scripts/gemm_log.py > $O/gemm_log_cached.txt 2>$O/err1.txt || tail -5 $O/err1.txt
head -12 $O/gemm_log_cached.txt
