#!/bin/bash
# GEMM shapes + times: cached-feature update and the per-step update
O=$GRAFT_REPO_ROOT/gpurun_out/r04_42
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/gemm_log.py > $O/gemm_log_cached.txt 2>$O/err1.txt || tail -5 $O/err1.txt
cat $O/gemm_log_cached.txt
timeout 300 python scripts/gemm_log.py --step > $O/gemm_log_step.txt 2>$O/err2.txt || tail -5 $O/err2.txt
cat $O/gemm_log_step.txt
timeout 300 python scripts/bench_data_path.py --update-only 2>/dev/null | tail -1
