#!/bin/bash
# split-K for the dW GEMMs with a 1024-row reduction (rgb_kv at num_envs 64): gemm log of the step, tail probe
O=$GRAFT_REPO_ROOT/gpurun_out/r04_63
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/gemm_log.py --step > $O/gemm_log_step.txt 2>/dev/null; head -6 $O/gemm_log_step.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or linear" -p no:cacheprovider 2>&1 | tail -1
timeout 300 python scripts/host_vs_gpu_probe.py 2>/dev/null | tail -7
