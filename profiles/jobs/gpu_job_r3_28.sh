#!/bin/bash
# round 3, job 28: instruction RNN at H=128 with 4 waves (1 per SIMD, weights in VGPRs+AGPRs) instead of 8
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_nw4.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k 'rnn or lstm or gru or instruction' 2>&1 | tail -2
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_nw4.so timeout 300 python scripts/seqbench.py > $O/seqbench_nw4.txt 2>&1; grep 'rnn_seq' $O/seqbench_nw4.txt
