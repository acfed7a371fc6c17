#!/bin/bash
# round 3, job 25: instruction RNN backward with two accumulator chains
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03y
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k 'rnn or lstm or gru or instruction' 2>&1 | tail -2
timeout 300 python scripts/seqbench.py > $O/seqbench.txt 2>&1; grep 'rnn_seq' $O/seqbench.txt
