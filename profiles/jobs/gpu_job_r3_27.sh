#!/bin/bash
# round 3, job 27: phase timers inside the instruction RNN kernels (debug build)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03aa
mkdir -p $O
cd $GRAFT_REPO_ROOT
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_seqtime.so timeout 300 python scripts/seqbench.py --reps 1 > $O/seqtime.txt 2>&1
grep "rnn_seq" $O/seqtime.txt | sort | uniq -c | sort -rn | head -30
