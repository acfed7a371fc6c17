#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04v; mkdir -p $out
timeout 600 python tests/rccl_single_rank_check.py > $out/rccl_single_rank.txt 2>&1; echo "rccl check rc=$?"; grep -E "trainable|OK|Error|assert" $out/rccl_single_rank.txt | tail -5
