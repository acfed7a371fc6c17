#!/bin/bash
# round 4, job 4: does a side stream progress while the RGB trunk runs (event stamps, no profiler)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04d; mkdir -p $out
timeout 300 python scripts/overlap_probe2.py 2>&1 | grep -E "graphs|main done" | tee $out/overlap_probe2.txt
VLNCE_HIP_GRAPHS=0 timeout 300 python scripts/overlap_probe2.py 2>&1 | grep -E "graphs|main done" | tee -a $out/overlap_probe2.txt
