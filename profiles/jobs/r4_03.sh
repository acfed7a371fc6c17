#!/bin/bash
# round 4, job 3: is it the null stream?  the same probe with the step on a pool stream, and with 8 hardware queues
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c; mkdir -p $out
echo "== pool main stream" | tee $out/overlap_probe_main_stream.txt
timeout 300 python scripts/overlap_probe.py --main-stream 2>&1 | grep -E "order|host:|main work" | tee -a $out/overlap_probe_main_stream.txt
echo "== GPU_MAX_HW_QUEUES=8, null stream" | tee -a $out/overlap_probe_main_stream.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/overlap_probe.py 2>&1 | grep -E "order|host:|main work" | tee -a $out/overlap_probe_main_stream.txt
