#!/bin/bash
# round 4, job 16: graphed instruction recurrence x side streams: which combination is slow
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04p; mkdir -p $out
for ss in 1 0; do for ig in 0 1; do
  echo "== SIDE_STREAMS=$ss INSTR_GRAPH=$ig"
  VLNCE_SIDE_STREAMS=$ss VLNCE_INSTR_GRAPH=$ig timeout 300 python scripts/tail_probe.py 2>&1 | grep -E "full step|phases"
done; done | tee $out/instr_graph_streams.txt
