#!/bin/bash
# round 3, job 43: whole-act() graph: parity test, latency
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03an
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_policy_gpu.py -m gpu -x -q -p no:cacheprovider -k "whole_act_graph or graph_replay" 2>&1 | tail -15
for v in 0 1; do
  for n in 1 4 8; do echo "ACT_GRAPH=$v $(VLNCE_ACT_GRAPH=$v timeout 200 python scripts/act_profile.py --num-envs $n --iters 30 2>&1 | tail -1)"; done
done | tee $O/act.txt
