#!/bin/bash
# round 3, job 54: graph holders dropped on Module._apply: smoke + the graph-replay tests on the device
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03apply
( timeout 25 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout 40 python -m pytest tests/test_policy_gpu.py -x -q -p no:cacheprovider \
    -k "graph_replay_equals_eager or whole_act_graph or golden" 2>&1 | tail -4 ) > gpurun_out/r03apply/out.txt 2>&1
cat gpurun_out/r03apply/out.txt
