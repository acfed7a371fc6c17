#!/bin/bash
# round 3, job 45: split-K convolutions through a workspace (one launch): parity, act() latency, policy tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ao
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "splitk or conv2d_fwd or bottleneck or block" 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_policy_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
for g in 0 1; do for w in 0 1; do
  for n in 1 4 8; do echo "ACT_GRAPH=$g SPLITK_WS=$w $(VLNCE_ACT_GRAPH=$g VLNCE_SPLITK_WS=$w timeout 200 python scripts/act_profile.py --num-envs $n --iters 40 2>&1 | tail -2 | tr '\n' ' ')"; done
done; done | tee $O/act.txt
