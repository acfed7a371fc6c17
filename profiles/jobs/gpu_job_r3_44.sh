#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 1 8; do VLNCE_ACT_GRAPH=1 timeout 200 python scripts/act_profile.py --num-envs $n --iters 50 2>&1 | tail -2; done
VLNCE_ACT_GRAPH=1 VLNCE_SIDE_STREAMS=0 timeout 200 python scripts/act_profile.py --num-envs 1 --iters 50 2>&1 | tail -2
