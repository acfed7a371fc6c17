#!/bin/bash
# state encoders without the one-step slice (its autograd put a memcpy node into the tail's backward graph): phases + bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_43
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/backward_phase_probe.py 2>/dev/null | tee $O/phase_probe.txt | tail -9
for rep in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done | tee $O/bench.txt
