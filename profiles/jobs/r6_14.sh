#!/bin/bash
# round 6, job 14: per-layer times with the operands in the Infinity Cache (one input buffer, as convbench
# always measured) against operands from HBM (8 input buffers in rotation), the plane kernels' default build
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_14
mkdir -p $O
cd $GRAFT_REPO_ROOT
for r in 1 8; do
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --rotate $r --iters 24 > $O/convbench_rotate$r.txt 2>&1
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual identity --rotate $r --iters 24 > $O/convbench_dual_rotate$r.txt 2>&1
done
paste <(awk '{print $1, $2, $3, $4, $5}' $O/convbench_rotate1.txt) <(awk '{print $5}' $O/convbench_rotate8.txt) | grep -v amdgpu
paste <(awk '{print $1, $2, $3, $4, $5}' $O/convbench_dual_rotate1.txt) <(awk '{print $5}' $O/convbench_dual_rotate8.txt) | grep -v amdgpu
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "frozen_weights or conv_p3_matches" 2>&1 | tail -3
