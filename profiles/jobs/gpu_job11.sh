#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "group_norm or gn or conv" 2>&1 | tail -3
for cfg in "VLNCE_GN_FUSED=0" "VLNCE_GN_FUSED=1"; do
  echo "== $cfg"; env $cfg timeout 300 python scripts/trunkbench.py 2>/dev/null | grep depth
done
