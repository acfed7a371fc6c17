#!/bin/bash
# persistent kernels on fewer than all CUs (room for the side streams' launches): plain-loop ms/step
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_17
mkdir -p $O
cd $GRAFT_REPO_ROOT
for c in 0 248 240 232 224 0 240; do
  VLNCE_PERSISTENT_CUS=$c timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('persistent_cus=$c', d['ms_per_step'], 'conv', d['roofline']['kernel_ms_per_step'], 'eager trunks', d['roofline']['eager_single_stream_trunks_ms'])"
done | tee $O/persistent_cus.txt
