#!/bin/bash
# GroupNorm finalize + apply in one launch (depth trunk): parity + A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_31
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "group_norm" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or baseline_shape" 2>&1 | tail -2
run() { timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2 3 4; do
  echo "finalize + apply (2 launches)  $(VLNCE_GN_TILES_APPLY=0 run)"
  echo "one launch                     $(run)"
done | tee $O/gn_tiles_apply_ab.txt
for v in 0 1; do
  echo "waypoint VLNCE_GN_TILES_APPLY=$v $(VLNCE_GN_TILES_APPLY=$v timeout 300 python bench.py --policy waypoint --steps 10 --warmup 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done | tee -a $O/gn_tiles_apply_ab.txt
