#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_47
mkdir -p $O
cd $GRAFT_REPO_ROOT
d() { (cd $1 && timeout 300 python scripts/bench_data_path.py 2>/dev/null | grep -o '"cma_update_ms": [0-9.]*'); }
for rep in 1 2 3; do
  echo "round-4 final  $(d gpurun_tmp/r4)"
  echo "this tree      $(d .)"
done | tee $O/data_path_same_box.txt
