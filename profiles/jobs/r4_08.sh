#!/bin/bash
# round 4, job 8: what the Categorical's argument validation (a host sync after the tail forward) costs a step
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04h; mkdir -p $out
for v in "" 1 "" 1; do
  echo "NOVALIDATE='$v': $(VLNCE_EXP_NOVALIDATE=$v timeout 200 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done | tee $out/novalidate.txt
