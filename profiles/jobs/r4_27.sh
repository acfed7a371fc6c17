#!/bin/bash
# round 4, job 27: plain-loop A/B of the sums path, 60 steps, alternating
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04za; mkdir -p $out
for v in 0 1 0 1 0 1; do
  echo "BN_FUSED=$v: $(VLNCE_BN_FUSED=$v timeout 200 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-f32-compare --no-pipeline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done | tee $out/bn_sums_plain_loop_ab.txt
