#!/bin/bash
# round 4, job 17: dual block-end 1x1 layers: default dispatch vs conv_u3 forced (64- / 128-row tiles)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04q; mkdir -p $out
for o in "" "u3=2" "u3=3"; do
  echo "== dual identity, options '$o'"
  timeout 200 python scripts/convbench.py --mode train --pro --dual identity --iters 10 --rounds 3 --opt "$o" 2>&1 | grep "^l[1-4]_"
done | tee $out/convbench_dual_u3.txt
for o in "" "u3=2"; do
  echo "== single input 1x1, options '$o'"
  timeout 200 python scripts/convbench.py --mode train --pro --iters 10 --rounds 3 --only 1x1 --opt "$o" 2>&1 | grep "^l[1-4]_"
done | tee -a $out/convbench_dual_u3.txt
