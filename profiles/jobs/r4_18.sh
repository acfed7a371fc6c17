#!/bin/bash
# round 4, job 18: in-kernel phase timers of conv_p3 on the four big 3x3 layers; check of the u3 64-row rule
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04r; mkdir -p $out
for l in l1_3x3_64_64 l2_3x3_128_128 l3_3x3_256_256 l4_3x3_512_512; do
  echo "== $l"
  VLNCE_HIP_LIB=build/variants/libvlnce_p3time.so timeout 100 python scripts/convbench.py --mode train --pro --iters 1 --rounds 1 --only $l 2>&1 | grep -E "^p3 |^l[1-4]_" | sort | uniq -c | sort -rn | head -8
done | cut -c1-330 | tee $out/p3_phase_timers.txt
timeout 200 python scripts/convbench.py --mode train --pro --dual identity --iters 10 --rounds 3 --only l3_1x1_1024_256,l4_1x1_2048_512 2>&1 | grep "^l[1-4]_" | tee $out/u3_rule_check.txt
