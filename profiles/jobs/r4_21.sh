#!/bin/bash
# round 4, job 21: bisection builds of conv_p3's matrix loop on the 3x3 layers (times only; results are garbage)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04u; mkdir -p $out
for v in "" nowait nowait_nob nowait_noa nowait_noab; do
  lib=""; [ -n "$v" ] && lib=build/variants/libvlnce_$v.so
  echo "$v: $(VLNCE_HIP_LIB=$lib timeout 200 python scripts/convbench.py --mode train --pro --iters 10 --rounds 3 --only 3x3_ 2>&1 | grep '^l[1-4]_' | awk '{printf "%s %s  ", $1, $5}')"
done | tee $out/p3_bisection.txt
