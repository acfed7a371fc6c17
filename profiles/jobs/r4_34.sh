#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04zh; mkdir -p $out
for v in "" NOA NOSTORE NOBUILD; do
  lib=""; [ -n "$v" ] && lib=build/variants/libvlnce_s7$v.so
  echo "variant '$v': $(VLNCE_HIP_LIB=$lib timeout 200 python scripts/stem_time.py 2>&1 | grep '^stem7 64 frames torch.float32' | awk '{printf "%s %s us  ", $5, $6}')"
done | tee $out/stem7_bisection.txt
