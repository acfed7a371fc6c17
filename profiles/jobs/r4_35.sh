#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04zi; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "stem7" 2>&1 | tail -2
for v in "" s7w8; do
  lib=""; [ -n "$v" ] && lib=build/variants/libvlnce_$v.so
  echo "variant '$v': $(VLNCE_HIP_LIB=$lib timeout 200 python scripts/stem_time.py 2>&1 | grep '^stem7' | awk '{printf "%s/%s/%s %s us  ", $2, $4, $5, $6}')"
done | tee $out/stem7_waves.txt
