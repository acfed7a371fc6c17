#!/bin/bash
# bisection / tuning builds of libvlnce_hip.so: one library per -D flag set, under build/variants/
# usage: scripts/build_variants.sh NAME "-DFLAG ..." [NAME "-DFLAG ..."]...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d=build/variants/obj_$name; mkdir -p $d
  for f in vln-ce_amd/csrc/*.hip vln-ce_amd/csrc/*.cpp; do
    b=$(basename ${f%.*})
    if [ "$b" = "igemm" ] || [ "$b" = "conv_p3" ] || [ ! -f build/$b.o ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $f -o $d/$b.o &
    else
      cp build/$b.o $d/$b.o
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o build/variants/libvlnce_$name.so
  echo built build/variants/libvlnce_$name.so
done
