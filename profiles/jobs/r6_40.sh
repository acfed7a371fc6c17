#!/bin/bash
# conv_u3: B fragments requested a chunk ahead (BA = 2 on 64-row forms, 1 on 128-row forms; default) vs one slab ahead (variant ba0 = the kernel before)
O=gpurun_out/r6_40; mkdir -p $O
V=$PWD/build/variants/libvlnce_ba0.so
for lib in default ba0; do
  [ $lib = ba0 ] && export VLNCE_HIP_LIB=$V || unset VLNCE_HIP_LIB
  timeout 600 python scripts/convbench.py --mode train --pro --rotate 8 --only 1x1 > $O/conv_1x1_$lib.txt 2>&1
  timeout 600 python scripts/convbench.py --mode train --pro --rotate 8 --only 1x1 --dual bn > $O/conv_dual_$lib.txt 2>&1
  tail -1 $O/conv_1x1_$lib.txt; tail -1 $O/conv_dual_$lib.txt
done
for lib in default ba0 default ba0; do
  [ $lib = ba0 ] && export VLNCE_HIP_LIB=$V || unset VLNCE_HIP_LIB
  timeout 600 python bench.py --steps 30 --warmup 4 2>/dev/null | tee $O/bench_$lib.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
unset VLNCE_HIP_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -3
