#!/bin/bash
# conv_p3: B fragments two steps ahead for the one-/two-block waves (default) vs one slab ahead (variant bd1 = before)
O=gpurun_out/r6_41; mkdir -p $O
V=$PWD/build/variants/libvlnce_bd1.so
for lib in default bd1; do
  [ $lib = bd1 ] && export VLNCE_HIP_LIB=$V || unset VLNCE_HIP_LIB
  timeout 600 python scripts/convbench.py --mode train --pro --rotate 4 --only 3x3 > $O/conv_3x3_$lib.txt 2>&1
  timeout 600 python scripts/convbench.py --mode eval --rotate 2 --set r18 --n 416 --only 3x3 > $O/conv_r18_$lib.txt 2>&1
  tail -n 1 $O/conv_3x3_$lib.txt; tail -n 1 $O/conv_r18_$lib.txt
done
unset VLNCE_HIP_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -n 3
for lib in default bd1 default bd1; do
  [ $lib = bd1 ] && export VLNCE_HIP_LIB=$V || unset VLNCE_HIP_LIB
  timeout 600 python bench.py --steps 30 --warmup 4 2>/dev/null | tee $O/bench_$lib.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
for lib in default bd1; do
  [ $lib = bd1 ] && export VLNCE_HIP_LIB=$V || unset VLNCE_HIP_LIB
  timeout 600 python bench.py --policy waypoint --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee $O/bench_waypoint_$lib.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waypoint $lib', d['ms_per_step'])"
done
