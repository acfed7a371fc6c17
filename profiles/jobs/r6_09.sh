#!/bin/bash
# round 6, job 9: sixteen row groups for the small-matrix-role conv_p3 tiles (256x64 tiles on 64x64 maps)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_09
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in main before; do
  L=$GRAFT_REPO_ROOT/vln-ce_amd/libvlnce_hip.so; [ $v = before ] && L=$GRAFT_REPO_ROOT/build/libvlnce_f6474c0.so
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode train --pro --backlog --only 3x3 > $O/convbench_3x3_$v.txt 2>&1
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode eval --backlog --set r18 --n 416 > $O/convbench_r18_$v.txt 2>&1
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode train --pro --backlog --set depth --n 416 --only 3x3 > $O/convbench_depth416_$v.txt 2>&1
  cat $O/convbench_3x3_$v.txt $O/convbench_r18_$v.txt $O/convbench_depth416_$v.txt | grep -v amdgpu
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or bn or planes" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], r['bf16_pipe']['frac'])
P
timeout 600 python bench.py --policy waypoint --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_waypoint.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_waypoint.json | sed 's/^/waypoint /'
