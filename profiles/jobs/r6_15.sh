#!/bin/bash
# round 6, job 15: conv_u3 raw-row ring of 3-4 chunks (main build) against the two-set ring (variant ring2),
# operands from HBM (--rotate) and cache-hot; tests; bench A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_15
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in main ring2; do
  L=$GRAFT_REPO_ROOT/vln-ce_amd/libvlnce_hip.so; [ $v = ring2 ] && L=$GRAFT_REPO_ROOT/build/variants/libvlnce_ring2.so
  for r in 1 20; do
    VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode train --pro --backlog --rotate $r --iters 40 --only 1x1 > $O/cb_${v}_r$r.txt 2>&1
    VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual identity --rotate $r --iters 40 --only l3_,l4_ > $O/cbd_${v}_r$r.txt 2>&1
  done
done
echo "layer / main hot / ring2 hot / main HBM / ring2 HBM"
paste <(awk '{print $1, $5}' $O/cb_main_r1.txt) <(awk '{print $5}' $O/cb_ring2_r1.txt) <(awk '{print $5}' $O/cb_main_r20.txt) <(awk '{print $5}' $O/cb_ring2_r20.txt) | grep -v amdgpu
paste <(awk '{print $1, $5}' $O/cbd_main_r1.txt) <(awk '{print $5}' $O/cbd_ring2_r1.txt) <(awk '{print $5}' $O/cbd_main_r20.txt) <(awk '{print $5}' $O/cbd_ring2_r20.txt) | grep -v amdgpu
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or bn" 2>&1 | tail -3
for rep in 1 2; do for v in main ring2; do
  L=$GRAFT_REPO_ROOT/vln-ce_amd/libvlnce_hip.so; [ $v = ring2 ] && L=$GRAFT_REPO_ROOT/build/variants/libvlnce_ring2.so
  VLNCE_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<P
import json
d=json.loads(open('$O/bench_$v.json').read().strip().split('\n')[-1]); r=d['roofline']
print('$v', d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'])
P
done; done | tee $O/bench_ab.txt
