#!/bin/bash
# round 3, job 31: conv_u3 short-K layers as 64-row tiles with two workgroups per CU
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ae
mkdir -p $O
cd $GRAFT_REPO_ROOT
ONLY=l1_1x1_64_256,l2_1x1_128_512,l2_1x1s2_256_512,l3_1x1_256_1024
for pr in 0 64 128 256; do
  VLNCE_U3_PAIR=$pr timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only $ONLY > $O/convbench_pair_$pr.txt 2>&1
done
paste <(grep -v amdgpu $O/convbench_pair_0.txt | awk '{print $1, $5}') <(grep -v amdgpu $O/convbench_pair_64.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_pair_128.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_pair_256.txt | awk '{print $5}') | tee $O/pair.txt
VLNCE_U3_PAIR=128 timeout 600 python -m pytest tests/test_policy_sizes_gpu.py -m gpu -x -q -p no:cacheprovider -k "bench_geometry" 2>&1 | tail -2
