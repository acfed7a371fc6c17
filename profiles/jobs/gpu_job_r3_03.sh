#!/bin/bash
# round 3, job 3: conv_p3 with 128-row DENSE producer items and LDS-DMA B for the 1x1 form
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_obs_transforms.py -x -q -m gpu -k "pack_weights or conv_p3 or conv2d_fwd or resize or obs_stack" -p no:cacheprovider > $O/p3_tests.log 2>&1
echo "p3 + resize tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/p3_tests.log
for rep in 1 2; do
  VLNCE_P3=1 timeout 200 python scripts/convbench.py --mode train --pro --set r50,r18 --iters 10 > $O/convbench_train_p3_1_rep$rep.txt 2>&1
done
VLNCE_P3=0 timeout 200 python scripts/convbench.py --mode train --pro --set r50,r18 --iters 10 > $O/convbench_train_p3_0.txt 2>&1
paste <(grep -v amdgpu $O/convbench_train_p3_0.txt | awk '{printf "%-22s %8s %6s %5s %9s\n", $1,$2,$3,$4,$5}') <(grep -v amdgpu $O/convbench_train_p3_1_rep1.txt | awk '{printf "%9s %7s\n", $5,$6}') <(grep -v amdgpu $O/convbench_train_p3_1_rep2.txt | awk '{printf "%9s\n", $5}') | tee $O/ab_train.txt
SEL=l1_3x3,l2_3x3_,l3_3x3_,l4_3x3_,l1_1x1_64_256,l3_1x1_256_1024,l3_1x1_1024_256,l2_1x1_128_512
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_p3time.so timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 1 --only $SEL > $O/p3time.txt 2>&1
grep "^p3\|^l[1-4]" $O/p3time.txt | awk '!seen[$0]++' | cut -c1-250 | head -60
