#!/bin/bash
# round 3, job 1: first contact of conv_p3_kernel with the GPU: parity, per-layer A/B, full suite, bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pack_weights or conv_p3 or conv2d_fwd" -p no:cacheprovider > $O/p3_tests.log 2>&1
echo "p3 tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/p3_tests.log
for mode in train eval; do
  for p3 in 0 1; do
    extra=""; [ $mode = train ] && extra="--pro"
    VLNCE_P3=$p3 timeout 150 python scripts/convbench.py --mode $mode $extra --set r50,depth,r18 --iters 10 > $O/convbench_${mode}_p3_$p3.txt 2>&1
    echo "convbench $mode p3=$p3 rc=$?" | tee -a $O/summary.txt
  done
done
paste <(grep -v amdgpu $O/convbench_train_p3_0.txt | awk '{print $1, $2, $3, $4, $5, $6}') <(grep -v amdgpu $O/convbench_train_p3_1.txt | awk '{print $5, $6}') | column -t > $O/ab_train.txt
paste <(grep -v amdgpu $O/convbench_eval_p3_0.txt | awk '{print $1, $2, $3, $4, $5, $6}') <(grep -v amdgpu $O/convbench_eval_p3_1.txt | awk '{print $5, $6}') | column -t > $O/ab_eval.txt
cat $O/ab_train.txt
timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "full gpu suite rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
cat $O/bench.json | cut -c1-1500
