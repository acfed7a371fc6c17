#!/bin/bash
# round-3 evidence run: full GPU tests, bench line, kernel-trace stats, PMC traffic + MFMA-busy passes, secondary configs
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03q
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
grep -n "passed\|failed" $O/gpu_tests.log | tail -2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
timeout 300 python scripts/conv_accuracy.py --n 16 > $O/conv_accuracy.txt 2>&1
VLNCE_P3=0 VLNCE_U3=0 timeout 300 python scripts/conv_accuracy.py --n 16 > $O/conv_accuracy_x3.txt 2>&1
grep -v amdgpu $O/conv_accuracy.txt
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -3 $O/conv_launch_times.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/kt -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/bench_kernel_stats.md 900 > /dev/null
head -16 $O/bench_kernel_stats.md | cut -c1-150
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --pmc-step > $O/pmc_$c.log 2>&1
  cd $GRAFT_REPO_ROOT
  db=$(find $O/pmc_$c -name "*.db" | head -1)
  python scripts/rocpd_pmc.py $db > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
  grep -n "segment\|conv_\|igemm\|copyBuffer\|elementwise" $O/pmc_$c.txt | head -24 | cut -c1-170
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_busy -- python $GRAFT_REPO_ROOT/scripts/convbench.py --mode train --pro --iters 3 --rounds 1 --only l1_,l2_,l3_,l4_ > $O/pmc_busy.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/pmc_busy -name "*.db" | head -1)
python scripts/rocpd_pmc_layers.py $db 6 conv_ > $O/pmc_busy_layers.txt 2>&1
rm -rf $O/pmc_busy
grep -v amdgpu $O/pmc_busy.log | grep "^l[1-4]_" | awk '{print $1}' > $O/pmc_busy_names.txt
cut -c1-250 $O/pmc_busy_layers.txt | head -30
timeout 400 python scripts/bench_policies.py > $O/bench_other_policies.jsonl 2> $O/bench_other_policies.err
tail -3 $O/bench_other_policies.jsonl | cut -c1-250
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> $O/bench_data_path.err
tail -1 $O/bench_data_path.json | cut -c1-400
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
tail -c 2500 $O/bench_trainable.json | head -c 400
