#!/bin/bash
# round-2 job 1: baseline GPU tests, new bench line, convbench + PMC passes of the conv layers
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-1500
timeout 300 python scripts/convbench.py --mode train > $O/convbench_train.txt 2>&1
tail -30 $O/convbench_train.txt
rocprofv3 -L > $O/counters.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_a -- python $GRAFT_REPO_ROOT/scripts/convbench.py --mode train --iters 3 > $O/pmc_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_b -- python $GRAFT_REPO_ROOT/scripts/convbench.py --mode train --iters 3 > $O/pmc_b.log 2>&1
cd $GRAFT_REPO_ROOT
for p in a b; do
  db=$(find $O/pmc_$p -name "*.db" | head -1)
  python scripts/rocpd_pmc_layers.py $db 6 > $O/pmc_${p}_layers.txt 2>&1
  rm -rf $O/pmc_$p
done
cat $O/pmc_a_layers.txt | cut -c1-200
