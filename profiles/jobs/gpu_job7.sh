#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $O
cd $GRAFT_REPO_ROOT
echo skip tests


for cfg in "VLNCE_RNN_STEP_FUSED=0 VLNCE_INSTR_DEDUP=0" "VLNCE_RNN_STEP_FUSED=0 VLNCE_INSTR_DEDUP=1" "VLNCE_RNN_STEP_FUSED=1 VLNCE_INSTR_DEDUP=1"; do
  env $cfg timeout 300 python scripts/bench_data_path.py --update-only 2>/dev/null | tail -1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/bench_data_path.py --update-only --iters 5 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/kt -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/kernel_stats_cached_update.md 2000 > /dev/null
head -40 $O/kernel_stats_cached_update.md
rm -rf $O/kt
