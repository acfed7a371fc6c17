#!/bin/bash
# same box: the final commit of round 4 (a worktree built in-tree) against this tree --
# the headline bench and the cached-feature update, alternating
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_41
mkdir -p $O
cd $GRAFT_REPO_ROOT
b() { (cd $1 && timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --no-pipeline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); }
d() { (cd $1 && timeout 300 python scripts/bench_data_path.py 2>/dev/null | grep -o '"cma_update_ms": [0-9.]*'); }
for rep in 1 2 3; do
  echo "round-4 final (0573ed6)  bench $(b gpurun_tmp/r4)   cached-feature $(d gpurun_tmp/r4)"
  echo "this tree                bench $(b .)   cached-feature $(d .)"
done | tee $O/r4_vs_r5_same_box.txt
echo "this tree, VLNCE_RNN_WGRAD_STREAMS=0: $(VLNCE_RNN_WGRAD_STREAMS=0 d .)" | tee -a $O/r4_vs_r5_same_box.txt
echo "this tree, VLNCE_INSTR_DEDUP=0: $(VLNCE_INSTR_DEDUP=0 d .)" | tee -a $O/r4_vs_r5_same_box.txt
