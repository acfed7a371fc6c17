#!/bin/bash
# the driver's bench command with the rank bound to its GPU's socket (default) and unbound, alternated;
# then the complete default line (cpu_baseline must still see the whole host)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_56
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-compare --no-pipeline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2 3 4 5; do
  echo "bound    $(run)"
  echo "unbound  $(VLNCE_BIND_SOCKET=0 run)"
done | tee $O/bound_vs_unbound.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; grep -E "bound|cpu_baseline" $O/bench.err | head -12
