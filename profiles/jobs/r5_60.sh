#!/bin/bash
# the driver's command with the rank on one L3 domain (default), the whole socket, unbound
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_60
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-compare --no-pipeline 2>>$O/err.txt | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2; do
  echo "l3       $(run)"
  echo "socket   $(VLNCE_BIND_SOCKET=socket run)"
  echo "unbound  $(VLNCE_BIND_SOCKET=0 run)"
done | tee $O/l3_socket_unbound.txt
grep "bound to" $O/err.txt | head -3
