#!/bin/bash
# kernel statistics of the two trunks alone at num_envs 64 (graph replay), and their stand-alone times
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_43
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/trunkbench.py --n 64 --iters 20 2>/dev/null | tee $O/trunkbench.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/trunkbench.py --n 64 --iters 17 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/trunks_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -45 $O/trunks_kernel_stats.md | cut -c1-150
