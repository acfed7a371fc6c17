#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $O
for f in 0 1; do
cd /tmp
VLNCE_GN_FUSED=$f timeout 300 rocprofv3 --kernel-trace -d $O/kt$f -- python $GRAFT_REPO_ROOT/scripts/trunkbench.py --n 1 --iters 5 > $O/kt$f.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/kt$f -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/stats_gn$f.md 1500 > /dev/null
head -16 $O/stats_gn$f.md | cut -c1-140
rm -rf $O/kt$f
done
