#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/trunkbench.py --n 64 --iters 5 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/kt -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/stats_trunks_n64.md 900 > /dev/null
head -30 $O/stats_trunks_n64.md | cut -c1-150
python scripts/rocpd_timeline.py $db 2>/dev/null | tail -5
rm -rf $O/kt
