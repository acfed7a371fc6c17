#!/bin/bash
# launch list of one cached-feature update on the current build
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_46
mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/ktu -- python $GRAFT_REPO_ROOT/scripts/bench_data_path.py --update-only --iters 6 > $O/ktu.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_one_step.py "$(find $O/ktu -name '*.db' | head -1)" > $O/cached_update_one_step.txt
rm -rf $O/ktu
head -50 $O/cached_update_one_step.txt | cut -c1-150
