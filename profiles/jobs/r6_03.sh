#!/bin/bash
# round 6, job 3: the whole GPU test tier on the fp16-plane default (+ the new DAgger hook test and
# the full-size Seq2Seq / Waypoint goldens)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_03
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
