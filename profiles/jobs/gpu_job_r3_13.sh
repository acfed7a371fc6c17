#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p $O
cd $GRAFT_REPO_ROOT
for ord in rgb_first side_first; do
  VLNCE_BRANCH_ORDER=$ord timeout 300 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench_$ord.json 2> $O/bench_$ord.err
  echo "$ord: $(python -c "import json;d=json.load(open('$O/bench_$ord.json'));print(d['ms_per_step'], d['config']['encode_ahead_ms_per_step'])")"
done
VLNCE_SIDE_STREAMS=0 timeout 300 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench_nostreams.json 2> $O/bench_nostreams.err
echo "no side streams: $(python -c "import json;d=json.load(open('$O/bench_nostreams.json'));print(d['ms_per_step'], d['config']['encode_ahead_ms_per_step'])")"
