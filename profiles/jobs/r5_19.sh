#!/bin/bash
# secondary workloads through bench.py, 1-rank RCCL exchange fields, new goldens on the GPU
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_19
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -k "golden" 2>&1 | tail -3
timeout 600 python bench.py --policy waypoint --force-dist --steps 10 --warmup 3 > $O/bench_waypoint.json 2> $O/bench_waypoint.err; tail -c 900 $O/bench_waypoint.json; echo
timeout 600 python bench.py --policy seq2seq --force-dist --steps 20 > $O/bench_seq2seq.json 2> $O/bench_seq2seq.err; tail -c 1200 $O/bench_seq2seq.json | cut -c1-900; echo
timeout 600 python bench.py --force-dist --no-cpu-baseline --no-f32-compare --no-pipeline --steps 30 2> $O/bench_cma_dist.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cma --force-dist', d['ms_per_step'], d.get('allreduce_ms'), d.get('allreduce_hidden_frac'), d.get('allreduce'))"
