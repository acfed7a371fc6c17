#!/bin/bash
# round 3, job 47: bench.py through a 1-rank RCCL group (the data-parallel code path) and as a 1-process torchrun
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03dist
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --force-dist --steps 15 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_force_dist.json 2> $O/bench_force_dist.err
python -c "
import json
d=json.loads(open('$O/bench_force_dist.json').read().strip().split('\n')[-1]); print('force-dist', d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('parallelism'))"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 15 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
python -c "
import json
d=json.loads(open('$O/bench_torchrun1.json').read().strip().split('\n')[-1]); print('torchrun 1 proc', d['value'], d['ms_per_step'], d['n_gpus'])"
tail -3 $O/bench_torchrun1.err
