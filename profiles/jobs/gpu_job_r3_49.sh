#!/bin/bash
# round 3, job 49: instruction RNN kernels without scratch (weight fragments of the last gate in LDS, fetches moved)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ap
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k 'rnn or lstm or gru or instruction' 2>&1 | tail -2
timeout 300 python scripts/seqbench.py > $O/seqbench.txt 2>&1; grep 'rnn_seq' $O/seqbench.txt
timeout 300 python scripts/bench_data_path.py --update-only --iters 30 2>/dev/null | tail -1
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('value', d['value'], d['ms_per_step'], d['config'].get('encode_ahead_ms_per_step'))"
