#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03n
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_policy_gpu.py -x -q -m gpu -k "embedding or instruction or golden or update or replay" -p no:cacheprovider > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench.json'));print('plain', d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv', d['roofline']['kernel_ms_per_step'])"
