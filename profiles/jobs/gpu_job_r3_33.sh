#!/bin/bash
# round 3, job 33: k-major LDS images for the transposed GEMM operands (weight gradients)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ag
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_trainable_encoders.py -m gpu -x -q -p no:cacheprovider -k "gemm or wgrad or linear or trainable or rollout or attention" 2>&1 | tail -2
timeout 300 python bench.py --trainable-encoders --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare > $O/bench_trainable.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_trainable.json').read().strip().split('\n')[-1]); print('trainable', d['value'], d['ms_per_step'])"
timeout 300 python scripts/bench_data_path.py --update-only --iters 30 2>/dev/null | tail -1
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('value', d['value'], d['ms_per_step'], d['config'].get('encode_ahead_ms_per_step'))"
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/step_profile.py --trainable-encoders --steps 4 --warmup 3 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_one_step.py "$(find $O/kt -name '*.db' | head -1)" > $O/trainable_one_step.txt
rm -rf $O/kt
head -8 $O/trainable_one_step.txt | cut -c1-150
