#!/bin/bash
# dx_from (frozen feature columns) + long-reduction split-K: policy tests, cached-feature update, gemm log, bench
O=$GRAFT_REPO_ROOT/gpurun_out/r04_45
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or linear" -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do timeout 300 python scripts/bench_data_path.py --update-only 2>/dev/null | tail -1; done
timeout 300 python scripts/gemm_log.py > $O/gemm_log_cached.txt 2>$O/err1.txt || tail -5 $O/err1.txt
head -8 $O/gemm_log_cached.txt
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print('bench', d['value'], d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'])
P
