#!/bin/bash
# round 6, job 2: fp16-plane arithmetic (format 2, three plane products) -- kernel tests, per-layer A/B
# against format 1 on the same box, bench line in both formats
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_02
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -15 > $O/pytest_kernels.txt
tail -3 $O/pytest_kernels.txt
for m in 2 1; do
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --opt conv_math=$m > $O/convbench_train_pro_math$m.txt 2>&1
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual identity --opt conv_math=$m > $O/convbench_dual_identity_math$m.txt 2>&1
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual bn --opt conv_math=$m > $O/convbench_dual_bn_math$m.txt 2>&1
  timeout 300 python scripts/convbench.py --mode train --pro --backlog --set depth --opt conv_math=$m > $O/convbench_depth_math$m.txt 2>&1
  timeout 300 python scripts/convbench.py --mode eval --backlog --set r18 --n 416 --opt conv_math=$m > $O/convbench_r18_math$m.txt 2>&1
done
paste <(awk '{print $1, $5, $6}' $O/convbench_train_pro_math1.txt) <(awk '{print $5, $6}' $O/convbench_train_pro_math2.txt) | column -t
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench_f16x3.json 2> $O/bench_f16x3.err
VLNCE_CONV_MATH=bf16 timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench_bf16x6.json 2> $O/bench_bf16x6.err
for f in f16x3 bf16x6; do python - <<P
import json
d=json.loads(open('$O/bench_$f.json').read().strip().split('\n')[-1]); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'])
P
done
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -30 $O/conv_launch_times.txt
