#!/bin/bash
# round 6, job 7: p3 producers' row offsets + prologue vectors in LDS, u3 prologue vectors in LDS --
# per-layer A/B against the build before (build/libvlnce_base_r6_06.so = 16 row groups in registers),
# conv tests, bench line, conv accuracy of both plane formats
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_07
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or bn or stem7 or fp16 or planes" 2>&1 | tail -8 > $O/pytest_conv.txt
tail -3 $O/pytest_conv.txt
timeout 300 python scripts/convbench.py --mode train --pro --backlog > $O/convbench_train_pro.txt 2>&1
timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual identity > $O/convbench_dual_identity.txt 2>&1
timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual bn > $O/convbench_dual_bn.txt 2>&1
timeout 300 python scripts/convbench.py --mode eval --backlog --set r18 --n 416 > $O/convbench_r18.txt 2>&1
cat $O/convbench_train_pro.txt $O/convbench_dual_identity.txt $O/convbench_r18.txt
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv', r['kernel_ms_per_step'], 'frac', r['frac'], r['bf16_pipe']['frac'])
P
timeout 300 python scripts/conv_accuracy.py > $O/conv_accuracy_f16x3.txt 2>&1
VLNCE_CONV_MATH=bf16 timeout 300 python scripts/conv_accuracy.py > $O/conv_accuracy_bf16x6.txt 2>&1
tail -30 $O/conv_accuracy_f16x3.txt; tail -30 $O/conv_accuracy_bf16x6.txt
