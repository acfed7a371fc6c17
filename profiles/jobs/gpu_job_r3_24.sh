#!/bin/bash
# round 3, job 24: conv_u3 with every second workgroup of an XCD starting late (short-K, store-bound layers)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03x
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k 'rnn or lstm or gru or instruction or wgrad' 2>&1 | tail -2
timeout 300 python scripts/seqbench.py > $O/seqbench.txt 2>&1; grep 'rnn_seq' $O/seqbench.txt
ONLY=l1_1x1_64_256,l2_1x1_128_512,l2_1x1s2_256_512,l3_1x1_256_1024,l3_1x1_512_256
for st in 0 1 2 3; do
  VLNCE_U3_STAGGER=$st timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only $ONLY > $O/convbench_stagger_$st.txt 2>&1
done
paste <(grep -v amdgpu $O/convbench_stagger_0.txt | awk '{print $1, $5}') <(grep -v amdgpu $O/convbench_stagger_1.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_stagger_2.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_stagger_3.txt | awk '{print $5}') | tee $O/stagger.txt
for st in 0 2; do
  VLNCE_U3_STAGGER=$st timeout 300 python bench.py --no-cpu-baseline --no-f32-compare --steps 20 > $O/bench_stagger_$st.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/bench_stagger_$st.json').read().strip().split('\n')[-1]); print('stagger $st:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
