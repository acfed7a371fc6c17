#!/bin/bash
# round 4, job 19: conv_p3 with A fragments one k-slab ahead: parity, per-layer times
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04s; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "conv" > $out/tests.txt 2>&1
echo "conv tests rc=$?"; tail -3 $out/tests.txt
timeout 200 python scripts/convbench.py --mode train --pro --iters 10 --rounds 3 --only 3x3 2>&1 | grep "^l[1-4]_" | tee $out/convbench_3x3.txt
timeout 200 python scripts/convbench.py --mode train --pro --iters 10 --rounds 3 --only 3x3 --set depth,r18 2>&1 | grep "^[dr][0-9c]" | tee -a $out/convbench_3x3.txt
