#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03j
mkdir -p $O
cd $GRAFT_REPO_ROOT
VLNCE_U3_WAVES=4 timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "(conv_p3 or conv2d_fwd) and not every_tile" -p no:cacheprovider > $O/u3w4_tests.log 2>&1
echo "u3 WAVES=4 tests rc=$?"; tail -3 $O/u3w4_tests.log
VLNCE_U3_WAVES=4 timeout 300 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only 1x1 > $O/cb_u3w4.txt 2>&1
grep "^l[1-4]" $O/cb_u3w4.txt
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "full gpu suite rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
cat $O/bench.json | cut -c1-3500
grep "plain\|conv attribution\|timed region\|encode_ahead" $O/bench.err
