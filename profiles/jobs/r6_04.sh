#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_04
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dagger_hooks_gpu.py -q -s 2>&1 | grep -E "step|passed|failed|Error" > $O/hooks_f16.txt
VLNCE_CONV_MATH=bf16 timeout 600 python -m pytest tests/test_dagger_hooks_gpu.py -q -s 2>&1 | grep -E "step|passed|failed|Error" > $O/hooks_bf16.txt
cat $O/hooks_f16.txt $O/hooks_bf16.txt
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_dagger_hooks_gpu.py 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
