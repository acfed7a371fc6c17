#!/bin/bash
# round 4, job 23: BatchNorm finished inside the convolution: kernel tests, policy tests, bench A/B
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04w; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "bn_fused" > $out/tests_bn.txt 2>&1
echo "bn tests rc=$?"; tail -15 $out/tests_bn.txt
