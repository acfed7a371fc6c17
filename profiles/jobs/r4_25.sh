#!/bin/bash
# round 4, job 25: BatchNorm inside the convolution without the fence: tests + bench A/B
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04y; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "bn_sums" > $out/tests_bn.txt 2>&1
echo "bn tests rc=$?"; tail -2 $out/tests_bn.txt
for v in 0 1 0 1; do
  echo "BN_FUSED=$v: $(VLNCE_BN_FUSED=$v timeout 200 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ahead", d["config"]["encode_ahead_ms_per_step"], "conv", r["kernel_ms_per_step"], "eager trunks", r["eager_single_stream_trunks_ms"], "launches", r["launches_per_step"])')"
done | tee $out/bn_fused_ab.txt
