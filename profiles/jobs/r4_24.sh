#!/bin/bash
# round 4, job 24: BatchNorm finished inside the convolution: bench A/B, then the full GPU suite
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04x; mkdir -p $out
for v in 0 1 0 1; do
  echo "BN_FUSED=$v: $(VLNCE_BN_FUSED=$v timeout 200 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ahead", d["config"]["encode_ahead_ms_per_step"], "conv", r["kernel_ms_per_step"], "eager trunks", r["eager_single_stream_trunks_ms"])')"
done | tee $out/bn_fused_ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/gpu_suite.txt 2>&1
echo "gpu suite rc=$?"; tail -5 $out/gpu_suite.txt
