#!/bin/bash
# round 4, job 31: stem7 kernel: parity, then the bench A/B and the Waypoint step
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04ze; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "stem7" > $out/tests_stem7.txt 2>&1
echo "stem7 tests rc=$?"; tail -12 $out/tests_stem7.txt
for v in 0 1 0 1; do
  echo "STEM7=$v: $(VLNCE_STEM7=$v timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-compare 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ahead", d["config"]["encode_ahead_ms_per_step"], "conv", r["kernel_ms_per_step"], "eager trunks", r["eager_single_stream_trunks_ms"], "launches", r["launches_per_step"])')"
done | tee $out/stem7_ab.txt
