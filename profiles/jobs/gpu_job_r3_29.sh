#!/bin/bash
# round 3, job 29: conv_u3 epilogue through a per-wave LDS square (16-byte row stores): tests, per-layer times, bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ac
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "u3 or conv_p3 or conv2d_fwd or block or bottleneck" 2>&1 | tail -3
ONLY=l1_1x1_64_256,l2_1x1_128_512,l2_1x1s2_256_512,l3_1x1_256_1024,l3_1x1_512_256,l3_1x1s2_512_1024,l4_1x1_1024_512,l4_1x1s2_1024_2048,l4_1x1_512_2048
timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 10 --rounds 3 --only $ONLY > $O/convbench_u3.txt 2>&1
grep -v amdgpu $O/convbench_u3.txt
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03ac/bench.json").read().strip().split("\n")[-1])
c=d["config"]; r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"ahead",c.get("encode_ahead_ms_per_step"))
print("conv ms",r["kernel_ms_per_step"],"frac",r["frac"],"bf16",r["bf16_pipe"]["frac"],r["bf16_pipe"]["by_kernel"])
print(json.dumps(r.get("per_launch_floor")))
PY
