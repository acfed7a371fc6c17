#!/bin/bash
# round 4, job 1 (prepared at the end of round 3, when the GPU budget was spent): the things
# written blind.  1) conv_s3p_kernel parity (forced over the conv / block cases), 2) its per-layer
# time against conv_s3_kernel on the two expansion shapes, 3) the depth trunk on its own stream in
# a training step, 4) a fresh per-stream timeline of the plain loop.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04a; mkdir -p $out
VLNCE_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider \
  -k "pipelined_epilogue" > $out/s3p_parity.txt 2>&1
echo "s3p parity rc=$?"; tail -3 $out/s3p_parity.txt
for pipe in 0 1; do
  echo "S3_PIPE=$pipe: $(VLNCE_S3_PIPE=$pipe timeout 120 python scripts/convbench.py --mode train --pro --set r50 \
     --iters 10 --rounds 3 --only l1_1x1_64_256,l2_1x1_128_512 2>&1 | grep '^l[12]_' | awk '{printf "%s %s us  ", $1, $5}')"
done | tee $out/s3p_convbench.txt
# conv_u3 with the chunk's raw-row / slab-1 loads pinned in front of the first MFMA: build the
# variant in the CPU container first:  scripts/build_variants.sh u3lf "-DU3_LOADS_FIRST"
if [ -f build/variants/libvlnce_u3lf.so ]; then
  for lib in "" build/variants/libvlnce_u3lf.so; do
    echo "lib='$lib': $(VLNCE_HIP_LIB=$lib timeout 200 python scripts/convbench.py --mode train --pro --set r50 \
       --iters 10 --rounds 3 --only 1x1 2>&1 | grep '^l[1-4]_' | awk '{printf "%s %s  ", $1, $5}')"
  done | tee $out/u3_loads_first_convbench.txt
fi
for br in "" split; do
  echo "TRAIN_BRANCHES='$br': $(VLNCE_TRAIN_BRANCHES=$br timeout 200 python bench.py --no-cpu-baseline --no-f32-compare \
     2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["config"]["encode_ahead_ms_per_step"])')"
done | tee $out/train_branches.txt
O=$GRAFT_REPO_ROOT/$out
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/step_profile.py --steps 12 --warmup 6 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_stats.py $db $O/step_kernel_stats.md > /dev/null 2>&1
python scripts/rocpd_timeline.py $db 0 > $O/step_timeline.txt 2>&1
rm -rf $O/trace
cut -c1-300 $O/step_timeline.txt | head -60
