#!/bin/bash
# round 4, job 20: same-box A/B of the A-fragment double buffering in conv_p3 (3x3 layers), then the bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04t; mkdir -p $out
for lib in build/variants/libvlnce_noadb.so "" build/variants/libvlnce_noadb.so ""; do
  echo "lib='$lib': $(VLNCE_HIP_LIB=$lib timeout 200 python scripts/convbench.py --mode train --pro --iters 10 --rounds 3 --only 3x3 2>&1 | grep '^l[1-4]_' | awk '{printf "%s %s  ", $1, $5}')"
done | tee $out/p3_adb_ab.txt
for t in 3 4; do
  echo "p3_tile=$t: $(timeout 200 python scripts/convbench.py --mode train --pro --iters 10 --rounds 3 --only l2_3x3_128,l1_3x3 --opt p3_tile=$t 2>&1 | grep '^l[1-4]_' | awk '{printf "%s %s  ", $1, $5}')"
done | tee -a $out/p3_adb_ab.txt
timeout 300 python bench.py --no-cpu-baseline --no-f32-compare 2>/dev/null | tail -1 > $out/bench.json
python -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']
print('ms/step', d['ms_per_step'], 'ahead', d['config']['encode_ahead_ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'bf16 frac', r['bf16_pipe']['frac'], 'floor frac', r['per_launch_floor']['frac'])"
