#!/bin/bash
# round 6, job 1: per-layer baseline of the round-5 kernels on this round's box + in-kernel phase
# timers of conv_u3 / conv_x3 on the layers the review names
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_01
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/convbench.py --mode train --pro --backlog > $O/convbench_train_pro.txt 2>&1
timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual identity > $O/convbench_dual_identity.txt 2>&1
timeout 300 python scripts/convbench.py --mode train --pro --backlog --dual bn > $O/convbench_dual_bn.txt 2>&1
timeout 300 python scripts/convbench.py --mode eval --backlog > $O/convbench_eval.txt 2>&1
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_dbg.so timeout 300 python scripts/convbench.py --mode train --pro --iters 1 --rounds 1 \
  --only l3_1x1_256_1024,l3_1x1_1024_256,l4_1x1_512_2048,l3_3x3_256_256,l2_1x1_512_128,l4_1x1_2048_512 > $O/dbg_timers.txt 2>&1
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_dbg.so timeout 300 python scripts/convbench.py --mode train --pro --iters 1 --rounds 1 --dual identity \
  --only l3_1x1_1024_256,l2_1x1_512_128,l4_1x1_2048_512,l1_1x1_256_64 > $O/dbg_timers_dual.txt 2>&1
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-f32-compare > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
