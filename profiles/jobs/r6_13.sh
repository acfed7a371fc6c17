#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_13
mkdir -p $O
cd $GRAFT_REPO_ROOT
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_dbg.so timeout 300 python scripts/convbench.py --mode train --pro --iters 1 --rounds 1 --only l1_3x3_64_64,l2_3x3_128_128,l3_3x3_256_256 > $O/dbg_3x3.txt 2>&1
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_dbg.so timeout 300 python scripts/convbench.py --mode eval --iters 1 --rounds 1 --set r18 --n 416 --only r18_3x3_64_64,r18_3x3_128_128 > $O/dbg_r18.txt 2>&1
grep -v amdgpu $O/dbg_3x3.txt | tail -40; grep -v amdgpu $O/dbg_r18.txt | tail -24
