#!/bin/bash
# round 6, job 18: conv_p3 128x64 tiles as two workgroups per CU (tile 7) against the one-workgroup form
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_18
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in main before; do
  L=$GRAFT_REPO_ROOT/vln-ce_amd/libvlnce_hip.so; [ $v = before ] && L=$GRAFT_REPO_ROOT/build/libvlnce_4b234cf.so
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode train --pro --backlog --only 3x3 > $O/cb_3x3_$v.txt 2>&1
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode eval --backlog --set r18 --n 416 > $O/cb_r18_$v.txt 2>&1
  VLNCE_HIP_LIB=$L timeout 300 python scripts/convbench.py --mode train --pro --backlog --set depth --n 416 --only 3x3 > $O/cb_d416_$v.txt 2>&1
done
for f in cb_3x3 cb_r18 cb_d416; do paste <(awk '{print $1, $5}' $O/${f}_main.txt) <(awk '{print $5}' $O/${f}_before.txt) | grep -v amdgpu; done
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/vln-ce_amd/libvlnce_hip.so timeout 300 python scripts/convbench.py --mode train --pro --backlog --only l1_3x3 --opt p3_tile=6 | grep l1_
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or bn" 2>&1 | tail -3
