#!/bin/bash
# round 3, job 2: conv_p3 with tall per-wave sub-tiles: parity incl. forced tiles, A/B, in-kernel timing, PMC
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pack_weights or conv_p3 or conv2d_fwd" -p no:cacheprovider > $O/p3_tests.log 2>&1
echo "p3 tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/p3_tests.log
VLNCE_P3=1 timeout 200 python scripts/convbench.py --mode train --pro --set r50,r18 --iters 10 > $O/convbench_train_p3_1.txt 2>&1
echo "convbench train p3=1 rc=$?" | tee -a $O/summary.txt
grep -v amdgpu $O/convbench_train_p3_1.txt
ONLY=l1_3x3,l2_3x3_,l3_3x3_,l4_3x3_,l2_1x1_128_512,l3_1x1_256_1024,l3_1x1_1024_256,l2_1x1_512_128,l4_1x1_512_2048
for t in 1 2 3 4 5 6; do
  VLNCE_P3_TILE=$t timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 5 --only $ONLY > $O/convbench_tile_$t.txt 2>&1
done
paste <(grep -v amdgpu $O/convbench_tile_1.txt | awk '{print $1, $5}') <(grep -v amdgpu $O/convbench_tile_2.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_tile_3.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_tile_4.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_tile_5.txt | awk '{print $5}') <(grep -v amdgpu $O/convbench_tile_6.txt | awk '{print $5}') | tee $O/tiles.txt
# in-kernel timing (debug build)
SEL=l1_3x3,l2_3x3_,l3_3x3_,l3_1x1_256_1024,l3_1x1_1024_256,l2_1x1_128_512
VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_p3time.so timeout 200 python scripts/convbench.py --mode train --pro --set r50 --iters 1 --only $SEL > $O/p3time.txt 2>&1
grep -v amdgpu $O/p3time.txt | cut -c1-260
# PMC
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $O/pmc_a -- python $GRAFT_REPO_ROOT/scripts/convbench.py --mode train --pro --iters 3 --only $SEL > $O/pmc_a.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/pmc_a -name "*.db" | head -1)
python scripts/rocpd_pmc_layers.py $db 6 conv_p3 > $O/pmc_a_layers.txt 2>&1
rm -rf $O/pmc_a
cut -c1-300 $O/pmc_a_layers.txt | head -20
grep -i "TA_\|TCP_\|SQ_INSTS\|SQ_WAIT\|LDS" $O/counters.txt | cut -c1-150 | head -60
