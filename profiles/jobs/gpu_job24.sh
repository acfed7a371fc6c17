#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_a -- python $GRAFT_REPO_ROOT/scripts/convbench.py --mode train --iters 3 --only l1_,l2_,l3_,l4_ > $O/pmc_a.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/pmc_a -name "*.db" | head -1)
python scripts/rocpd_pmc_layers.py $db 6 conv_x3 > $O/pmc_a_layers.txt 2>&1
rm -rf $O/pmc_a
grep -v amdgpu $O/pmc_a.log | grep "^l[1-4]_" | awk '{print $1}' > $O/names.txt
cat $O/pmc_a_layers.txt | cut -c1-220 | head -40
