#!/bin/bash
# PMC pass on the final kernels: effective clock and bf16-pipe busy fraction per layer (RGB ResNet-50 and depth trunk shapes)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_67
mkdir -p $O
for set in r50 depth; do
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_$set -- python $GRAFT_REPO_ROOT/scripts/convbench.py --set $set --mode train --pro --iters 3 > $O/convbench_$set.txt 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/pmc_$set -name "*.db" | head -1)
python scripts/rocpd_pmc_layers.py $db 6 conv_ > $O/pmc_mfma_busy_$set.txt 2>&1
rm -rf $O/pmc_$set
grep -v amdgpu $O/convbench_$set.txt | awk '{print $1, $2, $3, $4}' | head -40 > $O/layers_$set.txt
done
head -30 $O/pmc_mfma_busy_r50.txt | cut -c1-140
