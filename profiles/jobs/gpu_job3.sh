#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $O
cd $GRAFT_REPO_ROOT
python scripts/membench.py 2>&1 | tee $O/membench.txt
L="l1_1x1_64_256,l2_1x1_128_512,l1_3x3_64_64,l3_3x3_256_256,l3_1x1_256_1024"
for lib in "" nostore nomfma ntstore nostats; do
  for mode in "VLNCE_PK_TILES=8" "VLNCE_IGEMM_NO_PERSIST=1"; do
    libenv="VLNCE_X=1"; [ -n "$lib" ] && libenv="VLNCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libvlnce_$lib.so"
    echo "== lib=${lib:-base} $mode"
    env $libenv $mode timeout 120 python scripts/convbench.py --mode train --only $L 2>&1 | grep -v amdgpu.ids | awk '{printf "%s %s %s | ", $1, $5, $6} END {print ""}' | tee -a $O/variants.txt
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum -d $O/pmc_c -- python $GRAFT_REPO_ROOT/scripts/convbench.py --mode train --iters 3 --only $L > $O/pmc_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_d -- python $GRAFT_REPO_ROOT/scripts/convbench.py --mode train --iters 3 --only $L > $O/pmc_d.log 2>&1
cd $GRAFT_REPO_ROOT
for p in c d; do
  db=$(find $O/pmc_$p -name "*.db" | head -1)
  python scripts/rocpd_pmc_layers.py $db 6 conv_pk > $O/pmc_${p}_layers.txt 2>&1
  rm -rf $O/pmc_$p
  cat $O/pmc_${p}_layers.txt | cut -c1-250
done
