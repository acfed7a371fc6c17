#!/bin/bash
# round 4, job 13: RCCL single-rank check incl. trainable encoders (buckets issued from the hooks);
# what RCCL reports about itself (NCCL_DEBUG=INFO, 1 rank) under bench.py --force-dist
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04m; mkdir -p $out
timeout 600 python tests/rccl_single_rank_check.py > $out/rccl_single_rank.txt 2>&1; echo "rccl check rc=$?"; tail -3 $out/rccl_single_rank.txt
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,TUNING,GRAPH timeout 300 python bench.py --force-dist --steps 3 --warmup 2 --no-cpu-baseline --no-f32-compare --no-pipeline > $out/bench_force_dist.json 2> $out/rccl_debug_info.txt
grep -E "NCCL INFO" $out/rccl_debug_info.txt | head -60 | cut -c1-220
tail -1 $out/bench_force_dist.json | cut -c1-200
