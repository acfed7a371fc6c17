#!/bin/bash
# round-4 evidence on the current build: rocprofv3 kernel stats of the bench command, PMC HBM traffic
# (separate FETCH_SIZE / WRITE_SIZE passes over `bench.py --pmc-step`), then the default bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04zz
mkdir -p $O
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --pmc-step > $O/pmc_$c.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_pmc.py "$(find $O/pmc_$c -name '*.db' | head -1)" > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
python scripts/pmc_traffic_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt profiles/archive/r04_pmc_traffic.json "profiles/archive/r04_zz_pmc_fetch_size.txt, r04_zz_pmc_write_size.txt" | cut -c1-300
cp profiles/archive/r04_pmc_traffic.json $O/r04_pmc_traffic.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py "$(find $O/kt -name '*.db' | head -1)" $O/bench_kernel_stats.md 900 > /dev/null
rm -rf $O/kt
head -30 $O/bench_kernel_stats.md
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r['bf16_pipe']['frac'], r['traffic'], r['launches_per_step'])
P
timeout 300 python scripts/conv_launch_times.py > $O/conv_launch_times.txt 2>/dev/null
head -2 $O/conv_launch_times.txt
