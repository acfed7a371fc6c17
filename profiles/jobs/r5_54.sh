#!/bin/bash
# the two speeds of the same loop (9.25 / 9.65 ms): does the issuing thread's CPU / NUMA node decide?
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_54
mkdir -p $O
cd $GRAFT_REPO_ROOT
{
lscpu | grep -E "NUMA|Model name|Socket|Thread|Core"
for c in /sys/class/drm/card*/device/numa_node; do echo "$c $(cat $c)"; done
for c in /sys/class/kfd/kfd/topology/nodes/*/properties; do echo "$c $(grep -E 'cpu_cores_count|simd_count|domain|location_id' $c | tr '\n' ' ')"; done 2>>$O/err.txt | head -20
nproc; taskset -p $$
N0=$(lscpu | grep "NUMA node0 CPU" | awk '{print $NF}')
N1=$(lscpu | grep "NUMA node1 CPU" | awk '{print $NF}')
echo "node0 cpus $N0; node1 cpus $N1"
for rep in 1 2 3; do
  echo "--- free"; timeout 200 python scripts/step_jitter.py 100 2>>$O/err.txt
  echo "--- node0"; timeout 200 taskset -c $N0 python scripts/step_jitter.py 100 2>>$O/err.txt
  if [ -n "$N1" ]; then echo "--- node1"; timeout 200 taskset -c $N1 python scripts/step_jitter.py 100 2>>$O/err.txt; fi
done
} 2>&1 | tee $O/numa.txt
