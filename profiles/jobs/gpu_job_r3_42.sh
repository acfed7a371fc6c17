#!/bin/bash
# round 3, job 42: one-launch GroupNorm for small activations: tests (both paths), act() latency, policy tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03am
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "group_norm or gn_ or bottleneck or block" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_policy_gpu.py tests/test_obs_transforms.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for v in 0 1; do
  for n in 1 4 8; do echo "GN_SMALL=$v $(VLNCE_GN_SMALL=$v timeout 200 python scripts/act_profile.py --num-envs $n --iters 30 2>/dev/null | tail -1)"; done
done | tee $O/act.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kta -- python $GRAFT_REPO_ROOT/scripts/act_profile.py --num-envs 1 --iters 8 --sync > $O/kta.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_act.py "$(find $O/kta -name '*.db' | head -1)" > $O/act_one_call.txt 2>&1
rm -rf $O/kta
head -14 $O/act_one_call.txt | cut -c1-130
