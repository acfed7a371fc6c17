"""The plain trainer loop of bench.py (one _update_agent per batch, no encode_ahead) and nothing
else: the thing to put under `rocprofv3 --kernel-trace` for a per-step kernel breakdown
(scripts/rocpd_stats.py, scripts/rocpd_timeline.py).

    python scripts/step_profile.py [--steps 12] [--warmup 6] [--num-envs 64]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--warmup", type=int, default=6)
ap.add_argument("--num-envs", type=int, default=64)
ap.add_argument("--trainable-encoders", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
over = {"RGB_ENCODER.trainable": True, "DEPTH_ENCODER.trainable": True} if args.trainable_encoders else {}
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy", **over), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
vlnce_amd.AuxLosses.activate()
batches = [bench.synth_batch(args.num_envs, 256, 80, dev, seed=1 + 101 * i) for i in range(4)]
for i in range(args.warmup + args.steps):
    obs, prev, masks, tgt, w = batches[i % 4]
    update_agent(policy, opt, obs, prev, masks, tgt, w, 512)
    if i == args.warmup - 1:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print("done")
