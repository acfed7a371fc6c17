"""The plain-loop update with the trunk outputs fed through rgb_features / depth_features (what is
left of a step outside the visual trunks), for `rocprofv3 --kernel-trace`; scripts/rocpd_seq.py
lists the kernels of one step in time order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
vlnce_amd.AuxLosses.activate()
batches = []
for i in range(4):
    obs, prev, masks, tgt, w = bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i)
    with torch.no_grad():
        obs = dict(obs)
        obs["rgb_features"] = policy.net.rgb_encoder.trunk_features(obs).clone()
        obs["depth_features"] = policy.net.depth_encoder.trunk_features(obs).clone()
    batches.append((obs, prev, masks, tgt, w))
for i in range(18):
    obs, prev, masks, tgt, w = batches[i % 4]
    update_agent(policy, opt, obs, prev, masks, tgt, w, 512)
torch.cuda.synchronize()
print("done")
