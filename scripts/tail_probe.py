"""What a plain-loop step costs outside the visual trunks (no profiler): the same update with the
trunk outputs fed through the reference's own bypass keys (rgb_features / depth_features), and
GPU event stamps at the phase boundaries of one full step.

    python scripts/tail_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.aux_losses import AuxLosses  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
AuxLosses.activate()
batches = [bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i) for i in range(4)]


def loop(n, bs):
    for i in range(n):
        obs, prev, masks, tgt, w = bs[i % 4]
        update_agent(policy, opt, obs, prev, masks, tgt, w, 512)


def timed(name, bs, n=20):
    loop(6, bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(n, bs)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)


timed("full step", batches)
cached = []
for obs, prev, masks, tgt, w in batches:
    with torch.no_grad():
        o = dict(obs)
        o["rgb_features"] = policy.net.rgb_encoder.trunk_features(obs).clone()
        o["depth_features"] = policy.net.depth_encoder.trunk_features(obs).clone()
    cached.append((o, prev, masks, tgt, w))
timed("trunk outputs cached (tail + instruction + loss + backward + Adam)", cached)
os.environ["VLNCE_SIDE_STREAMS"] = "0"
timed("same, one stream", cached)
os.environ["VLNCE_SIDE_STREAMS"] = "1"

# phase stamps of a full step on the main stream
def ev():
    return torch.cuda.Event(enable_timing=True)


for rep in range(3):
    obs, prev, masks, tgt, w = batches[rep]
    torch.cuda.synchronize()
    e = [ev() for _ in range(6)]
    e[0].record()
    T, N = tgt.size()
    h = torch.zeros(N, policy.net.num_recurrent_layers, 512, device=dev)
    AuxLosses.clear()
    dist = policy.build_distribution(obs, h, prev, masks)
    e[1].record()
    logits = dist.logits.view(T, N, -1)
    al = F.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    al = ((w * al).sum(0) / w.sum(0)).mean()
    loss = al + AuxLosses.reduce((w > 0).view(-1))
    e[2].record()
    loss.backward()
    e[3].record()
    opt.step()
    e[4].record()
    opt.zero_grad()
    e[5].record()
    torch.cuda.synchronize()
    names = ["build_distribution", "loss", "backward", "Adam", "zero_grad"]
    print("phases (ms): " + ", ".join(f"{n} {e[i].elapsed_time(e[i + 1]):.3f}" for i, n in enumerate(names))
          + f" | total {e[0].elapsed_time(e[5]):.3f}", flush=True)
