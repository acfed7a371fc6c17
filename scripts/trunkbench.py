"""Stand-alone timing of the two frozen visual trunks (graph replay, one stream each):
    python scripts/trunkbench.py [--n 64] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vlnce_amd

ap = argparse.ArgumentParser()
ap.add_argument("--n", default="1,8,64")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
for n in [int(v) for v in a.n.split(",")]:
    obs = {"rgb": torch.randint(0, 256, (n, 256, 256, 3), device=dev).float(),
           "depth": torch.rand(n, 256, 256, 1, device=dev)}
    for name, enc in (("rgb", pol.net.rgb_encoder), ("depth", pol.net.depth_encoder)):
        with torch.no_grad():
            for _ in range(3):
                enc.trunk_features(obs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                enc.trunk_features(obs)
            e1.record()
            torch.cuda.synchronize()
        print(f"N={n:3d} {name:5s} trunk: {e0.elapsed_time(e1) / a.iters:8.3f} ms", flush=True)
