"""act() latency at 1 / 4 environments with one or both visual trunks taken out of the call (their
features handed over precomputed, as the cached-feature DAgger path does): which chain of
launches paces the forward-only step.

    python scripts/act_chain_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = vlnce_amd.make_config("CMAPolicy")
pol = vlnce_amd.build_model(cfg, *vlnce_amd.make_spaces(256, 256)).to(dev)
pol.eval()
batch = bench.synth_batch(8, 256, 80, dev, seed=1)
obs, prev, masks = batch[0], batch[1], batch[2]


def timed(o, n, iters=40):
    h0 = torch.zeros(n, pol.net.num_recurrent_layers, 512, device=dev)
    with torch.no_grad():
        for _ in range(5):
            pol.act(o, h0, prev[:n], masks[:n], deterministic=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            pol.act(o, h0, prev[:n], masks[:n], deterministic=True)
        torch.cuda.synchronize()
        back_to_back = 1e3 * (time.perf_counter() - t0) / iters
        lat = []
        for _ in range(iters):   # one call at a time: issue + execution, nothing to hide behind
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pol.act(o, h0, prev[:n], masks[:n], deterministic=True)
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - t0))
    return back_to_back, sorted(lat)[len(lat) // 2]


for n in (1, 4):
    o = {k: v[:n].contiguous() for k, v in obs.items()}
    with torch.no_grad():
        ahead = pol.encode_ahead(o)
        torch.cuda.synchronize()
    feats = {k: ahead[k] for k in ("rgb_features", "depth_features") if k in ahead}
    rows = [("both trunks in the call", o)]
    if "depth_features" in feats:
        rows.append(("depth features given", dict(o, depth_features=feats["depth_features"])))
    if "rgb_features" in feats:
        rows.append(("rgb features given", dict(o, rgb_features=feats["rgb_features"])))
    if len(feats) == 2:
        rows.append(("both given (tail + instruction only)", dict(o, **feats)))
    for name, oo in rows:
        b2b, one = timed(oo, n)
        print(f"n={n} {name:40s} back-to-back {b2b:6.3f} ms   one call at a time {one:6.3f} ms")
