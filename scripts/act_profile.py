"""Forward-only act() at a small batch (eval, no_grad), as bench.py's act_latency measures it: the
thing to put under `rocprofv3 --kernel-trace` to see what one call consists of
(scripts/rocpd_act.py prints the per-call launch list).

    python scripts/act_profile.py [--num-envs 1] [--iters 12] [--sync]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--num-envs", type=int, default=1)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--sync", action="store_true", help="synchronize after every call (pure latency)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
policy.eval()
obs, prev, masks = bench.synth_batch(8, 256, 80, dev, seed=1)[:3]
n = args.num_envs
o = {k: v[:n].contiguous() for k, v in obs.items()}
h0 = torch.zeros(n, policy.net.num_recurrent_layers, 512, device=dev)
with torch.no_grad():
    for _ in range(5):
        policy.act(o, h0, prev[:n], masks[:n], deterministic=True)
    torch.cuda.synchronize()
    marker = torch.zeros(1, device=dev)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        marker.add_(1.0)  # delimiter launch for the trace
        policy.act(o, h0, prev[:n], masks[:n], deterministic=True)
        if args.sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
print(f"act() at num_envs={n}: {1e3 * (time.perf_counter() - t0) / args.iters:.3f} ms per call"
      f"{' (synchronized)' if args.sync else ''}")

g = policy.__dict__.get("_act_graph")
if g is not None:
    ents = [v for v in g.entries.values() if isinstance(v, list)]
    if ents:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            ents[-1][0].replay()
        torch.cuda.synchronize()
        print(f"bare replay of the captured act() graph: {1e3 * (time.perf_counter() - t0) / args.iters:.3f} ms")
