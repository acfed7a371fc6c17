"""Do a side stream's kernels make progress while the RGB trunk runs on the main stream?
GPU-side event stamps (no profiler): main = the RGB trunk (graph replay or eager), side = N tiny
kernels / the depth trunk / the instruction encoder, both released by one event.

    python scripts/overlap_probe2.py            (VLNCE_HIP_GRAPHS=0 for the eager trunk)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.streams import pick_concurrent_stream  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
net = policy.net
obs = bench.synth_batch(64, 256, 80, dev, seed=1)[0]
main = torch.cuda.current_stream(dev)
side = pick_concurrent_stream(dev)
small = torch.zeros(1024, device=dev)
with torch.no_grad():
    pass
for _ in range(3):   # eager pass, capture pass, replay
    net.rgb_encoder(obs)
    with torch.cuda.stream(side):
        net.depth_encoder(obs)
        net.instruction_encoder(obs)
torch.cuda.synchronize()


def ev():
    return torch.cuda.Event(enable_timing=True)


def tiny(n):
    for _ in range(n):
        small.add_(1.0)


def run(name, side_fn, side_first=False, main_fn=lambda: net.rgb_encoder(obs)):
    for rep in range(2):
        torch.cuda.synchronize()
        e0, a1, b0, b1 = ev(), ev(), ev(), ev()
        e0.record(main)
        side.wait_event(e0)

        def do_side():
            with torch.cuda.stream(side):
                small.add_(1.0)
                b0.record(side)
                side_fn()
                b1.record(side)

        if side_first:
            do_side()
        main_fn()
        a1.record(main)
        if not side_first:
            do_side()
        torch.cuda.synchronize()
    print(f"{name:58s} main done {e0.elapsed_time(a1):6.2f} ms | side first kernel done "
          f"{e0.elapsed_time(b0):6.2f}, side done {e0.elapsed_time(b1):6.2f}", flush=True)


print("graphs:", os.environ.get("VLNCE_HIP_GRAPHS", "1"))
run("side alone: 200 tiny kernels", lambda: tiny(200), main_fn=lambda: None)
run("side alone: depth trunk", lambda: net.depth_encoder(obs), main_fn=lambda: None)
run("RGB trunk | 200 tiny kernels", lambda: tiny(200))
run("RGB trunk | 200 tiny kernels (side issued first)", lambda: tiny(200), side_first=True)
run("RGB trunk | depth trunk", lambda: net.depth_encoder(obs))
run("RGB trunk | depth trunk (side issued first)", lambda: net.depth_encoder(obs), side_first=True)
run("RGB trunk | instruction encoder", lambda: net.instruction_encoder(obs))
a = torch.randn(8192, 8192, device=dev)
run("20 x 8192^3 matmul | 200 tiny kernels", lambda: tiny(200), main_fn=lambda: [a @ a for _ in range(20)])
