"""Does work the host enqueues BEHIND the RGB trunk slow the trunk down?  GPU event stamps of the
RGB trunk (graph replay on the main stream) while the host (a) waits, (b) enqueues 200 tiny kernels
behind it on the same stream, (c) the same on a side stream that waits for the trunk's end event,
(d) replays the depth-trunk graph on a side stream ordered BEHIND the RGB trunk.
    python scripts/overlap_probe3.py"""
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
for _ in range(3):
    net.rgb_encoder(obs)
    with torch.cuda.stream(side):
        net.depth_encoder(obs)
torch.cuda.synchronize()


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(name, after):
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record(main)
        net.rgb_encoder(obs)
        e1.record(main)
        after(e1)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name:70s} RGB trunk {min(ts):6.2f} .. {max(ts):6.2f} ms", flush=True)


def tiny_main(e1):
    for _ in range(200):
        small.add_(1.0)


def tiny_side(e1):
    side.wait_event(e1)
    with torch.cuda.stream(side):
        for _ in range(200):
            small.add_(1.0)


def depth_behind(e1):
    side.wait_event(e1)
    with torch.cuda.stream(side):
        net.depth_encoder(obs)


def events_only(e1):
    for _ in range(100):
        e = torch.cuda.Event()
        e.record(main)
        side.wait_event(e)
        e2 = torch.cuda.Event()
        e2.record(side)
        main.wait_event(e2)


run("host waits", lambda e1: None)
run("200 tiny kernels enqueued behind it, same stream", tiny_main)
run("200 tiny kernels on a side stream that waits for the trunk's end", tiny_side)
run("depth-trunk graph on a side stream that waits for the trunk's end", depth_behind)
run("100 cross-stream event record/wait pairs behind it", events_only)
run("host waits (again)", lambda e1: None)
