"""What the CMA tail's captured graphs cost by themselves (GPU otherwise idle), and what a dependent
launch costs inside a HIP graph on this device:
  * forward replay / backward replay of net._tail at 64 environments (events around the replays),
  * a captured chain of N trivial dependent kernels (fill of 64 floats), per kernel.
    python scripts/tail_graph_time.py [num_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vlnce_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
net = policy.net
g = torch.Generator(device="cpu").manual_seed(0)


def mk(*shape, grad=False):
    return torch.randn(*shape, generator=g).to(dev).requires_grad_(grad)


ins, dep, rgb = mk(N, 80, 256, grad=True), mk(N, 16, 192, grad=True), mk(N, 16, 2112, grad=True)
act, h0 = mk(N, 32, grad=True), mk(N, 2, 512)
masks = torch.ones(N, dtype=torch.uint8, device=dev)
static = (2048, 128)


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(timed):
    a, b, c = ev(), ev(), ev()
    a.record()
    x, h = net._tail(ins, dep, rgb, act, h0, masks, static=static)
    b.record()
    (x.sum() + h.sum()).backward()
    c.record()
    if timed:
        torch.cuda.synchronize()
        return a.elapsed_time(b), b.elapsed_time(c)


for _ in range(4):
    run(False)
torch.cuda.synchronize()
fw, bw = zip(*[run(True) for _ in range(20)])
print(f"tail at {N} envs, graphs alone: forward replay (incl. input copies) {sorted(fw)[len(fw) // 2] * 1e3:.0f} us, "
      f"backward (2 sum kernels + their backward + graph replay) {sorted(bw)[len(bw) // 2] * 1e3:.0f} us")
os.environ["VLNCE_HIP_GRAPHS"] = "0"
fw, bw = zip(*[run(True) for _ in range(10)])
print(f"same, eager (host-issued): forward {sorted(fw)[5] * 1e3:.0f} us, backward {sorted(bw)[5] * 1e3:.0f} us")

# ---- dependent trivial kernels inside a captured graph
buf = torch.zeros(64, device=dev)
for n in (50, 200):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(n):
            buf.add_(1.0)
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = ev(), ev()
        a.record()
        gr.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"graph of {n} dependent 64-float adds: {min(ts) * 1e3:.0f} us = {min(ts) * 1e3 / n:.2f} us per kernel")
a, b = ev(), ev()
torch.cuda._sleep(20_000_000)
a.record()
for i in range(200):
    buf.add_(1.0)
b.record()
torch.cuda.synchronize()
print(f"200 dependent adds issued behind a backlog (stream, no graph): {a.elapsed_time(b) * 1e3 / 200:.2f} us per kernel")
