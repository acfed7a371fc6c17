"""Per layer of the CMA tail at 64 rows: GPU time of the forward and of the backward of ops.linear through
the skinny-linear kernels (VLNCE_LINEAR_ROWS=1) and through the general GEMM (=0), each measured as a
captured graph of REPS dependent repetitions (what the tail's graphs are made of).
    python scripts/linear_rows_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vlnce_amd import ops  # noqa: E402
from vlnce_amd.streams import capture_guard  # noqa: E402

KEEP = []   # graphs stay alive: destroying one while another capture is open aborts the process

dev = torch.device("cuda:0")
REPS = 20
LAYERS = [("rgb_linear", 64, 256, 2112, 1), ("depth_linear", 64, 128, 3072, 1), ("gru1 ih", 64, 1536, 416, 0),
          ("gru hh", 64, 1536, 512, 0), ("state_q", 64, 256, 512, 0), ("text_q", 64, 256, 256, 0),
          ("compress", 64, 512, 1184, 1), ("gru2 ih", 64, 1536, 512, 0)]


def timed_graph(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    KEEP.append(g)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):   # warm-up off the default stream, as make_graphed_callables does
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with capture_guard(), torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts) * 1e3 / REPS


print(f"{'layer':14s} {'M':>4s} {'N':>5s} {'K':>5s}   fwd us (rows | gemm)   bwd us (rows | gemm)")
tot = [0.0, 0.0, 0.0, 0.0]
for name, M, N, K, act in LAYERS:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).requires_grad_(True)
    b = torch.randn(N, device=dev, requires_grad=True)
    gy = torch.randn(M, N, device=dev)
    res = []
    for mode in ("1", "0"):
        os.environ["VLNCE_LINEAR_ROWS"] = mode
        with torch.no_grad():
            f = timed_graph(lambda: ops.linear(x, w, b, act))
        with torch.no_grad():
            y = ops.linear(x, w, b, act)
        ctx = type("Ctx", (), {})()   # LinearFn.backward called directly: no autograd engine in the capture
        ctx.saved_tensors = (x.detach(), w.detach(), y if act else None)
        ctx.act, ctx.has_bias, ctx.dx_from, ctx.needs_input_grad = act, True, 0, (True, True, True, False, False)
        ctx.rows = mode == "1"

        def bwd():
            ops.LinearFn.backward(ctx, gy)

        res += [f, timed_graph(bwd)]
    print(f"{name:14s} {M:4d} {N:5d} {K:5d}   {res[0]:8.1f} | {res[2]:6.1f}      {res[1]:8.1f} | {res[3]:6.1f}")
    for i in range(4):
        tot[i] += res[i]
print(f"{'sum':31s}   {tot[0]:8.1f} | {tot[2]:6.1f}      {tot[1]:8.1f} | {tot[3]:6.1f}")
