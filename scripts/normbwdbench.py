"""BatchNorm / GroupNorm backward at the trunk's layer shapes (batch 64): time and the HBM rate over
the ALGORITHMIC bytes (dy, y, x read twice; dx (+dres) written once).

    python scripts/normbwdbench.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vlnce_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
hip = _lib.get_lib()


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = 0.0
print("BatchNorm backward (RGB trunk, 64 x 224 x 224)")
for hw, Cc, res in [(112, 64, 0), (56, 64, 0), (56, 256, 1), (28, 128, 0), (28, 512, 1), (14, 256, 0),
                    (14, 1024, 1), (7, 512, 0), (7, 2048, 1)]:
    M = 64 * hw * hw
    t = [torch.randn(M, Cc, device=dev) for _ in range(3)]
    v = [torch.randn(Cc, device=dev).abs() + 0.5 for _ in range(3)]
    dx, dres = torch.empty(M, Cc, device=dev), (torch.empty(M, Cc, device=dev) if res else None)
    dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    ws = torch.empty(hip.bn_bwd_workspace_floats(M, Cc), device=dev)
    ms = timed(lambda: hip.bn_bwd(t[0], t[1], t[2], v[0], v[1], v[2], M, Cc, 1, 1, dx, dres, dg, db, ws))
    gb = M * Cc * 4 * (6 + 1 + res) / 1e9
    print(f"  M={M:7d} C={Cc:5d} res={res}: {ms*1e3:8.1f} us  {gb/ms:7.2f} GB/ms")
print("GroupNorm backward (depth trunk, 64 x 256 x 256)")
for hw, Cc, res in [(128, 32, 0), (64, 32, 0), (64, 128, 1), (32, 64, 0), (32, 256, 1), (16, 128, 0),
                    (16, 512, 1), (8, 256, 0), (8, 1024, 1)]:
    N, HW, G = 64, hw * hw, 16
    t = [torch.randn(N, HW, Cc, device=dev) for _ in range(3)]
    mean, rstd = torch.randn(N, G, device=dev), torch.rand(N, G, device=dev) + 0.5
    gamma = torch.randn(Cc, device=dev)
    dx, dres = torch.empty_like(t[0]), (torch.empty_like(t[0]) if res else None)
    dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    ws = torch.empty(hip.gn_bwd_workspace_floats(N, HW, Cc, G), device=dev)
    ms = timed(lambda: hip.gn_bwd(t[0], t[1], t[2], mean, rstd, gamma, N, HW, Cc, G, 1, dx, dres, dg, db, ws))
    gb = N * HW * Cc * 4 * (6 + 1 + res) / 1e9
    print(f"  N*HW={N*HW:7d} C={Cc:5d} res={res}: {ms*1e3:8.1f} us  {gb/ms:7.2f} GB/ms")
