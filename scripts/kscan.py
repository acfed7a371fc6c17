"""T(K) scan of the igemm kernel at fixed M, N: separates per-tile fixed cost from the
steady-state K-loop rate.  python scripts/kscan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlnce_amd import ops
dev = "cuda:0"
for (n, hw, cout) in [(64, 16, 1024), (64, 32, 256), (64, 8, 2048)]:
    for cin in (64, 128, 256, 512, 1024, 2048, 4096):
        x = torch.randn(n, hw, hw, cin, device=dev); w = torch.randn(cout, 1, 1, cin, device=dev) * cin ** -0.5
        for _ in range(3): ops.conv2d_nhwc(x, w, 1, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_nhwc(x, w, 1, 0)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        M = n * hw * hw
        print(f"M={M:6d} N={cout:5d} K={cin:5d}  {us:8.1f} us  {2.0*M*cin*cout/us/1e6:6.1f} TF/s")
