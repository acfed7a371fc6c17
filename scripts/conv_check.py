"""Compare ops.conv2d_nhwc with torch's fp64 conv on one shape and print where they differ.
    python scripts/conv_check.py N H Cin Cout k stride [stats]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vlnce_amd import ops

N, H, Cin, Cout, k, s = map(int, sys.argv[1:7])
stats = len(sys.argv) > 7
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn(N, H, H, Cin, device=dev)
w = torch.randn(Cout, k, k, Cin, device=dev) * (Cin * k * k) ** -0.5
pad = k // 2
out = ops.conv2d_nhwc(x, w, s, pad, want_stats=stats)
y = out[0] if isinstance(out, tuple) else out
ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=s, padding=pad).permute(0, 2, 3, 1)
d = (y.double() - ref).abs().reshape(-1, Cout)
bad = (d > 1e-3) | torch.isnan(d)
print("shape M", d.shape[0], "N", Cout, "K", Cin * k * k, "max err", d.max().item(), "nan", torch.isnan(d).sum().item(), "bad", bad.sum().item())
if bad.any():
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("bad rows:", rows[:16].tolist(), "...", rows[-4:].tolist(), "count", rows.numel())
    print("bad cols:", cols[:16].tolist(), "...", cols[-4:].tolist(), "count", cols.numel())
    r0 = rows[0].item()
    print("row", r0, "got", y.reshape(-1, Cout)[r0, :6].tolist(), "want", ref.reshape(-1, Cout)[r0, :6].tolist())
