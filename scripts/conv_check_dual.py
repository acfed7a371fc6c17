"""Dual-input prologue check: x' = relu((x-c)s+t + (x2-c2)s2+t2); y = conv1x1(x'), side_out = x'.
    python scripts/conv_check_dual.py N H Cin Cout
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlnce_amd import ops

N, H, Cin, Cout = map(int, sys.argv[1:5])
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn(N, H, H, Cin, device=dev)
x2 = torch.randn(N, H, H, Cin, device=dev)
w = torch.randn(Cout, 1, 1, Cin, device=dev) * Cin ** -0.5
v = [torch.randn(Cin, device=dev) * 0.3 for _ in range(6)]
s1, s2 = v[0].abs() + 0.5, v[3].abs() + 0.5
side = torch.zeros_like(x)
y = ops.conv2d_nhwc(x, w, 1, 0, in_scale=s1, in_shift=v[1], in_center=v[2], in_relu=True, x2=x2,
                    in2_scale=s2, in2_shift=v[4], in2_center=v[5], side_out=side)
xp = torch.relu((x.double() - v[2].double()) * s1.double() + v[1].double() + (x2.double() - v[5].double()) * s2.double() + v[4].double())
ref = xp.reshape(-1, Cin) @ w.reshape(Cout, Cin).double().t()
for name, got, want in (("side_out", side.reshape(-1, Cin), xp.reshape(-1, Cin)), ("y", y.reshape(-1, Cout), ref)):
    d = (got.double() - want).abs()
    bad = (d > 1e-3) | torch.isnan(d)
    print(name, "max err", d.max().item(), "bad", bad.sum().item(), "of", d.numel())
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("  bad rows:", rows[:12].tolist(), "...", rows[-4:].tolist(), "count", rows.numel())
        print("  bad cols:", cols[:12].tolist(), "...", cols[-4:].tolist(), "count", cols.numel())
