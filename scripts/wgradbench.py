"""Per-layer timing of vlnce_conv2d_wgrad (weight gradient of the trunks' convolutions, trainable
encoders) on the RGB ResNet-50 / depth layer shapes of scripts/convbench.py at num_envs frames:
the plane kernel (wgrad_x6_kernel, three bf16 planes on the 16-bit pipe) and, with
--opt wgrad_tile=1, the fp32-MFMA kernel.

    python scripts/wgradbench.py [--n 64] [--set r50|depth] [--only substr] [--opt wgrad_tile=1]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from convbench import DEPTH, R50  # noqa: E402
from vlnce_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--set", default="r50")
    ap.add_argument("--only", default="")
    ap.add_argument("--opt", default="")
    ap.add_argument("--f16", action="store_true", help="fp16 planes with dy's power of two")
    args = ap.parse_args()
    for kv in filter(None, args.opt.split(",")):
        k_, v_ = kv.split("=")
        ops.L().set_option(k_, int(v_))
    dev = "cuda:0"
    tot_t = tot_f = 0.0
    print(f"{'layer':22s} {'Cout':>5s} {'K':>6s} {'pixels':>8s} {'us':>9s} {'TF/s':>7s} x cnt")
    for name, hw, cin, cout, k, s, cnt in {"r50": R50, "depth": DEPTH}[args.set]:
        if cnt == 0 or (args.only and not any(o in name for o in args.only.split(","))):
            continue
        pad = k // 2 if k % 2 else 0
        x = torch.randn(args.n, hw, hw, cin, device=dev)
        g = ops.conv_geometry(x, torch.empty(cout, k, k, cin), s, pad)
        dy = torch.randn(args.n, g["Ho"], g["Wo"], cout, device=dev) * 1e-3
        dw = torch.empty(cout, k, k, cin, device=dev)
        pow2 = None
        if args.f16:
            import math
            up = 2.0 ** (14 - math.frexp(float(dy.abs().max()))[1])
            pow2 = torch.stack([torch.full((8,), up), torch.full((8,), 1.0 / up)]).to(dev)
        for _ in range(2):
            ops.L().conv2d_wgrad(x, dy, dw, g, pow2)
        us = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(int(2e7))
            e0.record()
            for _ in range(args.iters):
                ops.L().conv2d_wgrad(x, dy, dw, g, pow2)
            e1.record()
            torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / args.iters)
        M = args.n * g["Ho"] * g["Wo"]
        fl = 2.0 * M * cin * k * k * cout
        print(f"{name:22s} {cout:5d} {cin*k*k:6d} {M:8d} {us:9.1f} {fl/us/1e6:7.1f} x{cnt}")
        tot_t += us * cnt
        tot_f += fl * cnt
    print(f"total {tot_t/1e3:.3f} ms  ->  {tot_f/tot_t/1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
