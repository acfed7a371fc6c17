"""Per-layer timing of the implicit-GEMM convolution on the conv shapes the policies issue
 at num_envs frames.  HIP-event timed, eval-style epilogue
(scale/shift/ReLU) or train-style (raw + BatchNorm partial statistics).

    python scripts/convbench.py [--n 64] [--iters 20] [--mode eval|train] [--only substr]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vlnce_amd import ops  # noqa: E402

# name, H(=W) in, Cin, Cout, k, stride, count in the trunk
R50 = [
    ("stem7x7s2_3_64", 256, 3, 64, 7, 2, 1),
    ("stem_s2d_12_64", 131, 12, 64, 4, 1, 0),   # same layer as a 4x4 conv over 2x2 blocks (pad 0)
    ("l1_1x1_64_64", 64, 64, 64, 1, 1, 1),
    ("l1_3x3_64_64", 64, 64, 64, 3, 1, 3),
    ("l1_1x1_64_256", 64, 64, 256, 1, 1, 4),
    ("l1_1x1_256_64", 64, 256, 64, 1, 1, 2),
    ("l2_1x1_256_128", 64, 256, 128, 1, 1, 1),
    ("l2_3x3s2_128_128", 64, 128, 128, 3, 2, 1),
    ("l2_1x1s2_256_512", 64, 256, 512, 1, 2, 1),
    ("l2_1x1_128_512", 32, 128, 512, 1, 1, 4),
    ("l2_1x1_512_128", 32, 512, 128, 1, 1, 3),
    ("l2_3x3_128_128", 32, 128, 128, 3, 1, 3),
    ("l3_1x1_512_256", 32, 512, 256, 1, 1, 1),
    ("l3_3x3s2_256_256", 32, 256, 256, 3, 2, 1),
    ("l3_1x1s2_512_1024", 32, 512, 1024, 1, 2, 1),
    ("l3_1x1_256_1024", 16, 256, 1024, 1, 1, 6),
    ("l3_1x1_1024_256", 16, 1024, 256, 1, 1, 5),
    ("l3_3x3_256_256", 16, 256, 256, 3, 1, 5),
    ("l4_1x1_1024_512", 16, 1024, 512, 1, 1, 1),
    ("l4_3x3s2_512_512", 16, 512, 512, 3, 2, 1),
    ("l4_1x1s2_1024_2048", 16, 1024, 2048, 1, 2, 1),
    ("l4_1x1_512_2048", 8, 512, 2048, 1, 1, 3),
    ("l4_1x1_2048_512", 8, 2048, 512, 1, 1, 2),
    ("l4_3x3_512_512", 8, 512, 512, 3, 1, 2),
]


# habitat GroupNorm ResNet-50 (base 32) on the 128x128 pooled depth frame + compression conv
DEPTH = [
    ("d1_1x1_32_32", 32, 32, 32, 1, 1, 1),
    ("d1_3x3_32_32", 32, 32, 32, 3, 1, 3),
    ("d1_1x1_32_128", 32, 32, 128, 1, 1, 4),
    ("d1_1x1_128_32", 32, 128, 32, 1, 1, 2),
    ("d2_1x1_128_64", 32, 128, 64, 1, 1, 1),
    ("d2_3x3s2_64_64", 32, 64, 64, 3, 2, 1),
    ("d2_1x1s2_128_256", 32, 128, 256, 1, 2, 1),
    ("d2_1x1_64_256", 16, 64, 256, 1, 1, 4),
    ("d2_1x1_256_64", 16, 256, 64, 1, 1, 3),
    ("d2_3x3_64_64", 16, 64, 64, 3, 1, 3),
    ("d3_1x1_256_128", 16, 256, 128, 1, 1, 1),
    ("d3_3x3s2_128_128", 16, 128, 128, 3, 2, 1),
    ("d3_1x1s2_256_512", 16, 256, 512, 1, 2, 1),
    ("d3_1x1_128_512", 8, 128, 512, 1, 1, 6),
    ("d3_1x1_512_128", 8, 512, 128, 1, 1, 5),
    ("d3_3x3_128_128", 8, 128, 128, 3, 1, 5),
    ("d4_1x1_512_256", 8, 512, 256, 1, 1, 1),
    ("d4_3x3s2_256_256", 8, 256, 256, 3, 2, 1),
    ("d4_1x1s2_512_1024", 8, 512, 1024, 1, 2, 1),
    ("d4_1x1_256_1024", 4, 256, 1024, 1, 1, 3),
    ("d4_1x1_1024_256", 4, 1024, 256, 1, 1, 2),
    ("d4_3x3_256_256", 4, 256, 256, 3, 1, 2),
    ("dc_3x3_1024_128", 4, 1024, 128, 3, 1, 1),
]
# torchvision ResNet-18 (Waypoint RGB encoder), 3x3 layers only (the 1x1 downsamples are 1.2 %)
R18 = [
    ("r18_3x3_64_64", 64, 64, 64, 3, 1, 4),
    ("r18_3x3s2_64_128", 64, 64, 128, 3, 2, 1),
    ("r18_3x3_128_128", 32, 128, 128, 3, 1, 3),
    ("r18_3x3s2_128_256", 32, 128, 256, 3, 2, 1),
    ("r18_3x3_256_256", 16, 256, 256, 3, 1, 3),
    ("r18_3x3s2_256_512", 16, 256, 512, 3, 2, 1),
    ("r18_3x3_512_512", 8, 512, 512, 3, 1, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="eval")
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=3, help="timed rounds per layer; the MIN is reported")
    ap.add_argument("--set", default="r50", help="r50 | depth | r18 (comma separated)")
    ap.add_argument("--pro", action="store_true",
                    help="train mode: also apply the previous layer's BatchNorm + ReLU in the operand "
                         "loader, as the trunks do (in_scale / in_shift / in_center / in_relu)")
    ap.add_argument("--dual", default="", help="identity | bn: 1x1 stride-1 layers as residual block "
                    "ends (second input added in the operand loader, block output written)")
    ap.add_argument("--backlog", action="store_true",
                    help="issue each timed round behind a spin kernel: GPU-paced times for launches "
                         "shorter than the host's issue cost")
    ap.add_argument("--opt", default="", help="dispatch options, e.g. u3=2,s3=0 (vlnce_set_option)")
    ap.add_argument("--rotate", type=int, default=1,
                    help="R copies of the layer's input (and second input / block output) used in "
                         "rotation: with R x the input bytes beyond the 256 MB Infinity Cache every "
                         "launch reads its operands from HBM, as inside a trunk (the weights stay hot)")
    args = ap.parse_args()
    for kv in filter(None, args.opt.split(",")):
        k_, v_ = kv.split("=")
        ops.L().set_option(k_, int(v_))
    dev = "cuda:0"
    tot_t = tot_f = 0.0
    print(f"{'layer':22s} {'M':>8s} {'K':>6s} {'N':>5s} {'us':>9s} {'TF/s':>7s} x cnt")
    layers = []
    for nm in args.set.split(","):
        layers += {"r50": R50, "depth": DEPTH, "r18": R18}[nm]
    for name, hw, cin, cout, k, s, cnt in layers:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        x = torch.randn(args.n, hw, hw, cin, device=dev)
        w = torch.randn(cout, k, k, cin, device=dev) * (cin * k * k) ** -0.5
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev)
        pad = k // 2 if k % 2 else 0
        kw = dict(want_stats=True) if args.mode == "train" else dict(scale=sc, shift=sh, act=1)
        if args.pro and args.mode == "train" and cin % 32 == 0:
            kw.update(in_scale=torch.rand(cin, device=dev) + 0.5, in_shift=torch.randn(cin, device=dev),
                      in_center=torch.randn(cin, device=dev), in_relu=True)
        if args.dual and k == 1 and s == 1 and "in_scale" in kw:
            kw.update(x2=torch.randn_like(x), side_out=torch.empty_like(x))
            if args.dual == "bn":
                kw.update(in2_scale=torch.rand(cin, device=dev) + 0.5,
                          in2_shift=torch.randn(cin, device=dev), in2_center=torch.randn(cin, device=dev))
        elif args.dual:
            continue
        if cin == 3:
            kw.update(in_scale=torch.full((3,), 1 / 255.0, device=dev),
                      in_shift=torch.zeros(3, device=dev))
        xs = [x] + [x.clone() for _ in range(args.rotate - 1)]
        kws = [kw] + [dict(kw, **{k_: v_.clone() for k_, v_ in kw.items() if k_ in ("x2", "side_out")})
                      for _ in range(args.rotate - 1)]
        for _ in range(3):
            ops.conv2d_nhwc(x, w, s, pad, **kw)
        us = 1e30
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if args.backlog:
                torch.cuda._sleep(int(4e7))
            e0.record()
            for it_ in range(args.iters):
                ops.conv2d_nhwc(xs[it_ % args.rotate], w, s, pad, **kws[it_ % args.rotate])
            e1.record()
            torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / args.iters)
        ho = (hw + 2 * pad - k) // s + 1
        M = args.n * ho * ho
        fl = 2.0 * M * cin * k * k * cout
        print(f"{name:22s} {M:8d} {cin*k*k:6d} {cout:5d} {us:9.1f} {fl/us/1e6:7.1f} x{cnt}")
        tot_t += us * cnt
        tot_f += fl * cnt
    print(f"trunk total {tot_t/1e3:.3f} ms  ->  {tot_f/tot_t/1e6:.1f} TF/s  ({args.mode})")


if __name__ == "__main__":
    main()
