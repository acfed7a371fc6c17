"""Error of the implicit-GEMM convolution against an fp64 reference, per trunk layer shape.

Reports max |y - y64| / max|y64| and the rms relative error for the library selected by
VLNCE_HIP_LIB (default: the in-tree build), next to the same figures for torch's fp32 conv
(MIOpen / rocBLAS) so the number has a yardstick.

    python scripts/conv_accuracy.py [--n 4]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vlnce_amd import ops  # noqa: E402

SHAPES = [  # name, H, Cin, Cout, k, stride
    ("l1_1x1_64_256", 64, 64, 256, 1, 1),
    ("l1_3x3_64_64", 64, 64, 64, 3, 1),
    ("l2_3x3s2_128_128", 64, 128, 128, 3, 2),
    ("l3_1x1_1024_256", 16, 1024, 256, 1, 1),
    ("l3_3x3_256_256", 16, 256, 256, 3, 1),
    ("l4_3x3_512_512", 8, 512, 512, 3, 1),
    ("l4_1x1_2048_512", 8, 2048, 512, 1, 1),
    ("l3_1x1_256_1024", 16, 256, 1024, 1, 1),
    ("l2_3x3_128_128", 32, 128, 128, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    fam = {0: "fp32-mfma", 1: "planes-x3", 2: "planes-p3/u3"}
    print(f"{'layer':20s} {'kernel':>13s} {'K':>6s} {'max_rel(ours)':>14s} {'rms_rel(ours)':>14s} {'max_rel(torch)':>15s} {'rms_rel(torch)':>15s}")
    for name, hw, cin, cout, k, s in SHAPES:
        # activations with a wide dynamic range (post-ReLU like) and weights of mixed magnitude
        x = torch.randn(args.n, hw, hw, cin, device=dev).relu_() * torch.exp(2 * torch.randn(cin, device=dev))
        w = torch.randn(cout, k, k, cin, device=dev) * (cin * k * k) ** -0.5
        pad = k // 2
        y = ops.conv2d_nhwc(x, w, s, pad)
        path = fam.get(ops.L().conv2d_last_path(), "?")
        x64 = x.double().permute(0, 3, 1, 2)
        w64 = w.double().permute(0, 3, 1, 2)
        y64 = F.conv2d(x64, w64, stride=s, padding=pad).permute(0, 2, 3, 1)
        yt = F.conv2d(x64.float(), w64.float(), stride=s, padding=pad).permute(0, 2, 3, 1)
        den = y64.abs().max()
        rms = y64.pow(2).mean().sqrt()

        def err(a):
            d = a.double() - y64
            return (d.abs().max() / den).item(), (d.pow(2).mean().sqrt() / rms).item()

        eo, et = err(y), err(yt)
        print(f"{name:20s} {path:>13s} {cin*k*k:6d} {eo[0]:14.3e} {eo[1]:14.3e} {et[0]:15.3e} {et[1]:15.3e}")


if __name__ == "__main__":
    main()
