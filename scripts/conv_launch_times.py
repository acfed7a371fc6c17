"""Per-launch time of every convolution of the two trunks' forward (train-mode BatchNorm, eager,
one stream, behind a device-side backlog like bench.py's roofline timing), grouped by shape.
    python scripts/conv_launch_times.py [--n 64]
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VLNCE_HIP_GRAPHS"] = "0"
os.environ["VLNCE_SIDE_STREAMS"] = "0"
import torch  # noqa: E402

import vlnce_amd  # noqa: E402
from vlnce_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
obs = {"rgb": torch.randint(0, 256, (a.n, 256, 256, 3), device=dev).float(),
       "depth": torch.rand(a.n, 256, 256, 1, device=dev)}
orig, orig_bn, orig_stem = ops.conv2d_nhwc, ops.conv2d_bn_sums, ops.stem7
PATHS = {0: "igemm", 1: "x3", 2: "p3/u3/s3", 3: "m3", 9: "stem7"}
rec = []


def _rec(call, M, K, N, d):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = call()
    e1.record()
    rec.append((M, K, N, d, ops.L().conv2d_last_path(), e0, e1))
    return out


def _mkn(x, w, s, p):
    M = x.shape[0] * ((x.shape[1] + 2 * p - w.shape[1]) // s + 1) * ((x.shape[2] + 2 * p - w.shape[2]) // s + 1)
    return M, w.shape[1] * w.shape[2] * w.shape[3], w.shape[0]


def timed(x, w, s, p, **k):
    return _rec(lambda: orig(x, w, s, p, **k), *_mkn(x, w, s, p), "dual" if k.get("x2") is not None else "")


def timed_bn(x, w, s, p, acc, **k):   # the convolution that also adds its BatchNorm column sums
    return _rec(lambda: orig_bn(x, w, s, p, acc, **k), *_mkn(x, w, s, p),
                "dual" if k.get("x2") is not None else "")


def timed_stem(fr, wf, Cout, *a_, **k):   # the 7x7 / stride-2 RGB stem from the frames
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig_stem(fr, wf, Cout, *a_, **k)
    e1.record()
    rec.append((y.numel() // Cout, 147, Cout, "", 9, e0, e1))
    return y


best = {}
for it in range(4):
    rec.clear()
    ops.conv2d_nhwc, ops.conv2d_bn_sums, ops.stem7 = timed, timed_bn, timed_stem
    with torch.no_grad():
        torch.cuda._sleep(int(2.0e8))
        for enc in (pol.net.rgb_encoder, pol.net.depth_encoder):
            enc.trunk_features(obs)
    ops.conv2d_nhwc, ops.conv2d_bn_sums, ops.stem7 = orig, orig_bn, orig_stem
    torch.cuda.synchronize()
    for i, (M, K, N, d, path, e0, e1) in enumerate(rec):
        t = e0.elapsed_time(e1) * 1e3
        best[i] = (M, K, N, d + " " + PATHS.get(path, str(path)), min(t, best[i][4]) if i in best else t)
grp = collections.OrderedDict()
for i in sorted(best):
    M, K, N, d, t = best[i]
    g = grp.setdefault((M, K, N, d), [0, 0.0])
    g[0] += 1
    g[1] += t
tot = sum(v[1] for v in grp.values())
print(f"{len(best)} launches, {tot/1e3:.3f} ms")
for (M, K, N, d), (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1]):
    print(f"M {M:8d} K {K:5d} N {N:5d} {d:13s} x{c:2d}  {t:8.1f} us  {100*t/tot:5.1f}%  {2.0*M*K*N*c/t/1e6:7.1f} TF/s")
