"""Every vlnce_gemm of one cached-feature CMA update (or one policy step with --step): shape,
operand layout, epilogue and GPU time (HIP events around each call), grouped.

    python scripts/gemm_log.py [--step] [--episodes 5] [--steps 100]
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VLNCE_HIP_GRAPHS", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd import data_path, ops  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--step", action="store_true", help="the per-step policy update of bench.py instead")
ap.add_argument("--episodes", type=int, default=5)
ap.add_argument("--steps", type=int, default=100)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
if a.step:
    batch = bench.synth_batch(64, 256, 80, dev, seed=1)

    def run():
        update_agent(policy, opt, *batch, 512)
else:
    rng = np.random.RandomState(0)
    lens = [max(1, int(a.steps * f)) for f in np.linspace(1.0, 0.55, a.episodes)]
    trajs = []
    for T in lens:
        obs = {"rgb_features": rng.rand(T, 2048, 4, 4).astype(np.float16),
               "depth_features": rng.rand(T, 128, 4, 4).astype(np.float16),
               "instruction": np.tile(np.concatenate([rng.randint(1, 2504, size=80),
                                                      np.zeros(120, np.int64)])[None], (T, 1))}
        oracle = rng.randint(0, 4, size=T).astype(np.int64)
        trajs.append((obs, np.concatenate([[0], oracle[:-1]]).astype(np.int64), oracle))
    out = data_path.collate_trajectories(trajs, dev, inflection_coef=3.2)

    def run():
        update_agent(policy, opt, *out, 512)

for _ in range(3):
    run()
torch.cuda.synchronize()
lib = ops.L()
orig = lib.gemm
rec = []


def timed(A, lda, tA, B, ldb, tB, Cm, ldc, M, N, K, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(A, lda, tA, B, ldb, tB, Cm, ldc, M, N, K, **k)
    e1.record()
    rec.append(((M, N, K, tA, tB, "bias" if k.get("shift") is not None else "",
                 int(k.get("act", 0)), int(k.get("accumulate", 0))), e0, e1))


lib.gemm = timed
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
run()
t1.record()
torch.cuda.synchronize()
lib.gemm = orig
grp = collections.OrderedDict()
for key, e0, e1 in rec:
    g = grp.setdefault(key, [0, 0.0])
    g[0] += 1
    g[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in grp.values())
print(f"{len(rec)} gemm calls, {tot:.0f} us of {t0.elapsed_time(t1)*1e3:.0f} us (eager, event pairs include host gaps)")
print(f"{'M':>7s} {'N':>6s} {'K':>6s} tA tB bias act acc   calls   us/call   TF/s")
for (M, N, K, tA, tB, b, act, acc), (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1]):
    print(f"{M:7d} {N:6d} {K:6d} {tA:2d} {tB:2d} {b:4s} {act:3d} {acc:3d}  x{c:3d}  {t/c:9.1f}  {2.0*M*N*K*c/t/1e6:6.1f}")
