"""N1: the cached-feature DAgger step on one MI355X.

  1. collate: B ragged trajectories (fp16 feature cache shapes) -> padded fp32 batch on the
     device; reports GB/s of the pad/widen kernels against the HBM roof, the end-to-end time
     including the H2D of the compact rows, and the reference's way (pad on the host in the
     storage dtype, then .to(device, float32)) timed on the same box.
  2. the CMA update on that batch (CNN trunks bypassed: rgb_features / depth_features).

    python scripts/bench_data_path.py [--episodes 5] [--steps 100] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vlnce_amd
from oracle import data_cpu as od  # cpu baseline leg only
from vlnce_amd import data_path
from vlnce_amd.il_harness import update_agent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=5)   # IL.batch_size
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--update-only", action="store_true",
                    help="only the CMA update on the collated batch (for rocprofv3 runs)")
    a = ap.parse_args()
    dev = "cuda:0"
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
    B, Tmax, rows = a.episodes, max(lens), sum(lens)
    D = 2048 * 16 + 128 * 16
    alg = rows * D * 2 + Tmax * B * D * 4  # fp16 rows read once + fp32 padded batch written once

    for _ in range(3):
        out = data_path.collate_trajectories(trajs, dev, inflection_coef=3.2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = data_path.collate_trajectories(trajs, dev, inflection_coef=3.2)
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / a.iters

    if a.update_only:
        torch.manual_seed(0)
        policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"),
                                       *vlnce_amd.make_spaces(256, 256)).to(dev)
        opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
        obs_b, prev_b, masks_b, corr_b, w_b = out
        for _ in range(4):
            update_agent(policy, opt, obs_b, prev_b, masks_b, corr_b, w_b, 512)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            update_agent(policy, opt, obs_b, prev_b, masks_b, corr_b, w_b, 512)
        torch.cuda.synchronize()
        upd = (time.perf_counter() - t0) / a.iters
        print(json.dumps({"cma_update_ms": round(upd * 1e3, 3), "rows": Tmax * B,
                          "fused_steps": os.environ.get("VLNCE_RNN_STEP_FUSED", "1"),
                          "instr_dedup": os.environ.get("VLNCE_INSTR_DEDUP", "1")}))
        return

    # kernels alone: rows already on the device
    lib = vlnce_amd.ops.L()
    off = torch.zeros(B + 1, dtype=torch.int32)
    off[1:] = torch.tensor(lens, dtype=torch.int32).cumsum(0)
    off = off.to(dev)
    src = {k: torch.cat([torch.as_tensor(t[0][k]) for t in trajs]).to(dev)
           for k in ("rgb_features", "depth_features")}
    dst = {k: torch.empty((Tmax * B,) + tuple(v.shape[1:]), device=dev) for k, v in src.items()}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        ev0.record()
        for _ in range(a.iters):
            for k in src:
                lib.ragged_pad_rows(src[k], off, B, Tmax, src[k][0].numel(), 1.0, dst[k])
        ev1.record()
        torch.cuda.synchronize()
    kern = ev0.elapsed_time(ev1) / a.iters * 1e-3

    # the reference's way on this box: pad in the storage dtype on the host, then cast on device
    samples = [({k: torch.from_numpy(v) for k, v in o.items()}, torch.from_numpy(p),
                torch.from_numpy(q), od.inflection_weights(torch.from_numpy(q), 3.2))
               for o, p, q in trajs]

    def ref_way():
        obs_b, prev_b, m_b, c_b, w_b = od_collate_storage(samples)
        obs_d = {k: v.to(device=dev, dtype=torch.float32, non_blocking=True) for k, v in obs_b.items()}
        return obs_d, prev_b.to(dev), m_b.to(dev), c_b.to(dev), w_b.to(dev)

    def od_collate_storage(batch):  # od.collate without the fp32 cast (that happens on the GPU)
        T = max(s[1].size(0) for s in batch)
        obs = {}
        for sensor in batch[0][0]:
            st = torch.stack([od.pad_to(s[0][sensor], T, 1.0) for s in batch], dim=1)
            obs[sensor] = st.view(-1, *st.shape[2:])
        rest = od.collate([({}, s[1], s[2], s[3]) for s in batch])
        return (obs,) + tuple(rest[1:])

    for _ in range(2):
        ref_way()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(3, a.iters // 4)):
        ref_way()
    torch.cuda.synchronize()
    ref = (time.perf_counter() - t0) / max(3, a.iters // 4)

    # the update step on the collated batch
    torch.manual_seed(0)
    policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"),
                                   *vlnce_amd.make_spaces(256, 256)).to(dev)
    opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
    obs_b, prev_b, masks_b, corr_b, w_b = out

    def step():
        update_agent(policy, opt, obs_b, prev_b, masks_b, corr_b, w_b, 512)

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    upd = (time.perf_counter() - t0) / a.iters
    print(json.dumps({
        "workload": f"cached-feature DAgger batch: {B} episodes, lengths {lens}, fp16 cache, "
                    f"{Tmax * B} padded rows x {D} features + 200 tokens",
        "collate_kernels": {"ms": round(kern * 1e3, 3), "algorithmic_GB": round(alg / 1e9, 4),
                            "GB_per_s": round(alg / kern / 1e9, 1), "hbm_peak_GB_per_s": 8000,
                            "frac": round(alg / kern / 8e12, 3)},
        "collate_end_to_end_ms": round(e2e * 1e3, 3),
        "reference_way_ms (host pad + .to(device, fp32))": round(ref * 1e3, 3),
        "cma_update_ms": round(upd * 1e3, 3),
        "cma_update_rows_per_s": round(Tmax * B / upd, 1)}))


if __name__ == "__main__":
    main()
