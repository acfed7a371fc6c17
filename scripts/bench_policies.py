"""Secondary timings for the other BASELINE.json configurations (bench.py owns the headline
CMA line): configs[1] Seq2Seq num_envs=32 IL update, configs[4] WaypointPolicy num_envs=32
WDDPPO minibatch update with 200-token instructions.  256x256 RGB-D, synthetic, 1 GPU.

    python scripts/bench_policies.py [--steps 10] [--which seq2seq,waypoint]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import vlnce_amd
from vlnce_amd.il_harness import update_agent
from vlnce_amd.ppo_harness import PPOConfig, wddppo_minibatch_update

dev = torch.device("cuda", 0)


def timeit(fn, steps, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def roofline_of(policy, obs):
    """the `roofline` block of bench.py for this policy's trunks: per-launch HIP-event times of
    every convolution of one eager forward (bench.conv_kernel_time), priced like the headline."""
    c = bench.conv_kernel_time(policy, obs, dev)
    if c["reason"] is not None:
        return {"invalid_reason": c["reason"]}
    ms, flop = c["conv_ms"], c["flop"]
    bf = [v for k, v in c["by_path"].items() if k in (1, 2)]
    bf_ms, bf_flop = sum(v["ms"] for v in bf), sum(v["flop"] for v in bf)
    return {"bound": "mfma", "kernel": "conv2d fwd launches of the visual trunks",
            "launches": c["n"], "kernel_ms_per_step": round(ms, 3),
            "achieved": round(flop / (ms * 1e-3) / 1e12, 2), "peak": bench.FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(flop / (ms * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS, 4),
            "bf16_pipe_frac": (round(6 * bf_flop / (bf_ms * 1e-3) / 1e12 / bench.BF16_MFMA_PEAK_TFLOPS, 4)
                               if bf_ms else None),
            "per_launch_floor_frac": round(c["floor_ms"] / ms, 4),
            "algorithmic_GB": round(c["bytes"] / 1e9, 3), "traffic": None}


def seq2seq(steps, n=32):
    torch.manual_seed(0)
    policy = vlnce_amd.build_model(vlnce_amd.make_config("Seq2SeqPolicy"),
                                   *vlnce_amd.make_spaces(256, 256)).to(dev)
    opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
    obs, prev, masks, tgt, w = bench.synth_batch(n, 256, 80, dev)
    state = {"nxt": policy.encode_ahead(obs)}

    def step():
        cur = state["nxt"]
        state["nxt"] = policy.encode_ahead(obs)
        update_agent(policy, opt, cur, prev, masks, tgt, w, 512)

    def plain():   # the loop the unchanged trainers issue
        update_agent(policy, opt, obs, prev, masks, tgt, w, 512)

    sp = timeit(plain, steps)
    s = timeit(step, steps)
    return {"config": f"Seq2Seq DAgger update, num_envs={n}, 256x256 RGB-D, 80 tokens",
            "ms_per_step": round(1e3 * sp, 3), "policy_steps_per_sec": round(n / sp, 1),
            "encode_ahead_ms_per_step": round(1e3 * s, 3), "roofline": roofline_of(policy, obs)}


def waypoint(steps, n=32, tokens=200):
    torch.manual_seed(0)
    policy = vlnce_amd.build_model(vlnce_amd.make_config("WaypointPolicy"),
                                   *vlnce_amd.make_spaces(256, 256, pano=True)).to(dev)
    policy.train()
    policy.net.rgb_encoder.eval()  # ddppo_waypoint_trainer.py:526-530
    policy.net.depth_encoder.eval()
    opt = torch.optim.Adam([p for p in policy.parameters() if p.requires_grad], lr=2.5e-4)
    g = torch.Generator().manual_seed(1)
    obs = {"rgb": torch.randint(0, 256, (n, 12, 256, 256, 3), generator=g).float(),
           "depth": torch.rand(n, 12, 256, 256, 1, generator=g),
           "rgb_history": torch.randint(0, 256, (n, 256, 256, 3), generator=g).float(),
           "depth_history": torch.rand(n, 256, 256, 1, generator=g),
           "angle_features": torch.randn(n, 12, 4, generator=g),
           "instruction": torch.zeros(n, 200, dtype=torch.long)}
    obs["instruction"][:, :tokens] = torch.randint(1, 2504, (n, tokens), generator=g)
    obs = {k: v.to(dev) for k, v in obs.items()}
    prev = {"pano": torch.randint(0, 12, (n, 1), generator=g).to(dev),
            "offset": ((torch.rand(n, 1, generator=g) - 0.5) * 0.4).to(dev),
            "distance": (0.25 + torch.rand(n, 1, generator=g) * 2.0).to(dev)}
    masks = torch.ones(n, 1, dtype=torch.uint8, device=dev)
    h0 = torch.zeros(n, policy.net.num_recurrent_layers, 256, device=dev)
    with torch.no_grad():
        out = policy.act(obs, h0, {k: v.clone() for k, v in prev.items()}, masks,
                         deterministic=True)
    actions = {k: v.clone() for k, v in out[2].items()}
    actions["pano"] = torch.randint(0, 12, (n, 1), generator=g).to(dev)
    vp = torch.randn(n, 1, generator=g).to(dev) * 0.5
    sample = (obs, h0, actions, prev, vp, vp + 0.3, masks,
              torch.full((n, 1), -2.0, device=dev), torch.randn(n, 1, generator=g).to(dev))

    def step():
        s = list(sample)
        s[3] = {k: v.clone() for k, v in prev.items()}
        wddppo_minibatch_update(policy, opt, tuple(s), PPOConfig())

    s = timeit(step, steps, warm=3)
    frames = n * 13
    return {"config": f"WaypointPolicy WDDPPO minibatch update, num_envs={n}, 12+1 frames/env "
                      f"256x256 RGB-D ({frames} frames), {tokens} tokens",
            "ms_per_step": round(1e3 * s, 3), "policy_steps_per_sec": round(n / s, 1),
            "frames_per_sec": round(frames / s, 1), "roofline": roofline_of(policy, obs)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--which", default="seq2seq,waypoint")
    a = ap.parse_args()
    for name in a.which.split(","):
        print(json.dumps({"seq2seq": seq2seq, "waypoint": waypoint}[name](a.steps)), flush=True)
