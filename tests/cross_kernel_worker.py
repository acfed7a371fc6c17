"""Child process of tests/test_policy_sizes_gpu.py::test_bench_geometry_*: the visual trunks and
the H1 update at the BENCH geometry, with whatever convolution arithmetic the environment selects
(VLNCE_CONV_MATH is read once per process).  Writes a dict of tensors to argv[2].

    python tests/cross_kernel_worker.py cma|waypoint out.pt
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import torch  # noqa: E402

import vlnce_amd  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

DEV = "cuda:0"


def bn_state(mod):
    out = {}
    for name, buf in mod.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            out[name] = buf.detach().float().cpu().clone()
    return out


def cma(out):
    """num_envs=64, 256x256 RGB-D, 80 tokens, BatchNorm on batch statistics (bench.py's workload)"""
    torch.manual_seed(0)
    N, hw, L = 64, 256, 80
    pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(hw, hw)).to(DEV)
    g = torch.Generator().manual_seed(1)
    obs = {"rgb": torch.randint(0, 256, (N, hw, hw, 3), generator=g).float().to(DEV),
           "depth": torch.rand(N, hw, hw, 1, generator=g).to(DEV),
           "instruction": torch.zeros(N, 200, dtype=torch.long)}
    obs["instruction"][:, :L] = torch.randint(1, 2504, (N, L), generator=g)
    obs["instruction"] = obs["instruction"].to(DEV)
    if os.environ.get("VLNCE_TEST_PERTURB"):   # every frame value moved by ONE fp32 rounding
        obs["rgb"] = obs["rgb"] * (1.0 + 2.0 ** -23)
        obs["depth"] = obs["depth"] * (1.0 + 2.0 ** -23)
    prev = torch.randint(0, 4, (N, 1), generator=g).to(DEV)
    masks = (torch.rand(N, 1, generator=g) > 0.1).to(torch.uint8).to(DEV)
    tgt = torch.randint(0, 4, (1, N), generator=g).to(DEV)
    w = (torch.rand(1, N, generator=g) + 0.5).to(DEV)
    res = {}
    with torch.no_grad():
        res["rgb_trunk"] = pol.net.rgb_encoder.trunk_features(obs).float().cpu()
        res["depth_trunk"] = pol.net.depth_encoder.trunk_features(obs).float().cpu()
    for k, v in bn_state(pol.net.rgb_encoder).items():
        res["bn/" + k] = v
    loss, al, xl = update_agent(pol, None, obs, prev, masks, tgt, w, 512, step_grad=False)
    res["loss"] = torch.tensor([float(loss), float(al), float(xl)])
    res["grad_state_q"] = pol.net.state_q.weight.grad.float().cpu()
    res["grad_rgb_kv"] = pol.net.rgb_kv.weight.grad.float().cpu()
    torch.save(res, out)


def waypoint(out):
    """num_envs=32: 12 panorama + 1 history frame per env = 416 RGB-D frames of 256x256, encoders
    in eval mode as the trainer sets them (ddppo_waypoint_trainer.py:528-530)"""
    import cases
    from oracle import thirdparty as tp
    case = dict(policy="WaypointPolicy", hw=256, N=2, T=1, lengths=[200, 173], mode="eval",
                call="waypoint")
    pol, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                tp.synth_state_dict)
    pol = pol.to(DEV).eval()
    N = 32
    big = dict(case, N=N, lengths=[200 - 3 * (i % 11) for i in range(N)])
    obs, prev, masks, _ = cases.build_inputs(big)
    obs = {k: v.to(DEV) for k, v in obs.items()}
    prev = {k: v.to(DEV) for k, v in prev.items()}
    masks = masks.to(DEV)
    h0 = torch.zeros(N, pol.net.num_recurrent_layers, 256, device=DEV)
    with torch.no_grad():
        o = pol.act(obs, h0, prev, masks, deterministic=True)
    torch.save({"value": o[0].float().cpu(), "logits": o[7].logits.float().cpu(),
                "h": o[6].float().cpu()}, out)


if __name__ == "__main__":
    {"cma": cma, "waypoint": waypoint}[sys.argv[1]](sys.argv[2])
