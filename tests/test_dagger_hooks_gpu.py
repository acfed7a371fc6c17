"""DAgger's feature capture (dagger_trainer.py:296-314): forward hooks on
`policy.net.rgb_encoder.cnn` and `policy.net.depth_encoder.visual_encoder` that keep `o.cpu()` of
every act() -- the tensors the trainer writes to its LMDB feature cache and later feeds back as
`rgb_features` / `depth_features`.  The HIP trunks return a permuted NHWC view cloned out of a
graph-replayed static buffer, can run ahead on side streams (encode_ahead) and can sit inside a
whole-act() graph: each of those is a place where a hook could stop firing or capture a buffer the
next replay overwrites.  Every captured tensor is compared with what the same hook captures on the
CPU oracle."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

import vlnce_amd  # noqa: E402
from oracle import policy_cpu as oc  # noqa: E402
from oracle import thirdparty as tp  # noqa: E402

DEV = "cuda:0"


def hook_builder(tgt_tensor):
    # verbatim shape of the reference's closure (dagger_trainer.py:296-300)
    def hook(m, i, o):
        tgt_tensor.set_(o.cpu())

    return hook


def _policies(hw):
    torch.manual_seed(0)
    policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(hw, hw))
    ref = oc.build_model(tp.make_config("CMAPolicy"), *tp.make_spaces(hw, hw))
    sd = tp.synth_state_dict(ref)
    ref.load_state_dict(sd)
    policy.load_state_dict(sd)
    return policy.to(DEV), ref


def _batch(n, hw, seed):
    g = torch.Generator().manual_seed(seed)
    obs = {
        "rgb": torch.randint(0, 256, (n, hw, hw, 3), generator=g).float(),
        "depth": torch.rand(n, hw, hw, 1, generator=g),
        "instruction": torch.zeros(n, 200, dtype=torch.long),
    }
    obs["instruction"][:, :11] = torch.randint(1, 2504, (n, 11), generator=g)
    return obs


@pytest.mark.parametrize("mode", ["plain", "encode_ahead", "act_graph"])
def test_dagger_feature_hooks_capture_what_the_reference_captures(mode, monkeypatch):
    hw, n = 256, 2
    if mode == "act_graph":
        monkeypatch.setenv("VLNCE_ACT_GRAPH", "1")
    policy, ref = _policies(hw)
    rgb_f, dep_f = torch.zeros((1,)), torch.zeros((1,))
    rgb_r, dep_r = torch.zeros((1,)), torch.zeros((1,))
    hooks = [policy.net.rgb_encoder.cnn.register_forward_hook(hook_builder(rgb_f)),
             policy.net.depth_encoder.visual_encoder.register_forward_hook(hook_builder(dep_f)),
             ref.net.rgb_encoder.cnn.register_forward_hook(hook_builder(rgb_r)),
             ref.net.depth_encoder.visual_encoder.register_forward_hook(hook_builder(dep_r))]
    states = torch.zeros(n, policy.net.num_recurrent_layers, 512)
    prev = torch.zeros(n, 1, dtype=torch.long)
    masks = torch.ones(n, 1, dtype=torch.uint8)
    kept = []
    # _update_dataset runs under no_grad with the policy as constructed (train-mode BatchNorm in
    # the frozen RGB trunk, App. B-1): 1st call eager, 2nd captures the trunk graphs, 3rd / 4th replay
    with torch.no_grad():
        for step in range(4):
            obs = _batch(n, hw, 100 + step)
            dobs = {k: v.to(DEV) for k, v in obs.items()}
            if mode == "encode_ahead":
                dobs = policy.encode_ahead(dobs)
            policy.act(dobs, states.to(DEV), prev.to(DEV), masks.to(DEV), deterministic=True)
            ref.act(obs, states, prev, masks, deterministic=True)
            assert tuple(rgb_f.shape) == (n, 2048, 4, 4) and tuple(dep_f.shape) == (n, 128, 4, 4)
            assert rgb_f.device.type == "cpu" and dep_f.device.type == "cpu"
            assert rgb_f.shape == rgb_r.shape and dep_f.shape == dep_r.shape
            # The RGB trunk runs 53 BatchNorms on the statistics of a batch of TWO here (the last
            # ones over 2 x 8 x 8 values): measured 1.7e-4 (three bf16 planes) / 1.9e-4 (fp16 planes)
            # against the CPU oracle, the depth trunk (GroupNorm) 1.2e-5.  What this test is about
            # -- the hook fires on every call, on the right buffer, with the reference's shape -- shows
            # as O(1) errors; the policy-level 1e-4 parity is tests/test_policy_gpu.py's.
            er = ((rgb_f - rgb_r).abs() / (1.0 + rgb_r.abs())).max().item()
            ed = ((dep_f - dep_r).abs() / (1.0 + dep_r.abs())).max().item()
            print(f"{mode} step {step}: rgb err {er:.2e} (max |x| {rgb_r.abs().max():.2f}), "
                  f"depth err {ed:.2e} (max |x| {dep_r.abs().max():.2f})")
            assert er < 1e-3 and ed < 1e-4, (mode, step, er, ed)
            kept.append((rgb_f.clone(), dep_f.clone(), rgb_f, dep_f))
            # what the trainer stores must survive the next replay: set_() rebinds the target to
            # a fresh host copy every step, so step k's values must still be step k's afterwards
            rgb_f, dep_f = torch.zeros((1,)), torch.zeros((1,))
            for h in hooks[:2]:
                h.remove()
            hooks[0] = policy.net.rgb_encoder.cnn.register_forward_hook(hook_builder(rgb_f))
            hooks[1] = policy.net.depth_encoder.visual_encoder.register_forward_hook(hook_builder(dep_f))
    for snap_r, snap_d, live_r, live_d in kept:
        assert torch.equal(snap_r, live_r) and torch.equal(snap_d, live_d)
    # the captured features fed back through the cached-feature bypass give the same action
    # distribution as the frames (dagger_trainer.py:559-583 -> resnet_encoders.py:70-72,193-195)
    obs = _batch(n, hw, 103)
    dobs = {k: v.to(DEV) for k, v in obs.items()}
    with torch.no_grad():
        d_frames = policy.build_distribution(dobs, states.to(DEV), prev.to(DEV), masks.to(DEV))
        cached = dict(dobs, rgb_features=kept[-1][0].to(DEV), depth_features=kept[-1][1].to(DEV))
        d_cached = policy.build_distribution(cached, states.to(DEV), prev.to(DEV), masks.to(DEV))
    assert (d_frames.logits - d_cached.logits).abs().max().item() < 1e-4
    for h in hooks:
        h.remove()
