"""Observation transforms: oracle and product against the goldens produced by
the reference's own CenterCropperPerSensor / ObsStack (tests/golden/make_goldens_obs.py), the
uint8 ingest kernels against their contract, and end-to-end equality of the policy on uint8
frames / centre-crop views with the fp32 path."""
import os

import numpy as np
import pytest
import torch

import hostsim
import make_goldens_obs as mg
import vlnce_amd
from oracle import policy_cpu as oc
from vlnce_amd import obs_transforms as ot
from vlnce_amd import ops

GOLD = os.path.dirname(os.path.abspath(mg.__file__))
DEV = "cuda:0"


def gold(name):
    return {k[4:]: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


def same(a, b):
    assert set(a) == set(b), (sorted(a), sorted(b))
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert torch.equal(a[k].cpu(), b[k].cpu()), k


def single_cam():
    return {k: v for k, v in mg.inputs().items() if k in ("rgb", "depth", "instruction")}


# ------------------------------------------------------------------ CPU tier
def test_oracle_matches_reference_goldens():
    same(oc.center_cropper_per_sensor(single_cam(), mg.CROPS), gold("obs_center_crop.npz"))
    same(oc.obs_stack(mg.inputs(), mg.STACK), gold("obs_stack.npz"))
    same(oc.center_cropper_per_sensor(oc.obs_stack(mg.inputs(), mg.STACK), mg.CROPS),
         gold("obs_stack_crop.npz"))


def test_product_transforms_match_reference_goldens_on_cpu():
    same(ot.CenterCropperPerSensor(mg.CROPS)(single_cam()), gold("obs_center_crop.npz"))
    same(ot.ObsStack(mg.STACK)(mg.inputs()), gold("obs_stack.npz"))
    both = ot.apply_obs_transforms_batch(mg.inputs(), [ot.ObsStack(mg.STACK),
                                                       ot.CenterCropperPerSensor(mg.CROPS)])
    same(both, gold("obs_stack_crop.npz"))


def test_observation_space_rewrites():
    spaces, _ = vlnce_amd.make_spaces(20, 26)
    sp = {f"rgb{'' if i == 0 else '_' + str(i)}": spaces.spaces["rgb"] for i in range(12)}
    sp.update({f"depth{'' if i == 0 else '_' + str(i)}": spaces.spaces["depth"] for i in range(12)})
    space = vlnce_amd.config.Dict(sp)
    out = ot.ObsStack(mg.STACK).transform_observation_space(space)
    assert set(out.spaces) == {"rgb", "depth"}
    assert out.spaces["rgb"].shape == (12, 20, 26, 3) and out.spaces["depth"].shape == (12, 20, 26, 1)
    out = ot.CenterCropperPerSensor(mg.CROPS).transform_observation_space(out)
    assert out.spaces["rgb"].shape == (12, 14, 16, 3) and out.spaces["depth"].shape == (12, 16, 18, 1)
    assert set(space.spaces) == set(sp)  # the input space is not modified


def test_batch_obs_dtypes():
    """uint8 image sensors keep their storage dtype; everything else is fp32 exactly like
    habitat's batch_obs (`torch.tensor(..., dtype=torch.float)`): the reference's sensors return
    float64 progress / angle_features and int64 tokens / shortest-path actions."""
    envs = [{"rgb": np.full((4, 5, 3), i, np.uint8), "depth": np.full((4, 5, 1), i / 4, np.float32),
             "instruction": np.arange(6) + i, "progress": np.array([i / 3.0]),
             "angle_features": np.linspace(0, 1, 48).reshape(12, 4) * i,
             "shortest_path_sensor": np.array([i], np.int64)} for i in range(3)]
    b = ot.batch_obs(envs, "cpu")
    assert b["rgb"].dtype == torch.uint8 and b["rgb"].shape == (3, 4, 5, 3)
    assert torch.equal(b["rgb"][2], torch.full((4, 5, 3), 2, dtype=torch.uint8))
    for k in ("depth", "instruction", "progress", "angle_features", "shortest_path_sensor"):
        assert b[k].dtype == torch.float32, (k, b[k].dtype)
    assert b["progress"].shape == (3, 1) and b["angle_features"].shape == (3, 12, 4)
    assert torch.equal(b["instruction"][1], torch.arange(6).float() + 1)
    assert torch.equal(b["progress"], torch.tensor([[0.0], [1 / 3.0], [2 / 3.0]]))


def resize_inputs(tag):
    hs, ws, size, crops = mg.RESIZE_CASES[tag]
    o = {k: v for k, v in mg.inputs(seed=11, n=3, hs=hs, ws=ws).items()
         if k in ("rgb", "depth", "instruction")}
    return o, size, crops


@pytest.mark.parametrize("tag", list(mg.RESIZE_CASES))
def test_resize_shortest_edge_matches_goldens_on_cpu(tag):
    """habitat's ResizeShortestEdge (oracle/thirdparty.py restatement -> goldens) then the
    reference's own CenterCropperPerSensor: oracle, product in sequence, product fused."""
    from oracle import thirdparty as tp
    o, size, crops = resize_inputs(tag)
    full = np.load(os.path.join(GOLD, f"obs_resize_{tag}.npz"))
    want = {k[4:]: torch.from_numpy(full[k]) for k in full if k.startswith("out_")}
    resized = {k[8:]: torch.from_numpy(full[k]) for k in full if k.startswith("resized_")}
    same(tp.resize_shortest_edge(dict(o), size), resized)
    same(oc.center_cropper_per_sensor(tp.resize_shortest_edge(dict(o), size), crops), want)
    same(ot.ResizeShortestEdge(size)(dict(o)), resized)
    seq = ot.CenterCropperPerSensor(crops)(ot.ResizeShortestEdge(size)(dict(o)))
    same(seq, want)
    fused = ot.apply_obs_transforms_batch(dict(o), [ot.ResizeShortestEdge(size),
                                                    ot.CenterCropperPerSensor(crops)])
    same({k: v.contiguous() for k, v in fused.items()}, want)


def test_resize_shortest_edge_observation_space_and_config():
    spaces, _ = vlnce_amd.make_spaces(480, 640)
    out = ot.ResizeShortestEdge(256).transform_observation_space(spaces)
    assert out.spaces["rgb"].shape == (256, 341, 3) and out.spaces["depth"].shape == (256, 341, 1)
    assert spaces.spaces["rgb"].shape == (480, 640, 3)
    out = ot.CenterCropperPerSensor([("rgb", (224, 224)), ("depth", (256, 256))]) \
        .transform_observation_space(out)
    assert out.spaces["rgb"].shape == (224, 224, 3) and out.spaces["depth"].shape == (256, 256, 1)
    cfg = vlnce_amd.make_config("CMAPolicy")
    cfg.RL = vlnce_amd.config.Config(POLICY=vlnce_amd.config.Config(OBS_TRANSFORMS=vlnce_amd.config.Config(
        RESIZE_SHORTEST_EDGE=vlnce_amd.config.Config(SIZE=256))))
    assert ot.ResizeShortestEdge.from_config(cfg)._size == 256


def test_frame_descriptor_reads_crop_views_without_a_copy():
    base = torch.randint(0, 256, (3, 4, 10, 12, 3), dtype=torch.uint8)
    view = base[:, :, 2:8, 1:11, :]
    fr = ops.frames(view)
    assert fr["x"].data_ptr() == base.data_ptr() and (fr["Hs"], fr["Ws"]) == (10, 12)
    assert (fr["y0"], fr["x0"], fr["H"], fr["W"], fr["F"], fr["images"]) == (2, 1, 6, 10, 4, 12)
    want = view.float().reshape(12, 6, 10, 3)
    assert torch.equal(hostsim.HostSim._frames_f32(fr), want)
    # frame stack + masked extra frame (the waypoint net's 12 + 1 frames)
    hist = torch.randint(0, 256, (3, 6, 10, 3), dtype=torch.uint8)
    mask = torch.tensor([1, 0, 1])
    fr = ops.frames((view, hist, mask))
    got = hostsim.HostSim._frames_f32(fr).reshape(3, 5, 6, 10, 3)
    assert fr["images"] == 15 and torch.equal(got[:, :4], view.float())
    assert torch.equal(got[:, 4], hist.float() * mask.view(3, 1, 1, 1).float())


# ------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
def test_obs_stack_and_crop_kernels_match_reference_goldens():
    dev_in = {k: v.to(DEV) for k, v in mg.inputs().items()}
    same(ot.ObsStack(mg.STACK)(dict(dev_in)), gold("obs_stack.npz"))
    both = ot.apply_obs_transforms_batch(dict(dev_in), [ot.ObsStack(mg.STACK),
                                                        ot.CenterCropperPerSensor(mg.CROPS)])
    same({k: v.contiguous() for k, v in both.items()}, gold("obs_stack_crop.npz"))
    # stack + crop fused in the gather kernel
    srcs = [dev_in["rgb" + ("" if i == 0 else f"_{i}")] for i in range(12)]
    y0, x0, h, w = ot.center_crop_window(20, 26, (14, 16))
    assert torch.equal(ops.frames_gather(srcs, (y0, x0, h, w)).cpu(), gold("obs_stack_crop.npz")["rgb"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(mg.RESIZE_CASES))
def test_resize_kernel_matches_goldens(tag):
    """vlnce_frames_resize_area: the whole resized frame and the fused resize + centre crop,
    uint8 and fp32, bit for bit against the goldens."""
    o, size, crops = resize_inputs(tag)
    full = np.load(os.path.join(GOLD, f"obs_resize_{tag}.npz"))
    want = {k[4:]: torch.from_numpy(full[k]) for k in full if k.startswith("out_")}
    resized = {k[8:]: torch.from_numpy(full[k]) for k in full if k.startswith("resized_")}
    dev_in = {k: v.to(DEV) for k, v in o.items()}
    same(ot.ResizeShortestEdge(size)(dict(dev_in)), resized)
    fused = ot.apply_obs_transforms_batch(dict(dev_in), [ot.ResizeShortestEdge(size),
                                                         ot.CenterCropperPerSensor(crops)])
    assert fused["rgb"].is_contiguous() and fused["rgb"].dtype == torch.uint8
    same(fused, want)


@pytest.mark.gpu
def test_resize_crop_at_the_rxr_geometry_feeds_the_policy():
    """RxR sensors: 480 x 640 uint8 RGB + fp32 depth -> ResizeShortestEdge(256) -> centre crop
    224 x 224 / 256 x 256 (rxr_cma_en.yaml:27-30, config/default.py:132-175), 12-camera stacks
    included; against the oracle on the CPU, and straight into the CMA policy."""
    from oracle import thirdparty as tp
    g = torch.Generator().manual_seed(9)
    rgb = torch.randint(0, 256, (2, 480, 640, 3), generator=g, dtype=torch.uint8)
    depth = torch.rand(2, 480, 640, 1, generator=g)
    crops = [("rgb", (224, 224)), ("depth", (256, 256))]
    want = oc.center_cropper_per_sensor(tp.resize_shortest_edge({"rgb": rgb.clone(), "depth": depth.clone()}, 256), crops)
    tf = [ot.ResizeShortestEdge(256), ot.CenterCropperPerSensor(crops)]
    got = ot.apply_obs_transforms_batch({"rgb": rgb.to(DEV), "depth": depth.to(DEV)}, tf)
    assert got["rgb"].shape == (2, 224, 224, 3) and got["depth"].shape == (2, 256, 256, 1)
    assert torch.equal(got["rgb"].cpu(), want["rgb"])
    assert torch.equal(got["depth"].cpu(), want["depth"])
    stack = torch.randint(0, 256, (2, 3, 60, 80, 3), generator=g, dtype=torch.uint8)   # [N, cams, H, W, C]
    w5 = tp.image_resize_shortest_edge(stack.clone(), 32, channels_last=True)
    assert torch.equal(ot.ResizeShortestEdge(32)({"rgb": stack.to(DEV)})["rgb"].cpu(), w5)
    # the real sensor geometry: 224 x 224 RGB + 256 x 256 depth (only depth's shape is read
    # from the observation space, resnet_encoders.py:32-38)
    torch.manual_seed(0)
    pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256))
    pol.to(DEV).eval()
    obs = dict(got)
    ins = torch.zeros(2, 200, dtype=torch.long)
    ins[:, :7] = torch.randint(1, 2504, (2, 7), generator=g)
    obs["instruction"] = ins.to(DEV)
    h0 = torch.zeros(2, pol.net.num_recurrent_layers, 512, device=DEV)
    with torch.no_grad():
        a, _ = pol.act(obs, h0, torch.zeros(2, 1, dtype=torch.long, device=DEV),
                       torch.ones(2, 1, dtype=torch.uint8, device=DEV), deterministic=True)
    assert a.shape == (2, 1) and a.dtype == torch.int64


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_ingest_kernels_match_their_contract(dtype):
    sim = hostsim.HostSim()
    g = torch.Generator().manual_seed(3)
    mk = (lambda *s: torch.randint(0, 256, s, generator=g, dtype=torch.uint8)) if dtype == torch.uint8 \
        else (lambda *s: torch.rand(*s, generator=g) * 255)
    base = mk(3, 4, 12, 14, 3)
    view = base[:, :, 1:11, 2:12, :]                     # centre-crop view of a frame stack
    hist, mask = mk(3, 10, 10, 3), torch.tensor([1, 0, 1], dtype=torch.uint8)
    sc, sh = torch.rand(3) + 0.5, torch.randn(3)
    for cpu_in in (base[:, 0], view, (view, hist, mask)):
        dev_in = tuple(t.to(DEV) for t in cpu_in) if isinstance(cpu_in, tuple) else cpu_in.to(DEV)
        if not isinstance(cpu_in, tuple) and cpu_in is view:  # keep it a VIEW on the device too
            dev_in = base.to(DEV)[:, :, 1:11, 2:12, :]
        fc, fd = ops.frames(cpu_in), ops.frames(dev_in)
        assert fd["x"].is_cuda and (fd["y0"], fd["x0"]) == (fc["y0"], fc["x0"])
        n, H, W = fc["images"], fc["H"], fc["W"]
        want = torch.empty(n, H // 2 + 3, W // 2 + 3, 12)
        sim.frames_s2d(fc, want, 2, 1, sc, sh)
        # (x*scale+shift contracts to one fma on the GPU: equal to a rounding of the product)
        got = ops.frames_s2d(fd, 2, 1, sc.to(DEV), sh.to(DEV)).cpu()
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-4) and torch.equal(got == 0, want == 0)
        want = torch.empty(n, H, W, 3)
        sim.frames_f32(fc, want, sc, sh)
        assert torch.allclose(ops.frames_f32(fd, sc.to(DEV), sh.to(DEV)).cpu(), want, rtol=1e-6, atol=1e-4)
        raw = torch.empty(n, H, W, 3)
        sim.frames_f32(fc, raw)
        assert torch.equal(ops.frames_f32(fd).cpu(), raw)  # no transform: exact
        want = torch.empty(n, H // 2, W // 2, 3)
        sim.frames_avgpool2(fc, want)
        assert (ops.frames_avgpool2(fd).cpu() - want).abs().max() <= 1e-4 * 255


@pytest.mark.gpu
def test_policy_on_uint8_frames_and_crop_views_equals_fp32_frames():
    """CMA act() on uint8 RGB (a quarter of the H2D bytes) and on a centre-crop VIEW of larger
    frames gives the logits of the reference-style fp32 contiguous input (the ingest itself is
    bit-exact, test_ingest_kernels_match_their_contract; downstream split-K atomics round)."""
    torch.manual_seed(0)
    pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(64, 64))
    pol.to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    N = 3
    big_rgb = torch.randint(0, 256, (N, 80, 72, 3), generator=g, dtype=torch.uint8).to(DEV)
    big_d = torch.rand(N, 70, 76, 1, generator=g).to(DEV)
    cropped = ot.CenterCropperPerSensor([("rgb", (64, 64)), ("depth", (64, 64))])(
        {"rgb": big_rgb, "depth": big_d})
    assert not cropped["rgb"].is_contiguous() and cropped["rgb"].dtype == torch.uint8
    ins = torch.zeros(N, 200, dtype=torch.long)
    ins[:, :11] = torch.randint(1, 2504, (N, 11), generator=g)
    common = dict(instruction=ins.to(DEV))
    h0 = torch.zeros(N, pol.net.num_recurrent_layers, 512, device=DEV)
    prev = torch.zeros(N, 1, dtype=torch.long, device=DEV)
    masks = torch.ones(N, 1, dtype=torch.uint8, device=DEV)

    def logits(obs):
        with torch.no_grad():
            out = [pol.build_distribution(obs, h0, prev, masks).logits.clone() for _ in range(3)]
        # eager = capturing call = graph replay (to rounding: small-batch convolutions and tail
        # GEMMs combine split-K partial sums with fp32 atomics)
        assert (out[0] - out[1]).abs().max() < 1e-5 and (out[1] - out[2]).abs().max() < 1e-5
        return out[0]

    ref = logits(dict(common, rgb=cropped["rgb"].float().contiguous(),
                      depth=cropped["depth"].contiguous()))
    u8 = logits(dict(common, rgb=cropped["rgb"].contiguous(), depth=cropped["depth"].contiguous()))
    assert (u8 - ref).abs().max() < 1e-5                                          # uint8
    assert (logits(dict(common, **cropped)) - ref).abs().max() < 1e-5            # uint8 views
