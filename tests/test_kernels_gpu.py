"""GPU tier, per-kernel parity: every C-ABI entry point on the MI355X against
its contract (tests/hostsim.py: plain torch fp32 CPU ops) on the same seeded
random inputs.  Tolerance: 1e-4 relative to the output scale (fp32; the MFMA
is an exact-fp32 fmaf chain, differences are summation order only).
Shapes are the ones the policies issue scaled to a small
batch, plus ragged edges (rows / channels / K not multiples of the tile)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import hostsim
from vlnce_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SIM = hostsim.HostSim()


@pytest.fixture(scope="module")
def hip():
    return _lib.get_lib()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(hash((shape, seed)) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, tol=1e-4, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: max|d|={err:.3e} scale={scale:.3e}"


def both(method, tensors, scalars):
    """Run lib.<method>(**tensors, **scalars) on the simulator (CPU) and on the HIP
    library (GPU copies); return ({name: cpu tensor}, {name: gpu tensor})."""
    cpu = {k: (v.clone() if v is not None else None) for k, v in tensors.items()}
    gpu = {k: (v.to(DEV) if v is not None else None) for k, v in tensors.items()}
    getattr(SIM, method)(**cpu, **scalars)
    if method == "conv2d_fwd":  # the pre-split weights select the bf16-plane kernel where it applies
        gpu["w_split"] = ops.split_weights(gpu["w"])
        gpu["w_frag"] = ops.pack_weights(gpu["w"])
    getattr(_lib.get_lib(), method)(**gpu, **scalars)
    torch.cuda.synchronize()
    return cpu, gpu


# ------------------------------------------------------------------ conv
CONV_CASES = [
    # name,            N,  H,  W, Cin, Cout, k, s, p, extras
    ("1x1_64_256",     2, 16, 16,  64, 256, 1, 1, 0, dict(scale=True, relu=True)),
    ("1x1_256_64_res", 2, 16, 16, 256,  64, 1, 1, 0, dict(scale=True, relu=True, residual=True)),
    ("3x3_64_64",      2, 16, 16,  64,  64, 3, 1, 1, dict(scale=True, relu=True)),
    ("3x3s2_128",      2, 16, 16, 128, 128, 3, 2, 1, dict()),
    ("1x1s2_256_512",  2, 16, 16, 256, 512, 1, 2, 0, dict(scale=True)),
    ("3x3_512_8x8",    1,  8,  8, 512, 512, 3, 1, 1, dict(stats=True)),
    ("stem_rgb",       2, 64, 64,   3,  64, 7, 2, 3, dict(prologue=True, scale=True, relu=True)),
    ("stem_depth",     2, 32, 32,   1,  32, 7, 2, 3, dict()),
    ("big_128x128",    8, 32, 32,  64, 256, 1, 1, 0, dict(stats=True)),      # 128x128 tiles
    ("big_128x64",    16, 32, 32,  64,  64, 3, 1, 1, dict(stats=True)),      # 128x64 tiles
    ("ragged",         3,  7,  9,  36,  48, 3, 1, 1, dict(scale=True, residual=True, relu=True)),
    ("prologue_v4",    2, 12, 12,  32,  64, 3, 1, 1, dict(prologue=True, in_relu=True)),
    ("prologue_v4_c",  2, 12, 12,  32,  64, 3, 1, 1, dict(prologue=True, in_relu=True, center=True)),
    ("prologue_s_c",   2, 12, 12,   6,  64, 3, 1, 1, dict(prologue=True, in_relu=True, center=True)),
    ("prologue_buf_c", 2, 14, 14,  64, 128, 3, 2, 1, dict(prologue=True, in_relu=True, center=True)),
    # dual-input prologue (block end evaluated in the next block's first 1x1 conv)
    ("dual_identity",  2, 16, 16,  64,  64, 1, 1, 0, dict(prologue=True, in_relu=True, center=True,
                                                          dual="identity")),
    ("dual_bn",        3,  9, 11, 256, 128, 1, 1, 0, dict(prologue=True, in_relu=True, center=True,
                                                          dual="bn")),
    ("dual_big",      16, 64, 64, 256, 128, 1, 1, 0, dict(prologue=True, in_relu=True, center=True,
                                                          dual="bn")),               # 128x128 tiles
    ("dual_wide_n",    4, 16, 16, 128, 512, 1, 1, 0, dict(prologue=True, in_relu=True, center=True,
                                                          dual="identity")),         # 4 n-tiles
    ("compress_1024",  2,  4,  4, 1024, 128, 3, 1, 1, dict()),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_fwd(hip, case):
    name, N, H, W, Cin, Cout, k, s, p, ex = case
    x = rnd(N, H, W, Cin, seed=1)
    w = rnd(Cout, k, k, Cin, seed=2, scale=(Cin * k * k) ** -0.5)
    g = ops.conv_geometry(x, w, s, p)
    M = N * g["Ho"] * g["Wo"]
    t = dict(x=x, w=w, y=torch.zeros(N, g["Ho"], g["Wo"], Cout))
    t["in_scale"] = rnd(Cin, seed=3).abs() + 0.5 if ex.get("prologue") else None
    t["in_shift"] = rnd(Cin, seed=4) * 0.3 if ex.get("prologue") else None
    t["in_center"] = rnd(Cin, seed=14) * 0.5 if ex.get("center") else None
    if ex.get("dual"):
        t["x2"] = rnd(N, H, W, Cin, seed=15)
        t["side_out"] = torch.zeros(N, H, W, Cin)
        if ex["dual"] == "bn":
            t["in2_scale"] = rnd(Cin, seed=16).abs() + 0.5
            t["in2_shift"] = rnd(Cin, seed=17) * 0.3
            t["in2_center"] = rnd(Cin, seed=18) * 0.5
    t["scale"] = rnd(Cout, seed=5).abs() + 0.5 if ex.get("scale") else None
    t["shift"] = rnd(Cout, seed=6) if ex.get("scale") else None
    t["residual"] = rnd(N, g["Ho"], g["Wo"], Cout, seed=7) if ex.get("residual") else None
    sc = dict(g=g, in_relu=int(bool(ex.get("in_relu"))), ldr=Cout,
              act=1 if ex.get("relu") else 0, accumulate=0)
    # statistics use each side's own tiling; compare after finalize
    cpu, gpu = both("conv2d_fwd", t, sc)
    close(gpu["y"], cpu["y"], what=name)
    if ex.get("dual"):
        close(gpu["side_out"], cpu["side_out"], 1e-6, what=name + "/side_out")
    if ex.get("stats"):
        x_d, w_d = x.to(DEV), w.to(DEV)
        y, stats = ops.conv2d_nhwc(x_d, w_d, s, p, want_stats=True)
        gamma, beta = (rnd(Cout, seed=8).abs() + 0.5).to(DEV), rnd(Cout, seed=9).to(DEV)
        rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        scale, shift, center = ops.bn_finalize(stats, M, gamma, beta, 1e-5, 0.1, rm, rv)
        out = ops.scale_shift_act(y, scale, shift, center=center, act=1)
        raw = cpu["y"].reshape(M, Cout)
        rm_ref, rv_ref = torch.zeros(Cout), torch.ones(Cout)
        ref = F.batch_norm(raw.t().reshape(1, Cout, M), rm_ref, rv_ref, gamma.cpu(), beta.cpu(),
                           True, 0.1, 1e-5)
        ref = torch.relu(ref).reshape(Cout, M).t()
        close(out.reshape(M, Cout), ref, what=name + "/bn_train")
        close(rm, rm_ref, what=name + "/running_mean")
        close(rv, rv_ref, what=name + "/running_var")


def test_conv_identity_weight_is_not_transposed(hip):
    """A = I style check with an asymmetric operand: a 1x1 conv whose weight is a
    permutation matrix must permute channels exactly: bit-exact in plane format 1 (three bf16
    planes hold all 24 mantissa bits) and on the fp32-MFMA kernel, to 2^-22 relative in format 2
    (two fp16 planes: 11 + 11 bits and a sign)."""
    Cc = 64
    perm = torch.randperm(Cc, generator=torch.Generator().manual_seed(3))
    w = torch.zeros(Cc, 1, 1, Cc)
    w[torch.arange(Cc), 0, 0, perm] = 1.0
    x = rnd(2, 9, 5, Cc, seed=11).to(DEV)
    want = x[..., perm.to(DEV)]
    assert torch.equal(ops.conv2d_nhwc(x, w.to(DEV), 1, 0, w_format=1), want)
    with hip.options(conv_math=0):
        assert torch.equal(ops.conv2d_nhwc(x, w.to(DEV), 1, 0), want)
    y = ops.conv2d_nhwc(x, w.to(DEV), 1, 0, w_format=2)
    assert bool(((y - want).abs() <= want.abs() * 2.0 ** -22).all())


# ------------------------------------------------------------------ gemm
GEMM_CASES = [
    # M, N, K, transA, transB, act, bias
    (64, 256, 2112, 0, 0, 1, True),    # rgb_linear
    (64, 1536, 416, 0, 0, 0, True),    # GRU input projection
    (64, 4, 512, 0, 0, 0, True),       # action head
    (64, 1, 512, 0, 0, 3, True),       # progress monitor (tanh)
    (640, 512, 50, 0, 0, 0, True),     # instruction x W_ih (K % 4 != 0: scalar loaders)
    (100, 260, 388, 0, 0, 2, True),    # waypoint sizes (sigmoid)
    (64, 512, 256, 0, 1, 0, False),    # dX = dY W
    (70, 50, 36, 0, 1, 0, False),      # dX ragged, scalar A
    (256, 2112, 64, 1, 1, 0, False),   # dW = dY^T X
    (4, 512, 64, 1, 1, 0, False),      # dW of the action head
    (1, 512, 37, 1, 1, 0, False),      # dW of a 1-output head, ragged K
    (5000, 256, 256, 0, 0, 0, True),   # text_k over B*L rows (128-row tiles)
    (40000, 64, 64, 0, 0, 0, False),   # M > 32767: row folding
    (512, 50, 5120, 1, 1, 0, False),   # dW_ih of the instruction RNN: split-K over L*B rows
    (64, 128, 3072, 0, 0, 1, True),    # depth_linear: split-K + bias/ReLU second pass
    (64, 416, 1536, 0, 1, 0, False),   # dX of the GRU input projection: split-K
]


@pytest.mark.parametrize("case", GEMM_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}_{c[3]}{c[4]}" for c in GEMM_CASES])
def test_gemm(hip, case):
    M, N, K, ta, tb, act, bias = case
    A = rnd(K, M, seed=1) if ta else rnd(M, K, seed=1)
    B = rnd(K, N, seed=2) if tb else rnd(N, K, seed=2)
    A *= K ** -0.5
    t = dict(A=A, B=B, Cm=torch.zeros(M, N), shift=rnd(N, seed=3) if bias else None)
    sc = dict(lda=A.size(1), transA=ta, ldb=B.size(1), transB=tb, ldc=N, M=M, N=N, K=K, act=act)
    cpu, gpu = both("gemm", t, sc)
    close(gpu["Cm"], cpu["Cm"], what=str(case))


def test_gemm_accumulate_and_strided_views(hip):
    M, N, K = 48, 96, 128
    big = rnd(M, 2 * K, seed=1)
    A = big[:, K:]  # row stride 2K, 16-byte aligned offset
    B = rnd(N, K, seed=2)
    C0 = rnd(M, N, seed=3)
    t = dict(A=A, B=B, Cm=C0.clone())
    cpu = {k: v.clone() for k, v in t.items()}
    SIM.gemm(A, 2 * K, 0, B, K, 0, cpu["Cm"], N, M, N, K, accumulate=1)
    bg = big.to(DEV)
    Cg = C0.to(DEV)
    _lib.get_lib().gemm(bg[:, K:], 2 * K, 0, B.to(DEV), K, 0, Cg, N, M, N, K, accumulate=1)
    close(Cg, cpu["Cm"], what="accumulate")
    # skinny shapes take the split-K path: C += A B (+ bias) with atomics into the existing C
    for (M2, N2, K2, tb) in ((5, 512, 1536, 1), (5, 1536, 512, 0), (3, 64, 4096, 0)):
        A2, C2, bias = rnd(M2, K2, seed=4), rnd(M2, N2, seed=6), rnd(N2, seed=7)
        B2 = rnd(K2, N2, seed=5) if tb else rnd(N2, K2, seed=5)
        ref = C2.clone()
        SIM.gemm(A2, K2, 0, B2, N2 if tb else K2, tb, ref, N2, M2, N2, K2, shift=bias, accumulate=1)
        got = C2.to(DEV)
        _lib.get_lib().gemm(A2.to(DEV), K2, 0, B2.to(DEV), N2 if tb else K2, tb, got, N2, M2, N2,
                            K2, shift=bias.to(DEV), accumulate=1)
        close(got, ref, what=f"split-K accumulate {M2}x{N2}x{K2}")


# ------------------------------------------------------------------ norms / pools
@pytest.mark.parametrize("tiles,Cc", [(7, 96), (256, 64), (700, 64), (8192, 64), (300, 200),
                                      (16389, 64), (4099, 32)])
def test_bn_finalize_contract(hip, tiles, Cc):
    rows = 64
    M = tiles * rows - 40
    part = rnd(tiles, Cc, 2, seed=1).abs()
    t = dict(partial=part, gamma=rnd(Cc, seed=2), beta=rnd(Cc, seed=3),
             running_mean=rnd(Cc, seed=4), running_var=rnd(Cc, seed=5).abs(),
             scale_out=torch.zeros(Cc), shift_out=torch.zeros(Cc), mean_out=torch.zeros(Cc),
             rstd_out=torch.zeros(Cc))
    # more than 4096 tiles: the finalize first coarsens the partials into caller-provided scratch
    wb = _lib.get_lib().bn_finalize_workspace_bytes(tiles, Cc)
    assert (wb > 0) == (tiles > 4096)
    t["workspace"] = torch.zeros(wb // 8, dtype=torch.float64) if wb else None
    sc = dict(tiles_m=tiles, tile_rows=rows, M=M, Cc=Cc, eps=1e-5, momentum=0.1)
    cpu, gpu = both("bn_finalize", t, sc)
    for k in ("scale_out", "shift_out", "mean_out", "rstd_out", "running_mean", "running_var"):
        close(gpu[k], cpu[k], what=k)


@pytest.fixture(params=["one_launch", "pipeline"])
def gn_path(request, monkeypatch):
    """GroupNorm of a small activation is one launch (vlnce_group_norm_small), of a large one
    statistics + finalize + apply: run the small test shapes through both."""
    if request.param == "pipeline":
        monkeypatch.setattr(ops, "GN_SMALL_ELEMENTS", 0)
    return request.param


@pytest.mark.parametrize("shape", [(4, 32, 32, 32, 16), (2, 8, 8, 256, 16), (3, 4, 4, 1024, 16),
                                   (2, 4, 4, 128, 1), (2, 2, 2, 2048, 1), (5, 16, 16, 64, 16),
                                   (1, 64, 64, 128, 16), (1, 3, 5, 48, 16)])
def test_group_norm(hip, shape, gn_path):
    N, H, W, Cc, G = shape
    x = rnd(N, H, W, Cc, seed=1) + 0.5
    gamma, beta = rnd(Cc, seed=2), rnd(Cc, seed=3)
    res = rnd(N, H, W, Cc, seed=4)
    ref = F.group_norm(x.permute(0, 3, 1, 2), G, gamma, beta, 1e-5) + res.permute(0, 3, 1, 2)
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    out = ops.group_norm_act(x.to(DEV), G, gamma.to(DEV), beta.to(DEV), 1e-5,
                             residual=res.to(DEV), act=1)
    close(out, ref, what=str(shape))


def test_scale_shift_act_scalar_path(hip):
    x = rnd(33, 7, seed=1)
    s, b = rnd(7, seed=2), rnd(7, seed=3)
    out = ops.scale_shift_act(x.to(DEV), s.to(DEV), b.to(DEV), act=3)
    close(out, torch.tanh(x * s + b))


@pytest.mark.parametrize("shape", [(2, 32, 32, 64), (3, 17, 13, 32), (2, 9, 9, 3)])
def test_maxpool(hip, shape):
    x = rnd(*shape, seed=1)
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(ops.maxpool3x3s2(x.to(DEV)).cpu(), ref)


def test_avgpool_and_adaptive(hip):
    x = rnd(3, 64, 64, 1, seed=1)
    close(ops.avgpool2x2(x.to(DEV)), F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1), 1e-6)
    for (h, w, oh, ow) in [(8, 8, 4, 4), (2, 2, 4, 4), (1, 1, 4, 4), (8, 8, 1, 1), (7, 5, 4, 4)]:
        y = rnd(2, h, w, 96, seed=h * 10 + w)
        ref = F.adaptive_avg_pool2d(y.permute(0, 3, 1, 2), (oh, ow)).permute(0, 2, 3, 1)
        close(ops.adaptive_avgpool(y.to(DEV), oh, ow), ref, 1e-6, what=f"{h}x{w}->{oh}x{ow}")
    z = rnd(5, 16, 2112, seed=2)
    close(ops.mean_rows(z.to(DEV)), z.mean(1), 1e-6)


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("cfg", [(3, 80, 256, 256, 1), (2, 200, 256, 256, 1), (4, 16, 256, 128, 0),
                                 (5, 12, 128, 128, 0), (2, 37, 256, 256, 2), (2, 1, 64, 32, 0)])
def test_attention_fwd_bwd(hip, cfg):
    B, P, Dk, Dv, mode = cfg
    q, K, V = rnd(B, Dk, seed=1), rnd(B, P, Dk, seed=2), rnd(B, P, Dv, seed=3)
    mask = None
    if mode:
        mask = torch.zeros(B, P, dtype=torch.uint8)
        for b in range(B):
            mask[b, max(1, P - 3 - b):] = 1
    dout = rnd(B, Dv, seed=4)
    scale = Dk ** -0.5
    t = dict(q=q, K=K, V=V, mask=mask, out=torch.zeros(B, Dv), attn_out=torch.zeros(B, P))
    sc = dict(ldk=Dk, ldv=Dv, mask_mode=mode, scale=scale, B=B, P=P, Dk=Dk, Dv=Dv)
    cpu, gpu = both("attn_fwd", t, sc)
    close(gpu["out"], cpu["out"], what="attn out")
    close(gpu["attn_out"], cpu["attn_out"], what="attn probs")
    t = dict(dout=dout, q=q, K=K, V=V, mask=mask, attn=cpu["attn_out"], dq=torch.zeros(B, Dk),
             dK=torch.zeros(B, P, Dk), dV=torch.zeros(B, P, Dv))
    sc = dict(ldk=Dk, ldv=Dv, mask_mode=mode, scale=scale, lddk=Dk, lddv=Dv, B=B, P=P, Dk=Dk, Dv=Dv)
    cpu, gpu = both("attn_bwd", t, sc)
    for k in ("dq", "dK", "dV"):
        close(gpu[k], cpu[k], what=k)


def test_attention_autograd_matches_torch(hip):
    B, P, D = 3, 16, 256
    kv = rnd(B, P, 2 * D, seed=1).to(DEV).requires_grad_()
    q = rnd(B, D, seed=2).to(DEV).requires_grad_()
    out = ops.attention(q, kv[..., :D], kv[..., D:], None, 1, D ** -0.5)
    out.square().sum().backward()
    kv2, q2 = kv.detach().cpu().requires_grad_(), q.detach().cpu().requires_grad_()
    a = torch.softmax(torch.einsum("bd,bpd->bp", q2, kv2[..., :D]) * D ** -0.5, 1)
    torch.einsum("bp,bpd->bd", a, kv2[..., D:]).square().sum().backward()
    close(kv.grad, kv2.grad, what="d kv (strided K/V views)")
    close(q.grad, q2.grad, what="d q")


@pytest.mark.parametrize("M", [64, 1000])
def test_linear_input_gradient_for_the_trailing_columns_only(hip, M):
    """ops.linear(dx_from=c0): the input gradient is computed for columns [c0, K) only and the
    leading ones (a frozen trunk's features) come back zero; weight / bias gradients are the full
    ones.  Against fp64 autograd."""
    K, N, c0 = 2112, 320, 2048
    x, w, b, g = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(M, N, seed=4)
    xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
    (ops.linear(xd, wd, bd, dx_from=c0) * g.to(DEV)).sum().backward()
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    ((xr @ wr.t() + br) * g.double()).sum().backward()
    assert float(xd.grad[:, :c0].abs().max()) == 0.0
    close(xd.grad[:, c0:], xr.grad[:, c0:], what="dx trailing columns")
    close(wd.grad, wr.grad, what="dW")
    close(bd.grad, br.grad, what="db")


def test_large_linear_runs_on_the_bf16_plane_kernels(hip):
    """ops.linear with >= 1024 rows and >= 1 GFLOP (sequence-mode batches: rgb_kv at 500 x 16
    rows, the Waypoint tail at 416 frames) goes through vlnce_conv2d_fwd as a 1x1 convolution --
    forward with bias + ReLU, and the input gradient as dz (W^T)^T; values and all three gradients
    against fp64 at 1e-4.  Rows with a stride (a column slice of a wider matrix) included."""
    M, K, N = 4000, 544, 288     # 4000 = 4 x 1000 pixel rows; 2 M N K = 1.25 GFLOP
    wide = rnd(M, K + 32, seed=1)
    w, b, g = rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    for act in (ops.ACT_NONE, ops.ACT_RELU):
        xd = wide.to(DEV).requires_grad_()
        wd, bd = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        hip.set_option("m3", hip.get_option("m3"))   # (touch the dispatch record)
        y = ops.linear(xd[:, :K], wd, bd, act)
        assert hip.conv2d_last_path() in (1, 2, 3)       # a bf16-plane kernel took it
        (y * g.to(DEV)).sum().backward()
        xr = wide.double().requires_grad_()
        wr, br = w.double().requires_grad_(), b.double().requires_grad_()
        ref = xr[:, :K] @ wr.t() + br
        ref = torch.relu(ref) if act == ops.ACT_RELU else ref
        (ref * g.double()).sum().backward()
        close(y, ref, what="y")
        close(xd.grad, xr.grad, what="dx")
        close(wd.grad, wr.grad, what="dW")
        close(bd.grad, br.grad, what="db")


# ------------------------------------------------------------------ categorical action head
@pytest.mark.parametrize("cfg", [(64, 512, 4), (1, 512, 4), (7, 512, 6), (64, 514, 6), (300, 96, 16),
                                 (2500, 512, 4), (5, 33, 1)])
def test_action_head_fwd_bwd_match_torch(hip, cfg):
    """vlnce_action_head_fwd / _bwd against Categorical(logits=Linear(x)).logits and its autograd
    (fp64 reference, tolerance 1e-4 of the largest value); M > 1024 takes the atomic dW path."""
    M, K, A = cfg
    x, w, b = rnd(M, K, seed=1), rnd(A, K, seed=2, scale=0.2), rnd(A, seed=3)
    g = rnd(M, A, seed=4)
    xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
    n, cnt = ops.action_head(xd, wd, bd)
    (n * g.to(DEV)).sum().backward()
    assert int(cnt.item()) == 0
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    ref = torch.distributions.Categorical(logits=xr @ wr.t() + br).logits
    (ref * g.double()).sum().backward()
    close(n, ref, what="normalised logits")
    for got, want, what in ((xd.grad, xr.grad, "dx"), (wd.grad, wr.grad, "dW"), (bd.grad, br.grad, "db")):
        close(got, want, what=what)


def test_action_head_counts_nan_rows_and_strided_input(hip):
    M, K, A = 9, 512, 4
    big = rnd(M, K + 64, seed=5).to(DEV)
    x = big[:, :K]                       # row stride K + 64
    w, b = rnd(A, K, seed=6).to(DEV), rnd(A, seed=7).to(DEV)
    n, cnt = ops.action_head(x, w, b)
    close(n, torch.log_softmax(x @ w.t() + b, -1), what="strided rows")
    assert int(cnt.item()) == 0
    bad = x.clone()
    bad[2, 5] = float("nan")
    bad[7, 0] = float("inf")             # inf - inf in the normalisation: NaN, as in the reference
    n, cnt = ops.action_head(bad, w, b)
    assert int(cnt.item()) == 2 and bool(torch.isnan(n[2]).all()) and bool(torch.isnan(n[7]).any())
    cnt.zero_()
    n, cnt = ops.action_head(bad, w, None, count_nans=False)
    assert cnt is None


@pytest.mark.parametrize("cfg", [(12, 3, 40, 128, 256, 1), (500, 5, 200, 128, 256, 1), (7, 7, 9, 64, 32, 0),
                                 (9, 2, 33, 128, 128, 2)])
def test_attention_shared_blocks_and_segment_sum(hip, cfg):
    """ops.attention(index=...): K / V / mask blocks shared by groups of queries
    (vlnce_attn_fwd_shared / _bwd_shared + vlnce_segment_sum) against the same attention over the
    expanded K[index] / V[index] through torch autograd (fp64)."""
    B, U, P, Dk, Dv, mode = cfg
    q, K, V = rnd(B, Dk, seed=1), rnd(U, P, Dk, seed=2), rnd(U, P, Dv, seed=3)
    idx = (torch.arange(B) * 7 + 3) % U
    mask = None
    if mode:
        mask = torch.zeros(U, P, dtype=torch.uint8)
        for u in range(U):
            mask[u, max(1, P - 3 - u):] = 1
    g = rnd(B, Dv, seed=4)
    scale = Dk ** -0.5
    qd, Kd, Vd = (t.to(DEV).requires_grad_() for t in (q, K, V))
    out = ops.attention(qd, Kd, Vd, None if mask is None else mask.to(DEV), mode or 1, scale,
                        index=idx.to(DEV))
    (out * g.to(DEV)).sum().backward()
    qr, Kr, Vr = (t.double().requires_grad_() for t in (q, K, V))
    logits = torch.einsum("bd,bpd->bp", qr, Kr[idx])
    if mode == 1:
        logits = logits - mask[idx].double() * 1e8
    if mode == 2:
        logits = logits * mask[idx].double()
    ref = torch.einsum("bp,bpd->bd", torch.softmax(logits * scale, 1), Vr[idx])
    (ref * g.double()).sum().backward()
    close(out, ref, what="out")
    for got, want, what in ((qd.grad, qr.grad, "dq"), (Kd.grad, Kr.grad, "dK"), (Vd.grad, Vr.grad, "dV")):
        close(got, want, what=what)
    x = rnd(B, 520, seed=5)
    got = torch.empty(U, 520, device=DEV)
    ops.L().segment_sum(x.to(DEV), idx.to(DEV), B, U, 520, got)
    want = torch.zeros(U, 520, dtype=torch.float64).index_add_(0, idx, x.double())
    close(got, want, what="segment_sum")


def test_rowzero_mask(hip):
    x = rnd(4, 30, 256, seed=1)
    x[1, 20:] = 0
    x[3, 5:] = 0
    x[2, 7, :255] = 0  # one non-zero element left: NOT masked
    assert torch.equal(ops.rowzero_mask(x.to(DEV)).cpu(), (x == 0).all(2).to(torch.uint8))


# ------------------------------------------------------------------ recurrent cells
def test_gru_lstm_cells_match_torch(hip):
    B, D, H = 7, 96, 128
    for kind in ("GRU", "LSTM"):
        cell = (torch.nn.GRUCell if kind == "GRU" else torch.nn.LSTMCell)(D, H)
        x, h0, c0 = rnd(B, D, seed=1), rnd(B, H, seed=2), rnd(B, H, seed=3)
        pd = {k: v.detach().to(DEV).requires_grad_() for k, v in cell.named_parameters()}
        xd, hd, cd = (t.to(DEV).requires_grad_() for t in (x, h0, c0))
        gi = ops.linear(xd, pd["weight_ih"], pd["bias_ih"])
        xr, hr, cr = (t.clone().requires_grad_() for t in (x, h0, c0))
        if kind == "GRU":
            out = ops.gru_cell(gi, hd, pd["weight_hh"], pd["bias_hh"])
            ref = cell(xr, hr)
            (out * out).sum().backward()
            (ref * ref).sum().backward()
        else:
            out, cn = ops.lstm_cell(gi, hd, cd, pd["weight_hh"], pd["bias_hh"])
            ref, cref = cell(xr, (hr, cr))
            ((out * out).sum() + (cn * 0.5).sum()).backward()
            ((ref * ref).sum() + (cref * 0.5).sum()).backward()
            close(cn, cref, what="c'")
            close(cd.grad, cr.grad, what="dc")
        close(out, ref, what=kind + " h'")
        close(xd.grad, xr.grad, what=kind + " dx")
        close(hd.grad, hr.grad, what=kind + " dh")
        for k, v in cell.named_parameters():
            close(pd[k].grad, v.grad, what=f"{kind} d{k}")


def test_row_utilities(hip):
    x, y = rnd(9, 40, seed=1), rnd(9, 40, seed=2)
    m = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0, 1], dtype=torch.uint8)
    close(ops.mask_rows(x.to(DEV), m.to(DEV)), x * m[:, None].float(), 0)
    close(ops.select_rows(m.to(DEV), x.to(DEV), y.to(DEV)), torch.where(m[:, None] != 0, x, y), 0)
    close(ops.select_rows(m.to(DEV), x.to(DEV), None), torch.where(m[:, None] != 0, x, 0 * x), 0)
    t = dict(x=rnd(301, 77, seed=3), out=torch.zeros(77))
    cpu, gpu = both("colsum", t, dict(ldx=77, M=301, N=77, accumulate=0))
    close(gpu["out"], cpu["out"], what="colsum")
    for M_, acc in ((64, 0), (200, 1), (5000, 1)):  # direct (M <= 256) and sliced/atomic paths
        t = dict(x=rnd(M_, 130, seed=6), out=rnd(130, seed=7))
        cpu, gpu = both("colsum", t, dict(ldx=130, M=M_, N=130, accumulate=acc))
        close(gpu["out"], cpu["out"], what=f"colsum M={M_} acc={acc}")
    for act in (1, 2, 3):
        yv = torch.sigmoid(rnd(50, 12, seed=4)) if act == 2 else torch.tanh(rnd(50, 12, seed=4))
        t = dict(dy=rnd(50, 12, seed=5), y=yv, dz=torch.zeros(50, 12))
        cpu, gpu = both("act_bwd", t, dict(n=600, act=act))
        close(gpu["dz"], cpu["dz"], what=f"act_bwd {act}")


def test_linear_autograd_matches_torch(hip):
    for (M, K, N, act) in [(64, 1184, 512, 1), (33, 50, 20, 0), (16, 512, 1, 3)]:
        lin = torch.nn.Linear(K, N)
        x = rnd(M, K, seed=1)
        xd = x.to(DEV).requires_grad_()
        w, b = (p.detach().to(DEV).requires_grad_() for p in lin.parameters())
        y = ops.linear(xd, w, b, act)
        (y * y).sum().backward()
        xr = x.clone().requires_grad_()
        yr = hostsim._act(lin(xr), act)
        (yr * yr).sum().backward()
        close(y, yr, what="linear fwd")
        close(xd.grad, xr.grad, what="linear dx")
        close(w.grad, lin.weight.grad, what="linear dW")
        close(b.grad, lin.bias.grad, what="linear db")


# ------------------------------------------------------------------ persistent packed-sequence RNN
@pytest.mark.parametrize("cfg", [(0, 2, 128, 19, 23), (1, 2, 128, 5, 9), (0, 1, 64, 33, 7),
                                 (1, 1, 64, 16, 12), (0, 2, 128, 64, 80)])
def test_rnn_seq_kernels_match_contract(hip, cfg):
    kind, dirs, H, B, Lm = cfg
    G = 4 if kind == 0 else 3
    gen = torch.Generator().manual_seed(7)
    lengths = torch.randint(1, Lm + 1, (B,), generator=gen).to(torch.int32)
    lengths[0] = Lm
    gi = [rnd(Lm, B, G * H, seed=10 + d) for d in range(dirs)]
    w = [rnd(G * H, H, seed=20 + d, scale=H ** -0.5) for d in range(dirs)]
    bh = [rnd(G * H, seed=30 + d, scale=0.1) for d in range(dirs)]

    def run(lib, dev):
        mv = lambda ts: [t.to(dev) for t in ts]  # noqa: E731
        out = [torch.zeros(Lm, B, H, device=dev) for _ in range(dirs)]
        hf = [torch.zeros(B, H, device=dev) for _ in range(dirs)]
        gs = [torch.zeros(Lm, B, G * H, device=dev) for _ in range(dirs)]
        ax = [torch.zeros(Lm, B, H, device=dev) for _ in range(dirs)]
        lib.rnn_seq_fwd(kind, dirs, mv(gi), mv(w), mv(bh), lengths.to(dev), out, hf, gs, ax, B, Lm, H)
        dout = mv([rnd(Lm, B, H, seed=40 + d) for d in range(dirs)])
        dhf = mv([rnd(B, H, seed=50 + d) for d in range(dirs)])
        dgi = [torch.zeros(Lm, B, G * H, device=dev) for _ in range(dirs)]
        dgh = [torch.zeros(Lm, B, G * H, device=dev) for _ in range(dirs)] if kind == 1 else None
        wt = [x.t().contiguous().to(dev) for x in w]
        lib.rnn_seq_bwd(kind, dirs, wt, lengths.to(dev), out, gs, ax, dout, dhf, dgi, dgh, B, Lm, H)
        return out, hf, dgi, dgh

    ref = run(SIM, "cpu")
    got = run(_lib.get_lib(), DEV)
    torch.cuda.synchronize()
    names = ["out", "h_final", "dgi", "dgh"]
    for name, r, g in zip(names, ref, got):
        if r is None:
            continue
        for d in range(dirs):
            close(g[d], r[d], 2e-4, what=f"{name}[{d}] {cfg}")


@pytest.mark.parametrize("rnn_type,bidir,final_only", [("LSTM", True, False), ("LSTM", False, True),
                                                       ("GRU", True, False), ("GRU", False, True)])
def test_instruction_encoder_matches_torch_packed_rnn(hip, rnn_type, bidir, final_only):
    """Whole InstructionEncoder (embedding -> packed (bi)RNN) on HIP vs the CPU oracle's
    nn.LSTM/GRU + pack_padded_sequence, forward AND all parameter gradients."""
    from oracle import policy_cpu as oc
    from oracle import thirdparty as tp
    from vlnce_amd.encoders.instruction_encoder import InstructionEncoder

    cfg = tp.default_model_config().INSTRUCTION_ENCODER
    cfg.rnn_type, cfg.bidirectional, cfg.final_state_only = rnn_type, bidir, final_only
    ref = oc.InstructionEncoder(cfg)
    hipm = InstructionEncoder(cfg)
    hipm.load_state_dict(ref.state_dict())
    hipm.to(DEV)
    B = 21
    gen = torch.Generator().manual_seed(3)
    tok = torch.zeros(B, 200, dtype=torch.long)
    for i in range(B):
        n = int(torch.randint(1, 60, (1,), generator=gen))
        tok[i, :n] = torch.randint(1, 2504, (n,), generator=gen)
    yr = ref({"instruction": tok})
    yh = hipm({"instruction": tok.to(DEV)})
    close(yh, yr, 2e-4, what="instruction encoder output")
    wgt = rnd(*yr.shape, seed=9)
    (yr * wgt).sum().backward()
    (yh * wgt.to(DEV)).sum().backward()
    for (n, pr), (_, ph) in zip(ref.named_parameters(), hipm.named_parameters()):
        close(ph.grad, pr.grad, 5e-4, what=f"d {n}")


def test_instruction_encoder_backward_takes_an_expanded_gradient(hip):
    """ An upstream gradient that is an EXPANDED tensor -- seq.mean over the time axis, a
    broadcast add: unit inner stride, zero time stride -- must reach vlnce_rnn_seq_bwd2 as real rows
    (the entry point rejects non-positive strides)."""
    from oracle import policy_cpu as oc
    from oracle import thirdparty as tp
    from vlnce_amd.encoders.instruction_encoder import InstructionEncoder

    cfg = tp.default_model_config().INSTRUCTION_ENCODER
    cfg.rnn_type, cfg.bidirectional, cfg.final_state_only = "LSTM", True, False
    ref = oc.InstructionEncoder(cfg)
    hipm = InstructionEncoder(cfg)
    hipm.load_state_dict(ref.state_dict())
    hipm.to(DEV)
    tok = torch.zeros(5, 200, dtype=torch.long)
    gen = torch.Generator().manual_seed(4)
    for i in range(5):
        tok[i, :7 + 3 * i] = torch.randint(1, 2504, (7 + 3 * i,), generator=gen)
    yr = ref({"instruction": tok})        # [B, C, L]
    yh = hipm({"instruction": tok.to(DEV)})
    yr.mean(dim=2).sum().backward()       # d/dseq = 1 / L expanded along the time axis
    yh.mean(dim=2).sum().backward()
    for (n, pr), (_, ph) in zip(ref.named_parameters(), hipm.named_parameters()):
        close(ph.grad, pr.grad, 5e-4, what=f"d {n}")


@pytest.mark.parametrize("B,clipped,with_offset", [(6, True, True), (416, True, True), (1000, False, False),
                                                   (3, True, False)])
def test_ppo_loss_and_gradients_match_torch(hip, B, clipped, with_offset):
    """vlnce_ppo_loss (the WDDPPO minibatch loss, ddppo_alg.py:78-121, and all its gradients in one
    launch) against the reference's formulas evaluated by torch with autograd: rows inside and
    outside both clipping ranges, exact ties (value_preds == values, ratio == 1), zero advantages."""
    from vlnce_amd.ppo_harness import PPOConfig, PPOLossFn

    cfg = PPOConfig(pano_entropy_coef=1.5, offset_entropy_coef=1.0, distance_entropy_coef=0.25,
                    use_clipped_value_loss=clipped)
    values, vp = rnd(B, 1, seed=1), rnd(B, 1, seed=2) * 0.6
    vp[::5] = values[::5]                        # ties of the two value losses
    returns = values + rnd(B, 1, seed=3) * 0.5
    old = -2.0 + 0.3 * rnd(B, 1, seed=4)
    logp = old + 0.25 * rnd(B, 1, seed=5)       # ratios on both sides of [0.8, 1.2]
    logp[1::7] = old[1::7]                       # ratio == 1 exactly
    adv = rnd(B, 1, seed=6)
    adv[2::9] = 0.0
    ents = [rnd(B, seed=7 + k).abs() for k in range(3)]
    radians = rnd(B, 1, seed=11) * 0.2 if with_offset else None

    leaves = [t.clone().requires_grad_(True) for t in (values, logp, *ents)]
    v, lp, ep, eo, ed = leaves
    entropy_loss = (cfg.pano_entropy_coef * ep + cfg.offset_entropy_coef * eo
                    + cfg.distance_entropy_coef * ed).mean() * cfg.entropy_coef
    ratio = torch.exp(lp - old)
    action_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1 - cfg.clip_param, 1 + cfg.clip_param) * adv).mean()
    if clipped:
        vpc = vp + (v - vp).clamp(-cfg.clip_param, cfg.clip_param)
        value_loss = 0.5 * torch.max((v - returns).pow(2), (vpc - returns).pow(2)).mean()
    else:
        value_loss = 0.5 * (returns - v).pow(2).mean()
    value_loss = value_loss * cfg.value_loss_coef
    offset_loss = cfg.offset_regularize_coef * radians.abs().mean() if with_offset else 0.0
    loss_r = value_loss + action_loss + offset_loss - entropy_loss
    (3.0 * loss_r).backward()

    hl = [t.clone().to(DEV).requires_grad_(True) for t in (values, logp, *ents)]
    loss, stats = PPOLossFn.apply(*hl, vp.to(DEV), returns.to(DEV), old.to(DEV), adv.to(DEV),
                                  radians.to(DEV) if with_offset else None, cfg)
    (3.0 * loss).backward()
    want = torch.stack([loss_r.detach(), value_loss.detach(), action_loss.detach(), entropy_loss.detach(),
                        ents[0].mean(), ents[1].mean(), ents[2].mean(), torch.as_tensor(float(offset_loss))])
    close(stats, want, 2e-6, what="stats")
    close(loss, loss_r, 2e-6, what="loss")
    for name, h, r in zip(("values", "logp", "ent_pano", "ent_offset", "ent_distance"), hl, leaves):
        close(h.grad, r.grad, 1e-5, what=f"d {name}")


LINEAR_ROWS_CASES = [
    # M,   N,    K,   act, bias, strided x
    (64, 256, 2112, 1, True, False),    # rgb_linear of a 64-environment CMA step
    (64, 128, 3072, 1, True, False),    # depth_linear
    (64, 1536, 416, 0, True, False),    # first state encoder, input gates
    (64, 512, 1184, 1, True, True),     # second_state_compress, x a column window of a wider matrix
    (32, 768, 512, 3, True, False),
    (128, 64, 256, 2, False, False),
    (100, 36, 52, 1, True, False),      # ragged everywhere: rows, a 4-column strip tail, K % 16 != 0
    (5, 16, 4, 0, True, False),
    (1, 4, 2052, 3, True, False),
    (17, 260, 1028, 0, False, True),
]


@pytest.mark.parametrize("M,N,K,act,bias,strided", LINEAR_ROWS_CASES)
def test_linear_rows_fwd_bwd_match_torch(hip, M, N, K, act, bias, strided):
    """vlnce_linear_rows_fwd / _bwd (one launch per direction of a <= 128-row linear layer) against
    torch: y = act(x W^T + b), and dx, dW, db through the activation's derivative."""
    x_full = rnd(M, K + (8 if strided else 0), seed=1)
    x = x_full[:, 4:4 + K] if strided else x_full
    w, b = rnd(N, K, seed=2) * (K ** -0.5), (rnd(N, seed=3) if bias else None)
    gy = rnd(M, N, seed=4)
    fn = {0: lambda t: t, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh}[act]
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = fn(xr @ wr.t() + (br if bias else 0))
    yr.backward(gy)
    xh = x_full.to(DEV)
    xh = (xh[:, 4:4 + K] if strided else xh).requires_grad_(True)
    wh = w.to(DEV).requires_grad_(True)
    bh = b.to(DEV).requires_grad_(True) if bias else None
    lib = ops.L()
    assert lib.linear_rows_supported(M, N, K)
    calls = []
    orig = lib.linear_rows_fwd, lib.linear_rows_bwd
    lib.linear_rows_fwd = lambda *a: (calls.append("fwd"), orig[0](*a))[1]
    lib.linear_rows_bwd = lambda *a: (calls.append("bwd"), orig[1](*a))[1]
    try:
        yh = ops.linear(xh, wh, bh, act)
        yh.backward(gy.to(DEV))
    finally:
        del lib.linear_rows_fwd, lib.linear_rows_bwd
    # this layer did run on the skinny-linear kernels (the forward of a reduction past 2048 stays
    # on the general GEMM, ops.LinearFn.forward)
    assert calls == (["fwd", "bwd"] if K <= 2048 else ["bwd"]), calls
    close(yh, yr, 2e-5, what="y")
    close(xh.grad, xr.grad, 2e-5, what="dx")
    close(wh.grad, wr.grad, 2e-5, what="dW")
    if bias:
        close(bh.grad, br.grad, 2e-5, what="db")
    # parameters only (a frozen input) and input only (frozen parameters)
    y2 = ops.linear(xh.detach(), wh, bh, act)
    wh.grad = None
    y2.backward(gy.to(DEV))
    close(wh.grad, wr.grad, 2e-5, what="dW, no dx")
    x3 = xh.detach().requires_grad_(True)
    ops.linear(x3, wh.detach(), None if not bias else bh.detach(), act).backward(gy.to(DEV))
    close(x3.grad, xr.grad, 2e-5, what="dx, no dW")


@pytest.mark.parametrize("rnn_type,bidir", [("LSTM", True), ("GRU", True), ("GRU", False)])
def test_rnn_layer_fn_input_gradient_and_sliced_output_gradient(hip, rnn_type, bidir):
    """ops.RNNLayerFn (vlnce_rnn_seq_fwd2 / _bwd2 / _wgrad) against torch's packed nn.LSTM / nn.GRU:
    outputs in the consumer's [B, L, dirs*H] rows with zeros past each length (from an uninitialised
    buffer), final states, the gradient of the time-major INPUT rows and of every parameter, with
    the output gradient arriving as a slice of a longer padded buffer (what F.pad's backward hands
    over when the tail pads the instruction to a bucket of 8 tokens)."""
    B, Lm, E, H = 19, 13, 50, 128
    dirs = 2 if bidir else 1
    gen = torch.Generator().manual_seed(11)
    lengths = torch.randint(1, Lm + 1, (B,), generator=gen)
    lengths[3] = Lm
    rnn = (torch.nn.LSTM if rnn_type == "LSTM" else torch.nn.GRU)(E, H, bidirectional=bidir)
    x = rnd(Lm, B, E, seed=12)
    xr = x.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lengths, enforce_sorted=False)
    out_p, hn = rnn(packed)
    hn = hn[0] if rnn_type == "LSTM" else hn
    out_r, _ = torch.nn.utils.rnn.pad_packed_sequence(out_p, total_length=Lm)   # [L, B, dirs*H]
    wseq, wfin = rnd(B, Lm + 3, dirs * H, seed=13), rnd(dirs, B, H, seed=14)
    ((out_r.transpose(0, 1) * wseq[:, :Lm]).sum() + (hn * wfin).sum()).backward()

    xh = x.to(DEV).reshape(Lm * B, E).requires_grad_(True)
    prm = {n: p.detach().to(DEV).requires_grad_(True) for n, p in rnn.named_parameters()}
    quads = [tuple(prm[n + sfx] for n in ("weight_ih_l0", "bias_ih_l0", "weight_hh_l0", "bias_hh_l0"))
             for sfx in (["", "_reverse"] if bidir else [""])]
    seq, fin = ops.rnn_layer(0 if rnn_type == "LSTM" else 1, lengths.to(DEV).to(torch.int32), xh, B, Lm,
                             quads, True)
    close(seq, out_r.transpose(0, 1), 2e-4, what="seq")
    for d in range(dirs):
        close(fin[d], hn[d], 2e-4, what=f"final state {d}")
    padded = torch.nn.functional.pad(seq, (0, 0, 0, 3))     # backward: a [:, :Lm] slice of [B, Lm+3, C]
    loss = (padded * wseq.to(DEV)).sum() + sum((fin[d] * wfin[d].to(DEV)).sum() for d in range(dirs))
    loss.backward()
    close(xh.grad.view(Lm, B, E), xr.grad, 5e-4, what="d input rows")
    for n, p in rnn.named_parameters():
        close(prm[n].grad, p.grad, 5e-4, what=f"d {n}")


def test_fused_bn_consumers(hip):
    """max-pool with the stem's BatchNorm+ReLU applied on the fly, and the dual-input block-end pass."""
    x = rnd(2, 18, 22, 64, seed=1)
    sc, sh = rnd(64, seed=2), rnd(64, seed=3)
    ref = F.max_pool2d(torch.relu(x * sc + sh).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    got = ops.maxpool3x3s2(x.to(DEV), sc.to(DEV), sh.to(DEV), in_relu=True)
    close(got, ref, 1e-6, what="maxpool(bn+relu)")
    ce = rnd(64, seed=33)
    ref = F.max_pool2d(torch.relu((x - ce) * sc + sh).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    got = ops.maxpool3x3s2(x.to(DEV), sc.to(DEV), sh.to(DEV), in_relu=True, in_center=ce.to(DEV))
    close(got, ref, 1e-6, what="maxpool(centered bn+relu)")
    a, b = rnd(3, 5, 7, 128, seed=4), rnd(3, 5, 7, 128, seed=5)
    s1, t1, s2, t2 = (rnd(128, seed=6 + i) for i in range(4))
    ref = torch.relu(a * s1 + t1 + b * s2 + t2)
    ad = a.to(DEV)
    got = ops.scale_shift_add_act(ad, s1.to(DEV), t1.to(DEV), b.to(DEV), s2.to(DEV), t2.to(DEV),
                                  act=1, out=ad)
    close(got, ref, 1e-6, what="dual-input block end")
    c1, c2 = rnd(128, seed=20), rnd(128, seed=21)
    ref = torch.relu((a - c1) * s1 + t1 + (b - c2) * s2 + t2)
    got = ops.scale_shift_add_act(a.to(DEV), s1.to(DEV), t1.to(DEV), b.to(DEV), s2.to(DEV),
                                  t2.to(DEV), act=1, c1=c1.to(DEV), c2=c2.to(DEV))
    close(got, ref, 1e-6, what="dual-input block end (centered)")
    # centered single-input apply, per-channel and per-sample vectors
    xs = rnd(3, 35, 128, seed=22)
    got = ops.scale_shift_act(xs.to(DEV), s1.to(DEV), t1.to(DEV), center=c1.to(DEV), act=1)
    close(got, torch.relu((xs - c1) * s1 + t1), 1e-6, what="centered apply")
    sN, tN, cN = rnd(3, 128, seed=23), rnd(3, 128, seed=24), rnd(3, 128, seed=25)
    got = ops.scale_shift_act(xs.to(DEV), sN.to(DEV), tN.to(DEV), center=cN.to(DEV),
                              rows_per_sample=35)
    close(got, (xs - cN[:, None]) * sN[:, None] + tN[:, None], 1e-6, what="centered per-sample")


# ------------------------------------------------------------------ trunk backward kernels
WGRAD_CASES = [
    # name,        N,  H,  W, Cin, Cout, k, s, p
    ("1x1",        2, 16, 16,  64, 128, 1, 1, 0),
    ("3x3",        2, 16, 16,  64,  64, 3, 1, 1),
    ("3x3s2",      3, 15, 17,  64, 128, 3, 2, 1),
    ("1x1s2",      2, 16, 16, 128, 256, 1, 2, 0),
    ("stem_rgb",   2, 64, 64,   3,  64, 7, 2, 3),
    ("stem_depth", 2, 32, 32,   1,  32, 7, 2, 3),
    ("ragged",     3,  7,  9,  36,  48, 3, 1, 1),
    ("deep",       4,  8,  8, 512, 512, 3, 1, 1),
    ("many_rows", 16, 32, 32,  64,  64, 3, 1, 1),     # split-K over 16384 pixels
    # round 6: wgrad_x6_kernel (three bf16 planes; Cin % 32 == 0, Cout % 32 == 0, >= 256 pixels)
    ("depth_32",   3, 16, 16,  32,  32, 3, 1, 1),     # a 64-row tile half empty
    ("wide_k",     2, 16, 16, 256,  64, 3, 1, 1),     # K = 2304: 18 column tiles
    ("cout_160",   2, 16, 16,  64, 160, 1, 1, 0),     # 128-row tiles, the second one partial
    ("ragged_m",   3, 13, 11,  64,  96, 3, 2, 1),     # 126 pixels... below the kernel's 256: fp32 kernel
    ("ragged_m2",  5, 13, 11,  64,  96, 3, 1, 1),     # 715 pixels: last chunk partial
    ("k_32",       2, 16, 16,  32, 128, 1, 1, 0),     # K = 32: a quarter of a column tile
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv2d_wgrad(hip, case):
    name, N, H, W, Cin, Cout, k, s, p = case
    x = rnd(N, H, W, Cin, seed=1)
    g = ops.conv_geometry(x, torch.empty(Cout, k, k, Cin), s, p)
    t = dict(x=x, dy=rnd(N, g["Ho"], g["Wo"], Cout, seed=2), dw=torch.zeros(Cout, k, k, Cin))
    cpu, gpu = both("conv2d_wgrad", t, dict(g=g))
    close(gpu["dw"], cpu["dw"], what="wgrad/" + name)
    # gradient-sized operands (1e-7: far below fp16's range -- the planes are bf16) and the fp32-MFMA
    # kernel behind option "wgrad_tile" = 1 agree with the same reference
    t2 = dict(t, dy=t["dy"] * 1e-7, dw=torch.zeros_like(t["dw"]))
    cpu2, gpu2 = both("conv2d_wgrad", t2, dict(g=g))
    close(gpu2["dw"], cpu2["dw"], what="wgrad/" + name + "/1e-7")
    with hip.options(wgrad_tile=1):
        cpu3, gpu3 = both("conv2d_wgrad", dict(t, dw=torch.zeros_like(t["dw"])), dict(g=g))
    close(gpu3["dw"], cpu3["dw"], what="wgrad/" + name + "/fp32-MFMA")
    # accumulate: dW += (a trunk's backward zeroes one arena for all its layers)
    pre = rnd(Cout, k, k, Cin, seed=9)
    cpu5, gpu5 = both("conv2d_wgrad", dict(t, dw=pre.clone()), dict(g=g, accumulate=True))
    close(gpu5["dw"], cpu5["dw"], what="wgrad/" + name + "/accumulate")
    assert float((cpu5["dw"] - pre - cpu["dw"]).abs().max()) < 1e-4 * float(cpu["dw"].abs().max())
    # fp16 planes with the power of two of dy (what bn_bwd / gn_bwd hand out): tiny and huge dy
    for scale in (1e-7, 1.0, 3e4):
        dy = t["dy"] * scale
        up = 2.0 ** (14 - math.frexp(float(dy.abs().max()))[1])
        pow2 = torch.stack([torch.full((8,), up), torch.full((8,), 1.0 / up)])
        t4 = dict(t, dy=dy, dw=torch.zeros_like(t["dw"]), dy_pow2=pow2)
        cpu4, gpu4 = both("conv2d_wgrad", t4, dict(g=g))
        close(gpu4["dw"], cpu4["dw"], what=f"wgrad/{name}/fp16 planes x{scale:g}")
        ref = cpu4["dw"].double()   # (close() carries an absolute 1e-6: relative, for the tiny dy)
        rel = float((gpu4["dw"].cpu().double() - ref).abs().max() / ref.abs().max())
        assert rel < 1e-4, (name, scale, rel)


@pytest.mark.parametrize("stride,k,pad", [(1, 3, 1), (2, 3, 1), (2, 1, 0), (1, 1, 0), (2, 7, 3)])
def test_conv_backward_matches_autograd(hip, stride, k, pad):
    """data + weight gradient of the host-side conv_backward (dgrad via the forward kernel
    on flipped taps / zero-inserted dY) against torch autograd."""
    from vlnce_amd.encoders import trunk_backward as tb
    N, H, W, Cin, Cout = 2, 14, 18, 32, 64
    x = rnd(N, H, W, Cin, seed=3)
    w = rnd(Cout, k, k, Cin, seed=4) * 0.2
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, stride, pad)
    dy = rnd(*y.shape, seed=5)
    y.backward(dy)
    dx, dw = tb.conv_backward(x.to(DEV), w.to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV),
                              stride, pad, True)
    close(dx, xr.grad.permute(0, 2, 3, 1), what="dgrad")
    close(dw, wr.grad, what="wgrad")


@pytest.mark.parametrize("M,Cc", [(512, 64), (1031, 48), (70000, 256), (9, 512), (300, 5), (4099, 2048),
                                  (100003, 64), (37, 4), (2000, 1028)])
@pytest.mark.parametrize("relu,batch,res", [(1, 1, 1), (1, 1, 0), (0, 0, 1), (0, 1, 0)])
def test_bn_bwd(hip, M, Cc, relu, batch, res):
    x = rnd(M, Cc, seed=1) * 2 + 0.5
    mean, var = x.mean(0), x.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    gamma = rnd(Cc, seed=2).abs() + 0.5
    y = (x - mean) * rstd * gamma + rnd(Cc, seed=3)
    if relu:
        y = torch.relu(y)
    t = dict(dy=rnd(M, Cc, seed=4), y=y, x=x, mean=mean, rstd=rstd, gamma=gamma,
             dx=torch.zeros(M, Cc), dres=torch.zeros(M, Cc) if res else None,
             dgamma=torch.zeros(Cc), dbeta=torch.zeros(Cc),
             workspace=torch.zeros(max(_lib.get_lib().bn_bwd_workspace_floats(M, Cc), 1)))
    cpu, gpu = both("bn_bwd", t, dict(M=M, Cc=Cc, relu=relu, use_batch_stats=batch))
    for k in ("dx", "dgamma", "dbeta") + (("dres",) if res else ()):
        close(gpu[k], cpu[k], 2e-4, what=f"bn_bwd/{k}")


def test_bn_bwd_matches_autograd(hip):
    M, Cc = 777, 96
    x = rnd(M, Cc, seed=1) * 1.5 - 0.3
    gamma, beta = rnd(Cc, seed=2).abs() + 0.5, rnd(Cc, seed=3)
    xr, gr, br = (v.clone().requires_grad_(True) for v in (x, gamma, beta))
    y = torch.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5))
    dy = rnd(M, Cc, seed=4)
    y.backward(dy)
    mean, rstd = x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
    dx, dg, db = torch.zeros(M, Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    hip.bn_bwd(dy.to(DEV), y.detach().to(DEV), x.to(DEV), mean.to(DEV), rstd.to(DEV),
               gamma.to(DEV), M, Cc, 1, 1, dx, None, dg, db,
               torch.empty(hip.bn_bwd_workspace_floats(M, Cc), device=DEV))
    close(dx, xr.grad, 2e-4, what="bn dx")
    close(dg, gr.grad, 2e-4, what="bn dgamma")
    close(db, br.grad, 2e-4, what="bn dbeta")


@pytest.mark.parametrize("N,HW,Cc,groups", [(3, 64, 32, 16), (2, 1024, 64, 32), (5, 16, 256, 128),
                                            (2, 49, 128, 1), (1, 4096, 32, 16), (3, 1061, 512, 16),
                                            (2, 300, 1024, 512), (2, 130, 6, 3)])
@pytest.mark.parametrize("relu,res", [(1, 1), (0, 0)])
def test_gn_bwd(hip, N, HW, Cc, groups, relu, res):
    x = rnd(N, HW, Cc, seed=1) * 2 + 0.3
    cpg = Cc // groups
    xg = x.view(N, HW, groups, cpg)
    mean = xg.mean((1, 3))
    rstd = torch.rsqrt(xg.var((1, 3), unbiased=False) + 1e-5)
    gamma = rnd(Cc, seed=2).abs() + 0.5
    y = ((xg - mean.view(N, 1, groups, 1)) * rstd.view(N, 1, groups, 1)).reshape(N, HW, Cc) * gamma \
        + rnd(Cc, seed=3)
    if relu:
        y = torch.relu(y)
    ws = _lib.get_lib().gn_bwd_workspace_floats(N, HW, Cc, groups)
    t = dict(dy=rnd(N, HW, Cc, seed=4), y=y.contiguous(), x=x, mean=mean.contiguous(),
             rstd=rstd.contiguous(), gamma=gamma, dx=torch.zeros(N, HW, Cc),
             dres=torch.zeros(N, HW, Cc) if res else None, dgamma=torch.zeros(Cc),
             dbeta=torch.zeros(Cc), workspace=torch.zeros(max(ws, 1)))
    cpu, gpu = both("gn_bwd", t, dict(Nimg=N, HW=HW, Cc=Cc, groups=groups, relu=relu))
    for k in ("dx", "dgamma", "dbeta") + (("dres",) if res else ()):
        close(gpu[k], cpu[k], 2e-4, what=f"gn_bwd/{k}")


def test_gn_bwd_matches_autograd(hip):
    N, H, W, Cc, groups = 2, 6, 5, 64, 16
    x = rnd(N, Cc, H, W, seed=1)
    gamma, beta = rnd(Cc, seed=2).abs() + 0.5, rnd(Cc, seed=3)
    xr, gr, br = (v.clone().requires_grad_(True) for v in (x, gamma, beta))
    y = torch.relu(F.group_norm(xr, groups, gr, br, 1e-5))
    dy = rnd(N, Cc, H, W, seed=4)
    y.backward(dy)
    nhwc = lambda v: v.detach().permute(0, 2, 3, 1).contiguous()
    xg = nhwc(x).view(N, H * W, groups, Cc // groups)
    mean = xg.mean((1, 3)).contiguous()
    rstd = torch.rsqrt(xg.var((1, 3), unbiased=False) + 1e-5).contiguous()
    dx, dg, db = torch.zeros(N, H, W, Cc, device=DEV), torch.zeros(Cc, device=DEV), \
        torch.zeros(Cc, device=DEV)
    ws = torch.zeros(max(hip.gn_bwd_workspace_floats(N, H * W, Cc, groups), 1), device=DEV)
    hip.gn_bwd(nhwc(dy).to(DEV), nhwc(y).to(DEV), nhwc(x).to(DEV), mean.to(DEV), rstd.to(DEV),
               gamma.to(DEV), N, H * W, Cc, groups, 1, dx, None, dg, db, ws)
    close(dx, nhwc(xr.grad), 2e-4, what="gn dx")
    close(dg, gr.grad, 2e-4, what="gn dgamma")
    close(db, br.grad, 2e-4, what="gn dbeta")


@pytest.mark.parametrize("N,H,W,Cc", [(2, 16, 16, 64), (3, 15, 17, 32), (1, 5, 7, 6), (2, 64, 64, 64)])
def test_maxpool_argmax_and_bwd(hip, N, H, W, Cc):
    x = rnd(N, H, W, Cc, seed=1)
    x[0, :4, :4] = 0.25  # ties: the first tap in scan order wins, as in torch
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    t = dict(x=x, y=torch.zeros(N, Ho, Wo, Cc), argmax=torch.zeros(N, Ho, Wo, Cc, dtype=torch.uint8))
    cpu, gpu = both("maxpool3x3s2_argmax", t, dict(N=N, H=H, W=W, Cc=Cc, Ho=Ho, Wo=Wo))
    assert torch.equal(gpu["y"].cpu(), cpu["y"])
    assert torch.equal(gpu["argmax"].cpu(), cpu["argmax"])
    t = dict(dy=rnd(N, Ho, Wo, Cc, seed=2), argmax=cpu["argmax"], dx=torch.zeros(N, H, W, Cc))
    cpu2, gpu2 = both("maxpool3x3s2_bwd", t, dict(N=N, H=H, W=W, Cc=Cc, Ho=Ho, Wo=Wo))
    close(gpu2["dx"], cpu2["dx"], 1e-6, what="maxpool bwd")
    # and against autograd end to end
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.max_pool2d(xr, 3, 2, 1).backward(t["dy"].permute(0, 3, 1, 2))
    close(gpu2["dx"], xr.grad.permute(0, 2, 3, 1), 1e-6, what="maxpool bwd vs autograd")


@pytest.mark.parametrize("N,H,W,Cc,OH,OW", [(2, 8, 8, 128, 4, 4), (2, 7, 7, 64, 4, 4),
                                            (3, 4, 4, 512, 1, 1), (1, 8, 8, 2048, 4, 4)])
def test_adaptive_avgpool_bwd(hip, N, H, W, Cc, OH, OW):
    t = dict(dy=rnd(N, OH, OW, Cc, seed=1), dx=torch.zeros(N, H, W, Cc))
    cpu, gpu = both("adaptive_avgpool_bwd", t, dict(N=N, H=H, W=W, Cc=Cc, OH=OH, OW=OW))
    close(gpu["dx"], cpu["dx"], 1e-6, what="adaptive avgpool bwd")


@pytest.mark.parametrize("N,H,W,Cc", [(2, 16, 20, 3), (3, 8, 8, 1), (1, 64, 64, 3)])
def test_space_to_depth2(hip, N, H, W, Cc):
    x = rnd(N, H, W, Cc, seed=1) * 100
    sc, sh = rnd(Cc, seed=2) * 0.01, rnd(Cc, seed=3)
    for scale, shift in ((sc, sh), (None, None)):
        t = dict(x=x, y=torch.full((N, H // 2 + 3, W // 2 + 3, 4 * Cc), 7.0), scale=scale,
                 shift=shift)
        cpu, gpu = both("space_to_depth2", t, dict(N=N, H=H, W=W, Cc=Cc, pad_lo=2, pad_hi=1))
        close(gpu["y"], cpu["y"], 1e-6, what="space_to_depth2")


@pytest.mark.parametrize("Cc,Cout,hw,N", [(3, 64, 64, 3), (1, 32, 32, 4)])
def test_s2d_stem_matches_direct_conv(hip, Cc, Cout, hw, N):
    """the frozen trunks' stem formulation (space-to-depth + 4x4 conv) vs F.conv2d 7x7/s2/p3."""
    x = (rnd(N, hw, hw, Cc, seed=1).abs() * 80).clamp(0, 255)
    w = rnd(Cout, 7, 7, Cc, seed=2) * 0.1
    sc, sh = torch.full((Cc,), 1 / 255.0), torch.zeros(Cc)
    ref = F.conv2d((x * sc + sh).permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, 2, 3)
    y, stats = ops.conv2d_nhwc(ops.space_to_depth2(x.to(DEV), 2, 1, sc.to(DEV), sh.to(DEV)),
                               ops.stem_weight_s2d(w.to(DEV)), 1, 0, want_stats=True)
    close(y.permute(0, 3, 1, 2), ref, what="s2d stem")


@pytest.mark.parametrize("N,hw,Cin,Cout,k,groups", [(3, 32, 32, 32, 3, 16), (2, 16, 64, 128, 1, 16),
                                                    (4, 8, 128, 256, 3, 16), (2, 4, 256, 512, 1, 16),
                                                    (2, 16, 32, 64, 3, 1)])
def test_conv_group_norm_from_epilogue_stats(hip, N, hw, Cin, Cout, k, groups, gn_path):
    """depth trunk: GroupNorm statistics taken from the convolution epilogue's tile moments
    (falls back to the activation pass when a tile would straddle two samples: the 4x4 case)."""
    x = rnd(N, hw, hw, Cin, seed=1)
    w = rnd(Cout, k, k, Cin, seed=2) * (Cin * k * k) ** -0.5
    gamma, beta = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    res = rnd(N, hw, hw, Cout, seed=5)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, 1, k // 2)
    ref = torch.relu(F.group_norm(ref, groups, gamma, beta, 1e-5) + res.permute(0, 3, 1, 2))
    got = ops.conv_group_norm_act(x.to(DEV), w.to(DEV), 1, k // 2, groups, gamma.to(DEV),
                                  beta.to(DEV), 1e-5, residual=res.to(DEV), act=1)
    close(got.permute(0, 3, 1, 2), ref, what="conv+GN(from tiles)")


@pytest.mark.parametrize("lstm", [False, True])
@pytest.mark.parametrize("T,N,H", [(40, 5, 512), (7, 64, 256), (9, 16, 256), (5, 3, 24),
                                   (100, 5, 512), (33, 16, 512), (12, 8, 64), (6, 1, 128)])
def test_masked_rnn_rollout_vs_torch_cells(hip, lstm, T, N, H):
    """ops.MaskedRNNSeqFn on the GPU (the T-step state-encoder rollout of a cached-feature DAgger
    batch / DD-PPO minibatch) against torch cells stepped on the CPU, forward and all gradients."""
    torch.manual_seed(5)
    D = 24
    cell = (torch.nn.LSTMCell if lstm else torch.nn.GRUCell)(D, H)
    x = torch.randn(T * N, D) * 0.5
    h0, c0 = torch.randn(N, H) * 0.3, torch.randn(N, H) * 0.3
    masks = (torch.rand(T, N) > 0.15).to(torch.uint8)
    masks[0] = 0
    wts, wh = torch.randn(T * N, H), torch.randn(N, H)

    def run(dev):
        xs = x.to(dev).requires_grad_(True)
        h = h0.to(dev).requires_grad_(True)
        c = c0.to(dev).requires_grad_(True) if lstm else None
        ps = [p.detach().to(dev).requires_grad_(True)
              for p in (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
        if dev == "cpu":
            hh, cc, outs = h, c, []
            for t in range(T):
                m = masks[t].float().unsqueeze(1)
                gi = xs[t * N:(t + 1) * N] @ ps[0].t() + ps[2]
                gh = (hh * m) @ ps[1].t() + ps[3]
                if lstm:
                    i, f, g, o = (gi + gh).chunk(4, 1)
                    cc = torch.sigmoid(f) * (cc * m) + torch.sigmoid(i) * torch.tanh(g)
                    hh = torch.sigmoid(o) * torch.tanh(cc)
                else:
                    ir, iz, inn = gi.chunk(3, 1)
                    hr, hz, hn = gh.chunk(3, 1)
                    r, z = torch.sigmoid(ir + hr), torch.sigmoid(iz + hz)
                    hh = (1 - z) * torch.tanh(inn + r * hn) + z * (hh * m)
                outs.append(hh)
            y, hT = torch.cat(outs), hh
        else:
            gi = ops.linear(xs, ps[0], ps[2])
            y, hT, _ = ops.MaskedRNNSeqFn.apply(lstm, gi, h, c, masks.view(-1).to(dev), ps[1], ps[3])
        loss = (y * wts.to(dev)).sum() + (hT * wh.to(dev)).sum()
        grads = torch.autograd.grad(loss, [xs, h] + ([c] if lstm else []) + ps)
        return y.detach(), [g.detach() for g in grads]

    y_ref, g_ref = run("cpu")
    y_hip, g_hip = run(DEV)
    close(y_hip, y_ref, what="rollout outputs")
    for i, (a, b) in enumerate(zip(g_hip, g_ref)):
        close(a, b, 3e-4, what=f"rollout grad {i}")


@pytest.mark.parametrize("T,N,H", [(100, 5, 512), (17, 16, 512), (9, 9, 256), (3, 2, 64)])
def test_gru_rollout_one_launch_equals_step_launches(hip, T, N, H):
    """vlnce_gru_rollout_fwd / _bwd (the whole recurrence in one persistent launch, H/16 workgroups
    handing the state over as tagged 64-bit pairs) against T x vlnce_rnn_step_fwd / _bwd: same saved
    tensors and gradients to fp32 rounding (the summation order of the recurrent dot products
    differs), repeated to catch a barrier that lets a workgroup read a stale state."""
    lib = ops.L()
    assert lib.gru_rollout_supported(N, H)
    torch.manual_seed(11)
    GH = 3 * H
    gi = (torch.randn(T, N, GH) * 0.7).to(DEV)
    h0 = (torch.randn(N, H) * 0.4).to(DEV)
    w = (torch.randn(GH, H) * H ** -0.5).to(DEV)
    b = (torch.randn(GH) * 0.1).to(DEV)
    mask = (torch.rand(T, N) > 0.1).to(torch.uint8).to(DEV)
    dout = torch.randn(T, N, H).to(DEV)
    dhf = torch.randn(N, H).to(DEV)

    def buffers():
        return [torch.full((T, N, H), float("nan"), device=DEV), torch.full((T, N, H), float("nan"), device=DEV),
                torch.full((T, N, GH), float("nan"), device=DEV), torch.full((T, N, H), float("nan"), device=DEV)]

    hp_s, out_s, gates_s, aux_s = buffers()
    h = h0
    for t in range(T):
        lib.rnn_step_fwd(False, gi[t], h, None, mask[t], w, b, hp_s[t], out_s[t], aux_s[t], gates_s[t], N, H)
        h = out_s[t]
    wt = w.t().contiguous()
    dgi_s, dgh_s = torch.empty(T, N, GH, device=DEV), torch.empty(T, N, GH, device=DEV)
    carry, acc = dhf.clone(), torch.empty(N, H, device=DEV)
    for t in range(T - 1, -1, -1):
        lib.rnn_step_bwd(False, dout[t], carry, None, gates_s[t], aux_s[t], hp_s[t], None, mask[t], wt,
                         dgi_s[t], dgh_s[t], acc, None, N, H)
    word = torch.empty(lib.gru_rollout_workspace_bytes(N, H), dtype=torch.uint8, device=DEV)
    for rep in range(3):
        hp_r, out_r, gates_r, aux_r = buffers()
        lib.gru_rollout_fwd(gi, h0, mask, w, b, hp_r, out_r, gates_r, aux_r, word, T, N, H)
        for a, r_, what in ((out_r, out_s, "out"), (hp_r, hp_s, "hp"), (gates_r, gates_s, "gates"),
                            (aux_r, aux_s, "aux")):
            close(a, r_, 2e-5, what=f"rollout fwd {what} (rep {rep})")
        dgi_r = torch.full((T, N, GH), float("nan"), device=DEV)
        dgh_r = torch.full((T, N, GH), float("nan"), device=DEV)
        dh0_r = torch.full((N, H), float("nan"), device=DEV)
        lib.gru_rollout_bwd(dout, dhf, gates_s, aux_s, hp_s, mask, wt, dgi_r, dgh_r, dh0_r, word, T, N, H)
        close(dgi_r, dgi_s, 1e-4, what=f"rollout dgi (rep {rep})")
        close(dgh_r, dgh_s, 1e-4, what=f"rollout dgh (rep {rep})")
        close(dh0_r, carry, 1e-4, what=f"rollout dh0 (rep {rep})")
    # no output gradient, no final-state gradient: NULL operands
    lib.gru_rollout_bwd(None, None, gates_s, aux_s, hp_s, mask, wt, dgi_r, dgh_r, dh0_r, word, T, N, H)
    assert float(dgi_r.abs().max()) == 0.0 and float(dh0_r.abs().max()) == 0.0


# ------------------------------------------------------------------ bf16-plane convolution kernel
def test_split_weights_is_exact(hip):
    """w == plane0 + plane1 + plane2 bit for bit (round-to-nearest split: 8 + 8 + 8 mantissa bits),
    including huge, tiny and negative values.  (Only where a residual would be an fp32 denormal,
    |w| < ~2^-110, the GPU flushes it: absolute error below 2^-126, checked separately.)"""
    torch.manual_seed(3)
    w = torch.randn(64, 3, 3, 32) * torch.exp(8 * torch.randn(64, 1, 1, 1))
    w.view(-1)[:6] = torch.tensor([0.0, -0.0, 1e-30, -3e38, 1.0, -1.0 + 2.0 ** -23])
    wg = w.to(DEV)
    planes = ops.split_weights(wg, ops.PLANES_BF16X6)
    assert planes is not None and planes.shape == (3, w.numel()) and planes._vlnce_fmt == 1
    assert ops.split_weights(wg, ops.PLANES_BF16X6) is planes  # cached on the tensor, per format
    assert ops.split_weights(wg, ops.PLANES_F16X3) is not planes
    p = (planes.cpu().to(torch.int32) & 0xFFFF) << 16
    f = p.view(torch.float32).double()
    assert torch.equal((f[0] + f[1] + f[2]).float(), w.view(-1))
    wg.mul_(2.0)  # written in place: split again
    assert ops.split_weights(wg, ops.PLANES_BF16X6) is not planes
    tiny = torch.full((1, 1, 1, 32), 1e-38, device=DEV)
    q = (ops.split_weights(tiny, ops.PLANES_BF16X6).cpu().to(torch.int32) & 0xFFFF) << 16
    assert (q.view(torch.float32).double().sum(0) - 1e-38).abs().max().item() < 2.0 ** -126


def _f16_planes_check(planes_i16, w_flat):
    """format 2 (include/vlnce_hip.h): planes {h * 2^11, (w - h) * 2^11 rounded, h}, h = fp16(w) --
    bit-exact against torch's own fp16 rounding, and h + l within 2^-22 |w| of w"""
    pl = planes_i16.cpu().view(torch.float16).float()
    h = w_flat.to(torch.float16).float()
    assert torch.equal(pl[2], h)
    assert torch.equal(pl[0], h * 2048.0)
    lo = ((w_flat - h) * 2048.0).to(torch.float16).float()
    assert torch.equal(pl[1], lo)
    err = (h.double() + lo.double() / 2048.0 - w_flat.double()).abs()
    assert bool((err <= w_flat.double().abs() * 2.0 ** -22 + 2.0 ** -36).all())


def test_split_weights_fp16_planes(hip):
    """vlnce_conv2d_split_weights, format 2: weights of ordinary size (|w| < 32), tiny ones (the low
    plane is scaled by 2^11: no subnormals down to 2^-25) and signed zeros."""
    torch.manual_seed(4)
    w = torch.randn(64, 3, 3, 32) * torch.exp(2 * torch.randn(64, 1, 1, 1)) * 0.05
    w.clamp_(-31.0, 31.0)
    w.view(-1)[:6] = torch.tensor([0.0, -0.0, 1e-6, -31.99, 1.0, -1.0 + 2.0 ** -23])
    wg = w.to(DEV)
    planes = ops.split_weights(wg, ops.PLANES_F16X3)
    assert planes.shape == (3, w.numel()) and planes._vlnce_fmt == 2
    _f16_planes_check(planes, w.view(-1))


def _conv_cases_under(hip, cases, want_path=None, **opts):
    """every case of `cases` through test_conv2d_fwd with the dispatch options `opts` set
    (vlnce_set_option: in-process, restored afterwards); `want_path`: the kernel family at
    least one of the cases must really have been given to (vlnce_conv2d_last_path)."""
    seen = set()
    opts.setdefault("m3", 0)   # (conv_m3_kernel would take every unit-test shape: they are all small)
    with hip.options(**opts):
        for case in cases:
            try:
                test_conv2d_fwd(hip, case)
            except AssertionError as e:
                raise AssertionError(f"case {case[0]} under options {opts}: {e}") from e
            seen.add(hip.conv2d_last_path())
    assert want_path is None or want_path in seen, (opts, seen)


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
def test_conv_x3_every_tile_shape(hip, tile):
    """conv_x3_kernel has four tile shapes chosen by problem size; force each one (option
    "x3_tile") over the conv cases, including several tiles per workgroup, ragged M, N below the
    tile width, the dual-input prologue and statistics partials."""
    _conv_cases_under(hip, CONV_CASES, want_path=1, x3_tile=tile)


# ------------------------------------------------------------------ patch-resident bf16-plane kernel
def test_pack_weights_layout_and_exactness(hip):
    """vlnce_conv2d_pack_weights: fragment order [Cout/32][K/16][3][64 lanes][8] with the k-slabs
    ordered (32-channel chunk, tap, 16-channel half), and plane0 + plane1 + plane2 == w bit for
    bit (round-to-nearest three-way split), also for huge / tiny / negative values."""
    torch.manual_seed(5)
    Cout, KH, KW, Cin = 64, 3, 3, 64
    w = torch.randn(Cout, KH, KW, Cin) * torch.exp(6 * torch.randn(Cout, 1, 1, 1))
    w.view(-1)[:6] = torch.tensor([0.0, -0.0, 1e-30, -3e38, 1.0, -1.0 + 2.0 ** -23])
    wg = w.to(DEV)
    frag = ops.pack_weights(wg, ops.PLANES_BF16X6)
    assert frag is not None and frag.numel() == w.numel() * 3 and frag._vlnce_fmt == 1
    assert ops.pack_weights(wg, ops.PLANES_BF16X6) is frag  # cached on the tensor, per format
    T, KS = KH * KW, KH * KW * Cin // 16
    f = ((frag.cpu().to(torch.int32) & 0xFFFF) << 16).view(torch.float32).double()
    f = f.view(Cout // 32, KS, 3, 64, 8).sum(2)  # the three planes added up: [nb, ks, lane, e]
    lane = torch.arange(64)
    for nb in range(Cout // 32):
        for ks in range(KS):
            s, ct = ks & 1, ks >> 1
            t, c = ct % T, ct // T
            n = nb * 32 + (lane & 31)
            ci = c * 32 + s * 16 + (lane >> 5) * 8
            want = torch.stack([w.view(Cout, T, Cin)[n, t, ci + e] for e in range(8)], 1)
            assert torch.equal(f[nb, ks].float(), want), (nb, ks)
    assert ops.pack_weights(torch.randn(48, 1, 1, 32, device=DEV)) is None  # Cout % 32 != 0
    # format 2: the same fragment order, planes {h * 2^11, (w - h) * 2^11, h}
    w2 = (torch.randn(Cout, KH, KW, Cin) * 0.05)
    f2 = ops.pack_weights(w2.to(DEV), ops.PLANES_F16X3)
    assert f2._vlnce_fmt == 2
    f2 = f2.cpu().view(Cout // 32, KS, 3, 64, 8)
    flat = torch.empty(3, Cout, T, Cin, dtype=torch.int16)
    for nb in range(Cout // 32):
        for ks in range(KS):
            s, ct = ks & 1, ks >> 1
            t, c = ct % T, ct // T
            n = nb * 32 + (lane & 31)
            ci = c * 32 + s * 16 + (lane >> 5) * 8
            for e in range(8):
                flat[:, n, t, ci + e] = f2[nb, ks, :, :, e]
    _f16_planes_check(flat.view(3, -1), w2.view(-1))
    wg.mul_(2.0)
    assert ops.pack_weights(wg, ops.PLANES_BF16X6) is not frag


P3_CASES = [
    # name,          N,  H,  W, Cin, Cout, k, s, p, extras
    ("p3_3x3_w64",   2, 64, 64,  64,  64, 3, 1, 1, dict(stats=True)),            # 264-row patches
    ("p3_3x3_w8",    5,  8,  8,  96, 128, 3, 1, 1, dict(scale=True, relu=True)),  # tiles across images
    ("p3_3x3_w4",    6,  4,  4, 128,  96, 3, 1, 1, dict(prologue=True, in_relu=True, center=True)),
    ("p3_3x3_rag",   3,  7,  9,  64,  96, 3, 1, 1, dict(prologue=True, in_relu=True)),
    ("p3_5x5_p2",    2, 11, 13,  32,  64, 5, 1, 2, dict(scale=True)),
    ("p3_3x3_p0",    2, 12, 10,  64,  32, 3, 1, 0, dict(stats=True)),
    ("p3_1x1_wide",  3, 24, 24, 128, 512, 1, 1, 0, dict(prologue=True, in_relu=True, center=True)),
    ("p3_1x1_stats", 3, 24, 24, 128, 512, 1, 1, 0, dict(stats=True)),
    ("p3_1x1_tall",  9, 32, 32,  64, 128, 1, 1, 0, dict(stats=True)),            # 256-row tiles
    ("p3_3x3_tall",  9, 32, 32,  64, 128, 3, 1, 1, dict(scale=True, relu=True)),  # 256-row patches
    ("p3_1x1_s2",    3, 18, 14, 128, 256, 1, 2, 0, dict(prologue=True, in_relu=True)),
    ("p3_1x1_rag",   1,  5,  7, 160,  96, 1, 1, 0, dict(scale=True, relu=True)),
    ("p3_dual_wide", 3, 20, 12, 128, 256, 1, 1, 0, dict(prologue=True, in_relu=True, center=True,
                                                        dual="bn")),
    ("p3_dual_tall", 9, 32, 32,  64, 128, 1, 1, 0, dict(prologue=True, in_relu=True, center=True,
                                                        dual="bn")),
    ("p3_dual_id",   2, 16, 16,  64,  64, 1, 1, 0, dict(prologue=True, in_relu=True, dual="identity")),
    ("p3_many_tiles", 40, 32, 32, 64, 64, 3, 1, 1, dict(stats=True)),            # several tiles per CU
    ("u3_rag_768",   2, 13, 11,  96, 768, 1, 1, 0, dict(stats=True)),             # ragged M, 3 column tiles
    ("u3_scale_act", 5, 16, 16,  64, 256, 1, 1, 0, dict(scale=True, relu=True)),  # epilogue scale/shift/act
    ("s3_rag_m",     1,  8, 20, 128, 256, 1, 1, 0, dict(stats=True)),             # 64 + 64 + 32 rows
    ("s3_many",     48, 32, 32,  64, 256, 1, 1, 0, dict(prologue=True, in_relu=True, center=True)),  # 3 tiles per workgroup
    ("s3_many_512", 40, 32, 32, 128, 512, 1, 1, 0, dict(stats=True)),             # two column tiles, 5 rounds
]


@pytest.mark.parametrize("case", P3_CASES, ids=[c[0] for c in P3_CASES])
def test_conv_p3(hip, case):
    with hip.options(m3=0):   # the persistent kernels' own default dispatch
        test_conv2d_fwd(hip, case)


def test_conv_m3_every_eligible_case(hip):
    """conv_m3_kernel (small launches: A fragments straight from global memory, four-way k split
    through LDS or four row blocks per workgroup) over every conv case it covers (option "m3" = 2):
    3x3 / 5x5 / 1x1, strides 1 and 2, padding 0..2, ragged M, tiles across image borders, prologue
    with centre and ReLU, epilogue scale / shift / ReLU / residual, statistics partials of 32 and
    16 rows, N = 32 (one column block) to 768."""
    _conv_cases_under(hip, CONV_CASES + P3_CASES, want_path=3, m3=2)
    _conv_cases_under(hip, CONV_CASES + P3_CASES, want_path=3, m3=3)   # two row blocks per workgroup


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6])
def test_conv_p3_every_tile_shape(hip, tile):
    """conv_p3_kernel has six tile shapes chosen by problem size; force each one (option
    "p3_tile") over the conv cases: several tiles per workgroup, ragged M, N below the tile width,
    patches across image borders, the dual-input prologue, statistics."""
    _conv_cases_under(hip, CONV_CASES + P3_CASES, want_path=2, p3_tile=tile)


@pytest.mark.parametrize("mode", [2, 3])
def test_conv_u3_forced(hip, mode):
    """conv_u3_kernel (1x1, no producer waves) takes a layer only when its tiles fill the CUs,
    which no unit-test shape does: force it (option "u3" = 2: 64-row tiles, 3: 128-row tiles)
    over the 1x1 cases with N >= 256 -- prologue, statistics, stride 2, dual input, ragged M."""
    _conv_cases_under(hip, CONV_CASES + P3_CASES, want_path=2, u3=mode)


def test_conv_s3_forced(hip):
    """conv_s3_kernel (short-K wide 1x1: B fragments resident, raw rows two tiles ahead, the
    previous tile's stores under this tile's MFMAs) takes a layer by default only when every CU
    gets at least four of its 64-row tiles; option "s3" = 2 sends every eligible shape to it:
    prologue + centre, statistics, epilogue scale / act, ragged M, one to five tiles per
    workgroup (the peeled first two and the loop), two column tiles."""
    _conv_cases_under(hip, CONV_CASES + P3_CASES, want_path=2, s3=2)


FORCED = [dict(), dict(m3=2), dict(u3=2), dict(u3=3), dict(s3=2), dict(p3_tile=1), dict(p3_tile=2),
          dict(p3_tile=3), dict(p3_tile=4), dict(p3_tile=5), dict(p3_tile=6), dict(x3_tile=1),
          dict(x3_tile=2), dict(x3_tile=3), dict(x3_tile=4)]


@pytest.mark.parametrize("forced", FORCED, ids=["-".join(f"{k}{v}" for k, v in f.items()) or "default"
                                                for f in FORCED])
def test_conv_kernels_three_bf16_planes(hip, forced):
    """The tests above run in the default plane format (fp16 planes, three products per multiply);
    the same cases, kernel families and forced tile shapes in format 1 (three bf16 planes, six
    products: option "conv_math" = 1 -- what the data-gradient launches of the trainable encoders
    use)."""
    _conv_cases_under(hip, CONV_CASES + P3_CASES, conv_math=1, **forced)


def test_fp16_planes_range_and_gradient_sized_operands(hip):
    """What format 2 promises (include/vlnce_hip.h): activations up to fp16's range are fine,
    an activation beyond it gives inf / NaN (never a wrong finite value), and operands of gradient
    size (1e-7) need format 1 -- the reason trunk_backward passes PLANES_BF16X6."""
    x = rnd(2, 16, 16, 64, seed=71).to(DEV)
    w = (rnd(64, 3, 3, 64, seed=72) * 0.04).to(DEV)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1)
    ref = ref.permute(0, 2, 3, 1).cpu()
    big = ops.conv2d_nhwc(x * 3.0e4, w, 1, 1, w_format=2)          # |x| up to ~1.2e5 > 65504
    assert not bool(torch.isfinite(big).all())
    ok = ops.conv2d_nhwc(x * 1.0e4, w, 1, 1, w_format=2)           # |x| up to ~4e4
    close(ok, ref.float() * 1.0e4, 1e-5, what="large activations")
    tiny2 = ops.conv2d_nhwc(x * 1.0e-7, w, 1, 1, w_format=2)
    tiny1 = ops.conv2d_nhwc(x * 1.0e-7, w, 1, 1, w_format=1)
    e2 = ((tiny2.double().cpu() - ref * 1e-7).abs().max() / (ref.abs().max() * 1e-7)).item()
    e1 = ((tiny1.double().cpu() - ref * 1e-7).abs().max() / (ref.abs().max() * 1e-7)).item()
    assert e1 < 1e-5 and e2 > 10 * e1, (e1, e2)


def test_frozen_weights_outside_the_fp16_range_fall_back_to_bf16_planes(hip):
    """A frozen filter bank with |w| >= 16 (format 2 holds |w| < 32) is pinned to format 1 by the
    encoder's weight cache (ops.check_weight_range: one read-back per parameter version) instead of
    overflowing: the convolution stays finite and right."""
    import torch.nn as nn
    from vlnce_amd.encoders.resnet_encoders import _WeightCache

    conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(DEV)
    conv.weight.requires_grad_(False)
    conv.weight.data.mul_(0.3)
    conv.weight.data[5, 7, 1, 1] = 40.0
    cache = _WeightCache()
    w = cache.conv(conv)
    assert w._vlnce_force_fmt == ops.PLANES_BF16X6 and ops.plane_format(None, w) == 1
    x = rnd(2, 12, 12, 64, seed=81).to(DEV)
    y = ops.conv2d_nhwc(x, w, 1, 1)
    ref = F.conv2d(x.permute(0, 3, 1, 2), conv.weight, padding=1).permute(0, 2, 3, 1)
    assert bool(torch.isfinite(y).all())
    close(y, ref, 1e-5, what="large frozen weight")
    assert not bool(torch.isfinite(ops.conv2d_nhwc(x, w, 1, 1, w_format=2)).all())   # what it avoids
    with torch.no_grad():
        conv.weight[5, 7, 1, 1] = 0.4   # (in place on the parameter: version bump, the cache repacks)
    w2 = cache.conv(conv)
    assert w2 is not w and w2._vlnce_force_fmt is None and ops.plane_format(None, w2) == 2


def test_options_are_explicit_state(hip):
    """vlnce_set_option / vlnce_get_option / vlnce_option_default: set, read back, restore; an
    unknown name is an error (the library reads no environment variable)."""
    for name in hip.OPTION_NAMES:
        d = hip.option_default(name)
        before = hip.get_option(name)
        with hip.options(**{name: d + 1}):
            if name in hip.PER_LAUNCH:
                # convolution-kernel options travel with each launch (vlnce_prologue.options): the
                # library's process state is not touched
                assert hip.get_option(name) == before and hip._tls.scoped[name] == d + 1
            else:
                assert hip.get_option(name) == d + 1
        assert hip.get_option(name) == before and getattr(hip._tls, "scoped", None) is None
        assert before == hip.option_default(name) or os.environ.get("VLNCE_" + name.upper())
    with pytest.raises(RuntimeError, match="unknown option"):
        hip.set_option("no_such_option", 1)
    with pytest.raises(RuntimeError, match="unknown dispatch option"):
        hip.options(no_such_option=1)


def test_per_launch_options_do_not_leak_between_launches(hip):
    """vlnce_prologue.options: the same 1x1 convolution dispatched three ways from one process --
    explicit per-call options, an enclosing lib.options block, nothing -- lands on the kernel each
    call names (vlnce_conv2d_last_path) and leaves the process values alone."""
    x, w = rnd(2, 16, 16, 64, seed=1).to(DEV), (rnd(256, 1, 1, 64, seed=2) * 0.1).to(DEV)
    g = ops.conv_geometry(x, w, 1, 0)
    y = torch.empty(2, 16, 16, 256, device=DEV)
    kw = dict(w_split=ops.split_weights(w), w_frag=ops.pack_weights(w))
    hip.conv2d_fwd(x, w, y, g, **kw)
    default_path = hip.conv2d_last_path()
    ref = y.clone()
    hip.conv2d_fwd(x, w, y, g, options=dict(conv_math=0), **kw)
    assert hip.conv2d_last_path() == 0                      # the fp32-MFMA kernel, this launch only
    close(y, ref, 1e-5, what="fp32-MFMA kernel vs default")
    hip.conv2d_fwd(x, w, y, g, **kw)
    assert hip.conv2d_last_path() == default_path
    assert default_path == 3                                # a small launch: conv_m3_kernel
    with hip.options(m3=0):
        hip.conv2d_fwd(x, w, y, g, **kw)
        assert hip.conv2d_last_path() != 3                  # conv_m3 switched off for the block
        hip.conv2d_fwd(x, w, y, g, options=dict(conv_math=0), **kw)   # explicit options win
        assert hip.conv2d_last_path() == 0
    hip.conv2d_fwd(x, w, y, g, **kw)
    assert hip.conv2d_last_path() == default_path
    assert hip.get_option("conv_math") == hip.option_default("conv_math") or os.environ.get("VLNCE_CONV_MATH")


@pytest.mark.parametrize("fmt", [2, 1], ids=["f16x3", "bf16x6"])
def test_conv_p3_matches_fp64_better_than_1e_6(hip, fmt):
    """Both plane formats keep the convolution fp32-class: relative rms error against an fp64
    convolution of the same operands below 1e-6 (torch's own fp32 conv: ~2e-7), post-ReLU-like
    (non-negative) and signed activations."""
    for sign in (False, True):
        x = rnd(4, 16, 16, 256, seed=21)
        x = (x if sign else x.abs()).to(DEV)
        w = (rnd(256, 3, 3, 256, seed=22) * (256 * 9) ** -0.5).to(DEV)
        y = ops.conv2d_nhwc(x, w, 1, 1, w_format=fmt)
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1)
        ref = ref.permute(0, 2, 3, 1)
        rms = ((y.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt().item()
        assert rms < 1e-6, (fmt, sign, rms)


def test_embedding_backward_matches_torch(hip):
    """vlnce_embedding_bwd (atomic scatter-add) against torch's embedding_dense_backward, with a
    padding index and repeated tokens (instruction_encoder.py:41-45: Embedding(vocab, 50, padding_idx=0))."""
    g = torch.Generator().manual_seed(6)
    tok = torch.randint(0, 97, (64, 80), generator=g)
    tok[:, 60:] = 0
    w_ref = rnd(97, 50, seed=31).requires_grad_(True)
    wts = rnd(64, 80, 50, seed=32)
    (F.embedding(tok, w_ref, padding_idx=0) * wts).sum().backward()
    w_hip = w_ref.detach().to(DEV).requires_grad_(True)
    out = ops.embedding(tok.to(DEV), w_hip, 0)
    assert torch.equal(out.cpu(), F.embedding(tok, w_ref.detach(), padding_idx=0))
    (out * wts.to(DEV)).sum().backward()
    close(w_hip.grad, w_ref.grad, 1e-5, what="embedding grad")
    assert w_hip.grad[0].abs().max().item() == 0.0   # the padding row gets no gradient


# ------------------------------------------------------------------ BatchNorm sums added by the conv
BN_CASES = [
    # name,             N,  H,  W, Cin, Cout, k, s, p, extras
    ("bn_1x1_64_256",   2, 16, 16,  64, 256, 1, 1, 0, dict()),
    ("bn_3x3_64_64",    2, 16, 16,  64,  64, 3, 1, 1, dict(prologue=True)),
    ("bn_3x3s2_128",    2, 16, 16, 128, 128, 3, 2, 1, dict()),
    ("bn_1x1_rag",      1,  5,  7, 160,  96, 1, 1, 0, dict(prologue=True)),       # M = 35
    ("bn_many_tiles",  40, 32, 32,  64, 256, 1, 1, 0, dict(prologue=True)),       # several tiles per workgroup
    ("bn_two_coltiles", 3, 24, 24, 128, 512, 1, 1, 0, dict()),                    # the column tile changes
    ("bn_dual",         3, 20, 12, 128, 256, 1, 1, 0, dict(prologue=True, dual=True)),
    ("bn_stem_7x7",     2, 32, 32,   3,  64, 7, 2, 3, dict()),                    # fp32 kernel: moments + finalize behind
    ("bn_split_k",      1,  4,  4, 512, 512, 3, 1, 1, dict()),                    # split-K: the same fall-back
    ("bn_big_mean",     2, 16, 16,  64, 128, 1, 1, 0, dict(offset=30.0)),         # |mean| >> std
]


def _bn_case(hip, case, **opts):
    name, N, H, W, Cin, Cout, k, s, pad, ex = case
    x = rnd(N, H, W, Cin, seed=41) + ex.get("offset", 0.0)
    w = rnd(Cout, k, k, Cin, seed=42) * (Cin * k * k) ** -0.5
    bn = torch.nn.BatchNorm2d(Cout, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=torch.Generator().manual_seed(43)) + 0.5)
        bn.bias.copy_(rnd(Cout, seed=44))
        bn.running_mean.copy_(rnd(Cout, seed=45))
        bn.running_var.copy_(torch.rand(Cout, generator=torch.Generator().manual_seed(46)) + 0.5)
    kw, xin = {}, x
    if ex.get("prologue"):
        isc, ish, ict = torch.rand(Cin) + 0.5, rnd(Cin, seed=47), rnd(Cin, seed=48)
        kw = dict(in_scale=isc.to(DEV), in_shift=ish.to(DEV), in_center=ict.to(DEV), in_relu=True)
        xin = (x - ict) * isc + ish
        if ex.get("dual"):
            x2 = rnd(N, H, W, Cin, seed=49)
            kw.update(x2=x2.to(DEV), side_out=torch.empty(N, H, W, Cin, device=DEV))
            xin = xin + x2
        xin = torch.relu(xin)
    raw = F.conv2d(xin.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=s,
                   padding=pad).permute(0, 2, 3, 1)
    M = raw.numel() // Cout
    mean = raw.reshape(M, Cout).mean(0)
    var = raw.reshape(M, Cout).var(0, unbiased=False)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn_g = torch.nn.BatchNorm2d(Cout, momentum=0.1).to(DEV)
    bn_g.load_state_dict(bn.state_dict())
    with hip.options(**opts):
        for rep in range(2):   # the accumulators must be left clean: the second pass sees zeros again
            y, (scale, beta, center) = ops.conv2d_bn_train(x.to(DEV), w.to(DEV), s, pad, bn_g, **kw)
            torch.cuda.synchronize()
            close(y, raw.float(), 1e-4, what=f"{name} raw output")
            close(center, mean.float(), 2e-5, what=f"{name} batch mean")
            want = bn.weight.double() / torch.sqrt(var + bn.eps)
            assert ((scale.cpu().double() - want).abs() / want.abs()).max().item() < 3e-5, name
            assert torch.equal(beta.cpu(), bn.bias.detach())
            assert ops._bn_state(bn_g).abs().max().item() == 0.0, (name, rep)
    unb = var * M / max(M - 1, 1)
    for _ in range(2):
        rm0 = 0.9 * rm0 + 0.1 * mean.float()
        rv0 = 0.9 * rv0 + 0.1 * unb.float()
    close(bn_g.running_mean, rm0, 2e-5, what=f"{name} running mean")
    close(bn_g.running_var, rv0, 3e-5, what=f"{name} running var")
    return hip.conv2d_last_path()


@pytest.mark.parametrize("case", BN_CASES, ids=[c[0] for c in BN_CASES])
def test_conv_bn_sums(hip, case):
    """vlnce_bn_sums + vlnce_bn_finalize_sums: raw output, pending normalisation (scale, center), running statistics and
    clean accumulators against an fp64 torch convolution + batch statistics, twice in a row."""
    _bn_case(hip, case)


@pytest.mark.parametrize("opts", [dict(m3=0, u3=2), dict(m3=0, u3=3), dict(m3=0, s3=2),
                                  dict(m3=0, p3=1, p3_tile=3),
                                  dict(m3=0, p3=0, u3=0, s3=0, x3_tile=1),
                                  dict(m3=0, p3=0, u3=0, s3=0, x3_tile=4),
                                  dict(conv_math=0), dict(m3=2)],
                         ids=["u3_64", "u3_128", "s3", "p3_tile3", "x3_tile1", "x3_tile4", "f32_fallback",
                              "m3"])
def test_conv_bn_sums_every_kernel(hip, opts):
    """the same through each convolution kernel (forced with the dispatch options) and through
    the fall-back (fp32 kernel: tile moments reduced into the sums behind the convolution)."""
    for case in BN_CASES:
        _bn_case(hip, case, **opts)


# ------------------------------------------------------------------ RGB stem from the frames
STEM7_CASES = [
    # name,          N, F, Hs, Ws, crop (y0, x0, H, W) | None, Cout, dtype, extra frame, mode
    ("u8_64",        3, 1, 64, 64, None,             64, torch.uint8,   False, "bn"),
    ("f32_ragged",   2, 1, 50, 38, None,             64, torch.float32, False, "eval"),
    ("u8_crop",      2, 1, 72, 96, (4, 10, 64, 80),  64, torch.uint8,   False, "eval"),
    ("pano_extra",   2, 3, 32, 48, None,             64, torch.float32, True,  "bn"),
    ("cout32",       2, 1, 70, 66, None,             32, torch.uint8,   False, "bn"),
    ("many_tiles",  20, 1, 128, 128, None,           64, torch.uint8,   False, "bn"),   # several tiles per workgroup
]


@pytest.mark.parametrize("fmt", [2, 1], ids=["f16x3", "bf16x6"])
@pytest.mark.parametrize("case", STEM7_CASES, ids=[c[0] for c in STEM7_CASES])
def test_stem7_from_frames(hip, case, fmt):
    """vlnce_stem7_fwd (7x7 / stride 2 / pad 3 on the 16-bit matrix pipe, both plane formats,
    straight from uint8 / fp32 frames, crop window, frame stack + masked extra frame, input
    transform) against the CPU contract (tests/hostsim.py: F.conv2d on the transformed frames): raw
    output + BatchNorm column sums, or the folded-BatchNorm + ReLU epilogue."""
    name, N, F_, Hs, Ws, crop, Cout, dtype, extra, mode = case
    g = torch.Generator().manual_seed(61)
    shape = (N, F_, Hs, Ws, 3) if F_ > 1 or extra else (N, Hs, Ws, 3)
    x = torch.randint(0, 256, shape, generator=g)
    x = x.to(dtype) if dtype == torch.uint8 else x.float() + torch.rand(shape, generator=g)
    if crop is not None:
        y0, x0, H, W = crop
        x = x[..., y0:y0 + H, x0:x0 + W, :]
    x2 = mask = None
    if extra:
        x2 = torch.randint(0, 256, (N,) + tuple(x.shape[-3:]), generator=g).to(x.dtype)
        mask = torch.tensor([1, 0][:N], dtype=torch.uint8)
    w = rnd(Cout, 7, 7, 3, seed=62) * 147 ** -0.5
    isc = torch.tensor([1 / 58.4, 1 / 57.1, 1 / 57.4])
    ish = torch.tensor([-2.12, -2.04, -1.80])
    esc, esh = torch.rand(Cout, generator=g) + 0.5, rnd(Cout, seed=63)

    def run(lib, dev):
        to = (lambda t: t.to(dev)) if dev != "cpu" else (lambda t: t)
        xs = (to(x), to(x2), to(mask)) if extra else to(x)
        fr = ops.frames(xs)
        wf = ops.stem7_pack_weights(to(w), fmt)
        yv = torch.empty((fr["images"], (fr["H"] - 1) // 2 + 1, (fr["W"] - 1) // 2 + 1, Cout), device=dev)
        acc = torch.zeros((ops.BN_SHARDS, Cout, 2), device=dev, dtype=torch.float64)
        if mode == "bn":
            lib.stem7_fwd(fr, to(isc), to(ish), wf, yv, bn=acc)
        else:
            lib.stem7_fwd(fr, to(isc), to(ish), wf, yv, scale=to(esc), shift=to(esh), act=ops.ACT_RELU)
        return yv, acc.sum(0)

    want, wacc = run(SIM, "cpu")
    got, gacc = run(hip, DEV)
    torch.cuda.synchronize()
    close(got, want, 1e-4, what=f"{name} output")
    if mode == "bn":
        M = want.numel() // Cout
        close((gacc[:, 0] / M).float(), (wacc[:, 0] / M).float(), 2e-5, what=f"{name} sum")
        close((gacc[:, 1] / M).float(), (wacc[:, 1] / M).float(), 3e-5, what=f"{name} sum of squares")


def test_prepare_weights_matches_the_per_tensor_entry_points(hip):
    """vlnce_conv2d_prepare_weights (all weight images of a trainable trunk in one launch) writes,
    bit for bit, what the per-tensor path writes for the permuted tensors: OHWI fp32, planes,
    fragments, in both formats, for the forward bank and for the data-gradient bank (taps reversed,
    channels swapped)."""
    torch.manual_seed(3)
    params = [torch.randn(s, device=DEV) * 0.2 for s in
              [(64, 32, 1, 1), (96, 64, 3, 3), (32, 128, 1, 1), (64, 64, 3, 3), (256, 64, 1, 1)]]
    jobs, want = [], []
    for p in params:
        w = p.permute(0, 2, 3, 1).contiguous()
        wt = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
        for tr, bank in ((0, w), (1, wt)):
            dst = torch.empty_like(bank)
            jobs.append((p, dst, hip.WP_F32, tr, 0))
            want.append(bank)
            for fmt in (ops.PLANES_BF16X6, ops.PLANES_F16X3):
                pl = torch.empty((3, p.numel()), device=DEV, dtype=torch.int16)
                fr = torch.empty((p.numel() * 3,), device=DEV, dtype=torch.int16)
                jobs += [(p, pl, hip.WP_PLANES, tr, fmt), (p, fr, hip.WP_FRAGMENTS, tr, fmt)]
                want += [ops.split_weights(bank, fmt), ops.pack_weights(bank, fmt)]
    plan = hip.weight_prep_plan(jobs)
    hip.conv2d_prepare_weights(plan)
    torch.cuda.synchronize()
    for k, ((p, dst, kind, tr, fmt), ref) in enumerate(zip(jobs, want)):
        assert torch.equal(dst.view(-1), ref.view(-1)), (k, tuple(p.shape), kind, tr, fmt)
    with pytest.raises(ValueError):   # Cin = 3: not a job (the stems keep the per-tensor path)
        hip.weight_prep_plan([(torch.randn(64, 3, 7, 7, device=DEV),
                               torch.empty(64, 7, 7, 3, device=DEV), hip.WP_F32, 0, 0)])


@pytest.mark.parametrize("batch", [1, 0])
def test_bn_bwd_power_of_two_bounds_dx(hip, batch):
    """`pow2` of vlnce_bn_bwd / vlnce_gn_bwd: an exact power of two 2^k (and its inverse) with
    max|dx| * 2^k <= 2^14 -- fp16's range with room to spare -- and not more than a few binades
    below it (the bound comes from per-channel maxima, not from dx itself)."""
    for scale in (1.0, 3e-7, 5e3):
        M, Cc, P = 5000, 96, 128
        x = rnd(M, Cc, seed=1) * 2 + 0.5
        mean, rstd = x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
        gamma = rnd(Cc, seed=2).abs() + 0.5
        y = torch.relu((x - mean) * rstd * gamma + rnd(Cc, seed=3))
        dy = rnd(M, Cc, seed=4) * scale
        dx, dg, db = (torch.zeros(M, Cc, device=DEV), torch.zeros(Cc, device=DEV),
                      torch.zeros(Cc, device=DEV))
        pow2 = torch.zeros(2, P, device=DEV)
        hip.bn_bwd(dy.to(DEV), y.to(DEV), x.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV), M, Cc,
                   1, batch, dx, None, dg, db,
                   torch.empty(hip.bn_bwd_workspace_floats(M, Cc), device=DEV), pow2)
        up, down = pow2[0].cpu(), pow2[1].cpu()
        assert (up == up[0]).all() and (down == down[0]).all() and float(up[0] * down[0]) == 1.0
        assert math.frexp(float(up[0]))[0] == 0.5          # a power of two
        top = float(dx.abs().max()) * float(up[0])
        assert 2.0 ** 9 < top <= 2.0 ** 14, (scale, top)
    # GroupNorm
    N, HW, Cc, G = 3, 700, 64, 16
    x = rnd(N, HW, Cc, seed=1) * 2 + 0.3
    xg = x.view(N, HW, G, Cc // G)
    mean, rstd = xg.mean((1, 3)).contiguous(), torch.rsqrt(xg.var((1, 3), unbiased=False) + 1e-5).contiguous()
    gamma = rnd(Cc, seed=2).abs() + 0.5
    y = torch.relu(((xg - mean.view(N, 1, G, 1)) * rstd.view(N, 1, G, 1)).reshape(N, HW, Cc) * gamma)
    for scale in (1.0, 2e-8):
        dy = rnd(N, HW, Cc, seed=4) * scale
        dx, dg, db = torch.zeros(N, HW, Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        pow2 = torch.zeros(2, 80, device=DEV)
        ws = torch.empty(hip.gn_bwd_workspace_floats(N, HW, Cc, G), device=DEV)
        hip.gn_bwd(dy.to(DEV), y.contiguous().to(DEV), x.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV),
                   N, HW, Cc, G, 1, dx, None, dg, db, ws, pow2)
        up = float(pow2[0, 0])
        assert math.frexp(up)[0] == 0.5 and float(pow2[1, 79]) * up == 1.0
        top = float(dx.abs().max()) * up
        assert 2.0 ** 9 < top <= 2.0 ** 14, (scale, top)


@pytest.mark.parametrize("k,stride,pad", [(1, 1, 0), (3, 1, 1), (3, 2, 1), (1, 2, 0)])
def test_data_gradient_on_fp16_planes_with_the_power_of_two(hip, k, stride, pad):
    """conv_backward(..., pow2): the data gradient of a TINY dy (1e-7: below fp16's normal range) in
    plane format 2, scaled by the exact power of two in the prologue and back in the epilogue,
    against fp64 -- the accuracy of the forward's fp16 planes (2^-22), not a flush to zero."""
    from vlnce_amd.encoders import trunk_backward as tb
    N, H, W, Cin, Cout = 2, 14, 14, 64, 96
    x = rnd(N, H, W, Cin, seed=1)
    w = rnd(Cout, k, k, Cin, seed=2) * 0.1
    Ho = (H + 2 * pad - k) // stride + 1
    dy = rnd(N, Ho, Ho, Cout, seed=3) * 1e-7
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w.permute(0, 3, 1, 2).double(),
                                     dy.permute(0, 3, 1, 2).double(), stride=stride, padding=pad)
    ref = ref.permute(0, 2, 3, 1)
    up = 2.0 ** (14 - math.frexp(float(dy.abs().max()))[1])
    pow2 = torch.empty(2, 128, device=DEV)
    pow2[0].fill_(up)
    pow2[1].fill_(1.0 / up)
    add = rnd(N, H, W, Cin, seed=4) * 1e-7
    dx, _ = tb.conv_backward(x.to(DEV), w.to(DEV), dy.to(DEV), stride, pad, True, add.to(DEV), pow2)
    err = float((dx.cpu().double() - ref - add.double()).abs().max()) / float(ref.abs().max())
    assert err < 2e-6, err
