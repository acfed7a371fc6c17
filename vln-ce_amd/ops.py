"""Tensor-level operators of the policy hot path on top of the C ABI.

Plain functions launch forward-only kernels (the frozen visual trunks);
torch.autograd.Function subclasses pair each forward kernel with its
hand-written backward for the trainable tail (linear / 1x1 conv, attention,
GRU / LSTM cells, masked state reset, packed-sequence selects).
All tensors are fp32; images are channels-last [N,H,W,C].
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3


def L():
    return _lib.get_lib()


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _rows2d(t):
    """view t as [rows, cols] with unit inner stride; returns (tensor, ld)."""
    if t.dim() != 2 or t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1)):
        t = t.reshape(-1, t.size(-1)).contiguous()
    ld = t.stride(0) if t.size(0) > 1 else t.size(1)
    return t, ld


# ----------------------------------------------------------------- conv (forward only)
PLANES_BF16X6, PLANES_F16X3 = 1, 2   # plane formats of include/vlnce_hip.h (vlnce_prologue.w_format)


def plane_format(w_format=None, w=None):
    """the plane format of a launch: the caller's (backward launches pass PLANES_BF16X6: gradients
    live below fp16's normal range), else the one the weight tensor itself demands (`check_weight_range`
    found a value outside format 2's range), else the effective "conv_math" option (default 2 = fp16
    planes, three plane products per multiply)."""
    if w_format:
        return int(w_format)
    forced = getattr(w, "_vlnce_force_fmt", None) if w is not None else None
    return forced or L().plane_format()


# Plane format of the trunks' BACKWARD convolutions (data gradient, weight gradient).  Gradients lie
# far below fp16's normal range, so on their own they need format 1; the normalisation backward
# kernels hand out the exact power of two that brings each gradient tensor to the top of fp16's
# range (vlnce_bn_bwd / vlnce_gn_bwd `pow2`), and with it they take format 2's three plane products
# instead of six.  VLNCE_GRAD_PLANES=bf16 keeps format 1 (no scaling).
GRAD_PLANES = (PLANES_BF16X6 if os.environ.get("VLNCE_GRAD_PLANES", "f16") in ("bf16", "1")
               else PLANES_F16X3)
_ZEROS = {}


def zeros_vec(device, n):
    """a cached all-zero fp32 vector of at least n elements (the zero `in_shift` / `shift` beside a
    power-of-two prologue / epilogue scale); never written"""
    z = _ZEROS.get(device)
    if z is None or z.numel() < n:
        z = torch.zeros(max(4096, n), device=device, dtype=torch.float32)
        _ZEROS[device] = z
    return z[:n]


F16_PLANES_MAX_WEIGHT = 16.0   # format 2 holds |w| < 32 (b1 * 2^11 in fp16); a factor of two in hand


def check_weight_range(w_ohwi):
    """For FROZEN weights (one host read-back per parameter version, in the eager first pass, never
    inside a capture): a filter bank with a value beyond format 2's range is pinned to format 1
    (three bf16 planes: fp32's exponent range) instead of producing inf / NaN.  Trainable weights
    change every step and are not checked (their overflow shows as NaN in the loss)."""
    if w_ohwi.is_cuda and not torch.cuda.is_current_stream_capturing():
        w_ohwi._vlnce_force_fmt = (PLANES_BF16X6 if float(w_ohwi.abs().max()) >= F16_PLANES_MAX_WEIGHT
                                   else None)
    return w_ohwi


def split_weights(w_ohwi, fmt=None):
    """The weights as three 16-bit planes (vlnce_conv2d_split_weights, format `fmt`) for
    conv_x3_kernel, or None where that kernel does not apply (Cin % 32 != 0).  Cached on
    the weight tensor itself, per format, and redone when the tensor is written in place; the
    encoders pass their packed OHWI tensors (one object per parameter version), so a frozen trunk
    splits once."""
    if not w_ohwi.is_cuda or w_ohwi.shape[-1] % 32 != 0 or w_ohwi.dtype != torch.float32:
        return None
    fmt = plane_format(fmt)
    cache = w_ohwi.__dict__.setdefault("_vlnce_split", {})
    hit = cache.get(fmt)
    if hit is None or hit[0] != w_ohwi._version:
        planes = torch.empty((3, w_ohwi.numel()), device=w_ohwi.device, dtype=torch.int16)
        L().conv2d_split_weights(w_ohwi, planes, fmt)
        planes._vlnce_fmt = fmt
        hit = (w_ohwi._version, planes)
        cache[fmt] = hit
    return hit[1]


def pack_weights(w_ohwi, fmt=None):
    """The weights as 16-bit-plane MFMA B fragments (vlnce_conv2d_pack_weights, format `fmt`) for
    the fragment kernels (conv_p3 / u3 / s3 / m3), or None where they do not apply (Cin or Cout not
    a multiple of 32).  Cached on the weight tensor like split_weights()."""
    if (not w_ohwi.is_cuda or w_ohwi.dtype != torch.float32 or w_ohwi.shape[-1] % 32 != 0
            or w_ohwi.shape[0] % 32 != 0):
        return None
    fmt = plane_format(fmt)
    cache = w_ohwi.__dict__.setdefault("_vlnce_frag", {})
    hit = cache.get(fmt)
    if hit is None or hit[0] != w_ohwi._version:
        Cout, KH, KW, Cin = w_ohwi.shape
        g = dict(N=1, H=KH, W=KW, Cin=Cin, Cout=Cout, KH=KH, KW=KW, stride=1, pad=0, Ho=1, Wo=1,
                 ldx=Cin, ldy=Cout)
        nbytes = L().conv2d_pack_bytes(g)
        if nbytes <= 0:
            return None
        frag = torch.empty((nbytes // 2,), device=w_ohwi.device, dtype=torch.int16)
        L().conv2d_pack_weights(w_ohwi, frag, g, fmt)
        frag._vlnce_fmt = fmt
        hit = (w_ohwi._version, frag)
        cache[fmt] = hit
    return hit[1]


def conv_geometry(x, w, stride, pad, ldx=None):
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    return dict(N=N, H=H, W=W, Cin=Cin, Cout=Cout, KH=KH, KW=KW, stride=stride, pad=pad,
                Ho=Ho, Wo=Wo, ldx=ldx or Cin, ldy=Cout)


def conv2d_nhwc(x, w_ohwi, stride, pad, *, in_scale=None, in_shift=None, in_center=None,
                in_relu=False, scale=None, shift=None, residual=None, act=ACT_NONE,
                want_stats=False, x2=None, in2_scale=None, in2_shift=None, in2_center=None,
                side_out=None, w_format=None):
    """y[N,Ho,Wo,Cout] = act((conv(prologue(x), w)) * scale + shift + residual), with
    prologue(x) = act((x - in_center) * in_scale + in_shift).
    want_stats=True additionally returns the BatchNorm partials of the RAW
    accumulator: (partial[tiles_m, Cout, 2], tiles_m, tile_rows).
    w_format: plane format of the launch (plane_format()); backward launches pass PLANES_BF16X6."""
    assert x.is_contiguous() and w_ohwi.is_contiguous()
    g = conv_geometry(x, w_ohwi, stride, pad)
    y = torch.empty((g["N"], g["Ho"], g["Wo"], g["Cout"]), device=x.device, dtype=torch.float32)
    stats = None
    partial = None
    if want_stats:
        tiles_m, tile_rows = L().conv2d_tiles(g)
        partial = torch.empty((tiles_m, g["Cout"], 2), device=x.device, dtype=torch.float32)
        stats = (partial, tiles_m, tile_rows)
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == y.shape
    dual = {}
    if x2 is not None:  # dual-input prologue (+ materialised transformed input), see the header
        assert x2.is_contiguous() and x2.shape == x.shape
        assert side_out is None or (side_out.is_contiguous() and side_out.shape == x.shape)
        dual = dict(x2=x2, in2_scale=in2_scale, in2_shift=in2_shift, in2_center=in2_center,
                    side_out=side_out)
    fmt = plane_format(w_format, w_ohwi)
    L().conv2d_fwd(x, w_ohwi, y, g, in_scale=in_scale, in_shift=in_shift, in_center=in_center,
                   in_relu=int(in_relu), **dual, scale=scale, shift=shift, residual=residual, ldr=g["Cout"], act=act,
                   stat_partial=partial, w_split=split_weights(w_ohwi, fmt),
                   w_frag=pack_weights(w_ohwi, fmt), w_format=fmt)
    return (y, stats) if want_stats else y


BN_SHARDS = 16   # VLNCE_BN_SHARDS of include/vlnce_hip.h


def _bn_state(bn):
    """the persistent column sums (vlnce_bn_sums.acc) of one BatchNorm layer; bn_finalize_sums
    leaves them zero.  Created on the first (eager) call of a layer, i.e. before any graph capture.
    Lives ON the module (a plain attribute: not a buffer, not in state_dict; freed with the module,
    never mistaken for another layer's after an id() is re-used -- ADVICE r4)."""
    w = bn.weight
    acc = bn.__dict__.get("_vlnce_acc")
    if acc is None or acc.device != w.device or acc.size(1) != w.numel():
        acc = torch.zeros((BN_SHARDS, w.numel(), 2), device=w.device, dtype=torch.float64)
        bn.__dict__["_vlnce_acc"] = acc
    return acc


def conv2d_bn_sums(x, w_ohwi, stride, pad, acc, **pro):
    """raw convolution output; the launch adds the output's per-channel {sum, sum of squares} to
    `acc` (vlnce_bn_sums).  `pro`: the operand-loader arguments of conv2d_nhwc."""
    assert x.is_contiguous() and w_ohwi.is_contiguous()
    g = conv_geometry(x, w_ohwi, stride, pad)
    y = torch.empty((g["N"], g["Ho"], g["Wo"], g["Cout"]), device=x.device, dtype=torch.float32)
    lib = L()
    ws = torch.empty(max(lib.conv2d_bn_workspace_bytes(g), 16), device=x.device, dtype=torch.uint8)
    if pro.get("x2") is not None:
        assert pro["x2"].is_contiguous() and pro["x2"].shape == x.shape
    if "in_relu" in pro:
        pro = dict(pro, in_relu=int(pro["in_relu"]))
    fmt = plane_format(None, w_ohwi)
    lib.conv2d_fwd(x, w_ohwi, y, g, **pro, ldr=g["Cout"], w_split=split_weights(w_ohwi, fmt),
                   w_frag=pack_weights(w_ohwi, fmt), bn=(acc, ws), w_format=fmt)
    return y


def bn_finalize_sums(acc, M, bn):
    """the column sums in `acc` -> pending normalisation (scale, shift = beta, center = mean);
    running statistics of `bn` updated like torch; acc is zero afterwards."""
    Cc = bn.weight.numel()
    scale = torch.empty(Cc, device=acc.device, dtype=torch.float32)
    mean = torch.empty_like(scale)
    L().bn_finalize_sums(acc, M, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum,
                         bn.running_mean, bn.running_var, scale, mean)
    return scale, bn.bias.detach(), mean


def conv2d_bn_train(x, w_ohwi, stride, pad, bn, **pro):
    """Convolution + train-mode BatchNorm STATISTICS without a pass over tile moments: the
    convolution adds its column sums (conv2d_bn_sums), a one-workgroup launch finishes them
    (vlnce_bn_finalize_sums).  Returns the RAW output and the pending normalisation (scale =
    gamma * rstd, shift = beta, center = batch mean) that the consumer's operand loader applies;
    the running statistics of `bn` are updated like torch.  (resnet_encoders.py:136-139: the
    frozen trunk's BatchNorm stays in training mode.)"""
    assert bn.momentum is not None
    acc = _bn_state(bn)
    try:
        y = conv2d_bn_sums(x, w_ohwi, stride, pad, acc, **pro)
        return y, bn_finalize_sums(acc, y.numel() // y.size(-1), bn)
    except Exception:
        # the sums must be zero between launches: a failure between the convolution and the
        # finalize (an allocation, an argument check) would otherwise leak this pass's sums into
        # the next forward's statistics
        if not (acc.is_cuda and torch.cuda.is_current_stream_capturing()):
            acc.zero_()
        raise


def bn_finalize(stats, M, gamma, beta, eps, momentum, running_mean, running_var):
    """batch-statistics BatchNorm from the conv's tile moments; updates the running stats.
    Returns the apply triple (scale, shift, center) = (gamma*rstd, beta, mean): consumers
    compute (x - center) * scale + shift, the reference's own arithmetic (the folded
    x*scale + (beta - mean*scale) loses bits when |mean| >> std)."""
    partial, tiles_m, tile_rows = stats
    Cc = partial.size(1)
    scale = torch.empty(Cc, device=partial.device, dtype=torch.float32)
    folded = torch.empty_like(scale)
    mean = torch.empty_like(scale)
    lib = L()
    wb = lib.bn_finalize_workspace_bytes(tiles_m, Cc)
    ws = torch.empty(wb // 8, device=partial.device, dtype=torch.float64) if wb else None
    lib.bn_finalize(partial, tiles_m, tile_rows, M, Cc, gamma, beta, float(eps), float(momentum),
                    running_mean, running_var, scale, folded, mean, None, workspace=ws)
    return scale, beta.detach(), mean


def scale_shift_act(x, scale, shift, *, center=None, rows_per_sample=0, residual=None,
                    act=ACT_NONE, out=None):
    """act((x - center) * scale + shift + residual); x: [..., C] contiguous; vectors: [C]
    (rows_per_sample=0) or [S, C]."""
    assert x.is_contiguous()
    Cc = x.size(-1)
    M = x.numel() // Cc
    y = out if out is not None else torch.empty_like(x)
    L().scale_shift_act(x, scale, shift, rows_per_sample, residual, y, M, Cc, act, center=center)
    return y


GN_SMALL_ELEMENTS = 1 << 20  # activations up to this size take the one-launch GroupNorm


def _gn_small(x):
    """True where GroupNorm is latency, not bandwidth: a few environments (act(), evaluation),
    16 * N workgroups re-reading a slab of at most a few hundred KB out of L2."""
    return (x.is_cuda and x.numel() <= GN_SMALL_ELEMENTS
            and os.environ.get("VLNCE_GN_SMALL", "1") != "0")


def group_norm_act(x, groups, gamma, beta, eps, *, residual=None, act=ACT_NONE):
    """GroupNorm over channels-last x[N,H,W,C] (+residual)(+act)."""
    assert x.is_contiguous()
    N, H, W, Cc = x.shape
    HW = H * W
    lib = L()
    if _gn_small(x):  # statistics + finalize + apply in ONE launch
        y = torch.empty_like(x)
        lib.group_norm_small(x, N, HW, Cc, groups, gamma, beta, float(eps), residual, act, y)
        return y
    chunks = lib.gn_chunks(HW)
    partial = torch.empty((N, chunks, Cc, 2), device=x.device, dtype=torch.float32)
    lib.gn_partial(x, N, HW, Cc, partial)
    scale = torch.empty((N, Cc), device=x.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    center = torch.empty_like(scale)
    lib.gn_finalize(partial, N, HW, Cc, groups, gamma, beta, float(eps), scale, shift,
                    center_out=center)
    return scale_shift_act(x, scale, shift, center=center, rows_per_sample=HW, residual=residual,
                           act=act)


def conv_group_norm_act(x, w_ohwi, stride, pad, groups, gamma, beta, eps, *, residual=None,
                        act=ACT_NONE):
    """act(GroupNorm(conv(x)) + residual).  The GroupNorm statistics come from the convolution's
    own epilogue when its M-tiles do not straddle samples; otherwise from a pass over y."""
    g = conv_geometry(x, w_ohwi, stride, pad)
    if x.is_cuda and g["N"] * g["Ho"] * g["Wo"] * g["Cout"] <= GN_SMALL_ELEMENTS and \
            os.environ.get("VLNCE_GN_SMALL", "1") != "0":
        # small activation: the plain convolution, then GroupNorm in one launch
        return group_norm_act(conv2d_nhwc(x, w_ohwi, stride, pad), groups, gamma, beta, eps,
                              residual=residual, act=act)
    y, stats = conv2d_nhwc(x, w_ohwi, stride, pad, want_stats=True)
    partial, _tiles_m, tile_rows = stats
    N, Ho, Wo, Cc = y.shape
    HW = Ho * Wo
    if HW % tile_rows != 0:
        return group_norm_act(y, groups, gamma, beta, eps, residual=residual, act=act)
    scale = torch.empty((N, Cc), device=y.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    center = torch.empty_like(scale)
    L().gn_finalize_tiles(partial, tile_rows, N, HW, Cc, groups, gamma, beta, float(eps), scale,
                          shift, center_out=center)
    return scale_shift_act(y, scale, shift, center=center, rows_per_sample=HW, residual=residual,
                           act=act)


def maxpool3x3s2(x, in_scale=None, in_shift=None, in_relu=False, in_center=None):
    """3x3/s2/p1 max pool of act((x-in_center)*in_scale+in_shift) (transform optional)."""
    N, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    L().maxpool3x3s2(x, y, N, H, W, Cc, Ho, Wo, in_scale, in_shift, int(in_relu),
                     in_center=in_center)
    return y


def scale_shift_add_act(x1, s1, t1, x2, s2, t2, act=ACT_NONE, out=None, c1=None, c2=None):
    """act((x1-c1)*s1+t1 + (x2-c2)*s2+t2), per-channel vectors; `out` may alias x1."""
    assert x1.is_contiguous() and x2.is_contiguous() and x1.shape == x2.shape
    Cc = x1.size(-1)
    y = out if out is not None else torch.empty_like(x1)
    L().scale_shift_add_act(x1, s1, t1, x2, s2, t2, y, x1.numel() // Cc, Cc, act, c1=c1, c2=c2)
    return y


def space_to_depth2(x, pad_lo, pad_hi, scale=None, shift=None):
    """[N,H,W,C] -> [N,H/2+pad,W/2+pad,4C] (2x2 pixel blocks into channels, zero border)."""
    assert x.is_contiguous()
    N, H, W, Cc = x.shape
    y = torch.empty((N, H // 2 + pad_lo + pad_hi, W // 2 + pad_lo + pad_hi, 4 * Cc),
                    device=x.device, dtype=torch.float32)
    L().space_to_depth2(x, y, N, H, W, Cc, pad_lo, pad_hi, scale, shift)
    return y


def stem_weight_s2d(w_ohwi):
    """7x7 taps [Cout,7,7,C] regrouped for the 4x4 convolution over space_to_depth2(x, 2, 1):
    pad to 8x8 with a zero first row/column, then (kh8, kw8) = (2*bh+dy, 2*bw+dx)."""
    Cout, KH, KW, Cc = w_ohwi.shape
    assert KH == 7 and KW == 7
    w8 = torch.zeros((Cout, 8, 8, Cc), device=w_ohwi.device, dtype=w_ohwi.dtype)
    w8[:, 1:, 1:] = w_ohwi
    return (w8.view(Cout, 4, 2, 4, 2, Cc).permute(0, 1, 3, 2, 4, 5)
            .reshape(Cout, 4, 4, 4 * Cc).contiguous())


# ----------------------------------------------------------------- observation ingest
def _frame_view(t):
    """[..., H, W, C] frames tensor (uint8 or fp32) -> (base, Hs, Ws, y0, x0): `base` is a
    contiguous-layout [images, Hs, Ws, C] tensor over the same memory and (y0, x0) the window
    `t` occupies in each image.  A contiguous tensor is its own base; a centre-crop VIEW
    (habitat's center_crop returns a slice, obs_transformers.py:69-79) is decoded from its strides
    so that the crop costs nothing; any other layout is materialised."""
    if t.dtype not in (torch.uint8, torch.float32):
        t = t.float()
    imgs = t.numel() // (t.size(-3) * t.size(-2) * t.size(-1)) if t.numel() else 0
    H, W, Cc = t.shape[-3:]
    if t.is_contiguous():
        return t.reshape(imgs, H, W, Cc), H, W, 0, 0
    t4 = None
    if t.dim() == 4:
        t4 = t
    elif t.dim() == 5 and t.stride(0) == t.size(1) * t.stride(1):  # [N, F] merges without a copy
        t4 = torch.as_strided(t, (imgs, H, W, Cc), t.stride()[1:], t.storage_offset())
    if t4 is not None and imgs > 1 and t4.stride(3) == 1 and t4.stride(2) == Cc:
        sN, sH = t4.stride(0), t4.stride(1)
        # (sN == 0: an expanded / broadcast frame, e.g. a zero history frame made with .expand)
        if sH % Cc == 0 and sH > 0 and sN > 0 and sN % sH == 0:
            Ws, Hs = sH // Cc, sN // sH
            off = t4.storage_offset() % sN
            y0, rem = divmod(off, sH)
            if rem % Cc == 0 and y0 + H <= Hs and rem // Cc + W <= Ws:
                x0 = rem // Cc
                try:
                    base = torch.as_strided(t4, (imgs, Hs, Ws, Cc), (sN, sH, Cc, 1),
                                            t4.storage_offset() - off)
                    return base, Hs, Ws, y0, x0
                except RuntimeError:  # the decoded base would reach outside the storage
                    pass
    return t.contiguous().reshape(imgs, H, W, Cc), H, W, 0, 0


def frames(x):
    """Frame descriptor of an encoder input: a [N,H,W,C] sensor, a [N,F,H,W,C] frame stack, or the
    tuple (stack [N,F,H,W,C], extra [N,H,W,C], mask [N] | None) the waypoint net feeds (12
    panorama frames + the done-masked history frame, waypoint_predictors.py:330-375)."""
    x2 = mask2 = None
    if isinstance(x, (tuple, list)):
        x, x2, mask2 = x
    F_ = x.size(1) if x.dim() == 5 else 1
    N = x.size(0)
    base, Hs, Ws, y0, x0 = _frame_view(x)
    H, W, Cc = x.shape[-3:]
    fr = dict(x=base, x2=None, mask2=None, N=N, F=F_, Hs=Hs, Ws=Ws, C=Cc, y0=y0, x0=x0, H=H, W=W)
    if x2 is not None:
        assert x2.shape == (N, H, W, Cc), (x2.shape, x.shape)
        if x2.dtype != base.dtype:
            x2 = x2.to(base.dtype)
        b2, Hs2, Ws2, y2, x2o = _frame_view(x2)
        if (Hs2, Ws2, y2, x2o) != (Hs, Ws, y0, x0):  # the extra frame shares the window geometry
            b2 = x2.contiguous()
            if (Hs, Ws, y0, x0) != (H, W, 0, 0):
                base = x.contiguous().reshape(N * F_, H, W, Cc)
                fr.update(x=base, Hs=H, Ws=W, y0=0, x0=0)
        fr["x2"] = b2
        if mask2 is not None:
            fr["mask2"] = mask2.reshape(-1).to(torch.uint8).contiguous()
    fr["images"] = N * (F_ + (x2 is not None))
    return fr


def frames_signature(x):
    parts = x if isinstance(x, (tuple, list)) else (x,)
    return tuple((tuple(t.shape), t.dtype) if t is not None else None for t in parts)


def frames_s2d(fr, pad_lo, pad_hi, scale=None, shift=None):
    """space_to_depth2 of every frame of the descriptor (+ per-channel input transform)."""
    y = torch.empty((fr["images"], fr["H"] // 2 + pad_lo + pad_hi, fr["W"] // 2 + pad_lo + pad_hi,
                     4 * fr["C"]), device=fr["x"].device, dtype=torch.float32)
    L().frames_s2d(fr, y, pad_lo, pad_hi, scale, shift)
    return y


def stem7_pack_weights(w_ohwi, fmt=None):
    """[Cout, 7, 7, 3] -> the B fragments vlnce_stem7_fwd reads: the three 16-bit planes of format
    `fmt` (include/vlnce_hip.h: 1 = exact bf16 split, 2 = fp16 {h * 2^11, (w - h) * 2^11, h}) of the
    filters laid out [Cout/32][11 k-slabs][3][64 lanes][8], k' = kh * 24 + kw * 3 + c."""
    fmt = plane_format(fmt)
    Cout = w_ohwi.size(0)
    assert tuple(w_ohwi.shape[1:]) == (7, 7, 3) and Cout % 32 == 0
    wk = torch.zeros((Cout, 7, 24), device=w_ohwi.device, dtype=torch.float32)
    wk[:, :, :21] = w_ohwi.detach().float().reshape(Cout, 7, 21)
    wk = F.pad(wk.reshape(Cout, 168), (0, 8))                      # [Cout, 176]
    if fmt == PLANES_F16X3:
        h = wk.to(torch.float16)
        lo = ((wk - h.float()) * 2048.0).to(torch.float16)
        planes = [(h.float() * 2048.0).to(torch.float16).view(torch.int16), lo.view(torch.int16),
                  h.view(torch.int16)]
    else:
        planes, r = [], wk
        for _ in range(3):
            hb = r.to(torch.bfloat16)
            planes.append(hb.view(torch.int16))
            r = r - hb.float()
    pl = torch.stack(planes, 0)                                    # [3, Cout, 176]
    # [3, nb, l31, ks, half, e] -> [nb, ks, 3, half, l31, e]  (lane = half * 32 + l31)
    frag = pl.view(3, Cout // 32, 32, 11, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
    frag._vlnce_fmt = fmt
    return frag


def stem7(fr, w_frag, Cout, in_scale=None, in_shift=None, *, scale=None, shift=None, act=ACT_NONE,
          bn_acc=None, w_format=None):
    """The 7x7 / stride-2 / pad-3 RGB stem in one launch from the frame descriptor
    (vlnce_stem7_fwd); raw output + BatchNorm column sums when bn_acc is given."""
    Ho, Wo = (fr["H"] - 1) // 2 + 1, (fr["W"] - 1) // 2 + 1
    y = torch.empty((fr["images"], Ho, Wo, Cout), device=fr["x"].device, dtype=torch.float32)
    L().stem7_fwd(fr, in_scale, in_shift, w_frag, y, scale=scale, shift=shift, act=act, bn=bn_acc,
                  w_format=w_format)
    return y


def frames_avgpool2(fr):
    y = torch.empty((fr["images"], fr["H"] // 2, fr["W"] // 2, fr["C"]), device=fr["x"].device,
                    dtype=torch.float32)
    L().frames_avgpool2(fr, y)
    return y


def frames_f32(fr, scale=None, shift=None):
    y = torch.empty((fr["images"], fr["H"], fr["W"], fr["C"]), device=fr["x"].device,
                    dtype=torch.float32)
    L().frames_f32(fr, y, scale, shift)
    return y


def frames_gather(srcs, crop=None):
    """out[n, f] = crop(srcs[f][n]): ObsStack (+ CenterCropperPerSensor) of F same-shaped sensors
    [N, Hs, Ws, C] in one pass, element type preserved.  crop = (y0, x0, H, W) | None."""
    s0 = srcs[0]
    N, Hs, Ws, Cc = s0.shape
    srcs = [t if t.is_contiguous() else t.contiguous() for t in srcs]
    assert all(t.shape == s0.shape and t.dtype == s0.dtype for t in srcs)
    y0, x0, H, W = crop if crop is not None else (0, 0, Hs, Ws)
    out = torch.empty((N, len(srcs), H, W, Cc), device=s0.device, dtype=s0.dtype)
    L().frames_gather(srcs, s0.element_size(), N, Hs, Ws, Cc, y0, x0, H, W, out)
    return out


def frames_resize_area(x, out_hw, crop=None):
    """habitat's image_resize_shortest_edge arithmetic (F.interpolate(mode="area") to out_hw, cast
    back to x.dtype) on [..., Hs, Ws, C] uint8 / fp32 frames, restricted to the window
    crop = (y0, x0, H, W) of the resized image (None = all of it)."""
    assert x.dtype in (torch.uint8, torch.float32), x.dtype
    lead, (Hs, Ws, Cc) = x.shape[:-3], x.shape[-3:]
    x = x if x.is_contiguous() else x.contiguous()
    OH, OW = out_hw
    y0, x0, H, W = crop if crop is not None else (0, 0, OH, OW)
    NF = 1
    for d in lead:
        NF *= int(d)
    out = torch.empty(tuple(lead) + (H, W, Cc), device=x.device, dtype=x.dtype)
    L().frames_resize_area(x, x.dtype == torch.uint8, NF, Hs, Ws, Cc, OH, OW, y0, x0, H, W, out)
    return out


def avgpool2x2(x):
    N, H, W, Cc = x.shape
    y = torch.empty((N, H // 2, W // 2, Cc), device=x.device, dtype=torch.float32)
    L().avgpool2x2(x, y, N, H, W, Cc)
    return y


def adaptive_avgpool(x, OH, OW):
    N, H, W, Cc = x.shape
    y = torch.empty((N, OH, OW, Cc), device=x.device, dtype=torch.float32)
    L().adaptive_avgpool(x, y, N, H, W, Cc, OH, OW, Cc)
    return y


def rowzero_mask(x3):
    """x3 [B, P, C] contiguous -> uint8 [B, P], 1 where the whole row is zero."""
    B, P, Cc = x3.shape
    m = torch.empty((B, P), device=x3.device, dtype=torch.uint8)
    L().rowzero_mask(x3, Cc, B * P, Cc, m)
    return m


# ----------------------------------------------------------------- linear / 1x1 conv
def _planes_eligible(x, M, K, N):
    """x[M,K] w[N,K]^T is large enough for the bf16-plane kernels (see _planes_gemm)"""
    return (x.is_cuda and M >= 1024 and K % 32 == 0 and N % 32 == 0 and 2.0 * M * N * K >= 1e9
            and os.environ.get("VLNCE_LINEAR_PLANES", "1") != "0")


def _planes_gemm(x, ldx, w, y, bias=None, act=ACT_NONE, w_format=None):
    """y[M,N] = act(x[M,K] w[N,K]^T + bias) through the convolution entry point (a 1x1 convolution
    over M pixels: the bf16-plane kernels, fp32-class arithmetic at 1.5-2.5x the fp32-MFMA GEMM's
    rate) when the product is large enough (>= 1024 rows and >= 1 GFLOP: rgb_kv of a 64-environment
    step, 1024 x 2112 -> 512, is 60 us on the fp32-MFMA GEMM) to pay for splitting the weights (two
    small launches per call: a trainable layer's weights change every step).  False: not taken, use lib.gemm."""
    M, K = x.shape
    N = w.size(0)
    if not _planes_eligible(x, M, K, N):
        return False
    Wd = next((d for d in range(min(M, 1024), 0, -1) if M % d == 0), 1)   # rows of <= 1024 pixels
    if M // Wd > 32767:
        return False
    g = dict(N=1, H=M // Wd, W=Wd, Cin=K, Cout=N, KH=1, KW=1, stride=1, pad=0, Ho=M // Wd, Wo=Wd,
             ldx=ldx, ldy=N)
    w4 = w.view(N, 1, 1, K)
    fmt = plane_format(w_format)
    L().conv2d_fwd(x, w4, y, g, shift=bias, ldr=N, act=act, w_split=split_weights(w4, fmt),
                   w_frag=pack_weights(w4, fmt), w_format=fmt)
    return True


class LinearFn(Function):
    """y = act(x W^T + b).  x [M,K] (row stride >= K), W [N,K], b [N] | None.
    dx_from > 0: only columns [dx_from, K) of the input gradient are wanted (the leading columns
    of x are a frozen trunk's features, the rest trainable spatial embeddings concatenated to
    them, resnet_encoders.py:199-204): dx is returned with the leading columns ZERO and the GEMM
    runs on the trailing ones only."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, dx_from=0):
        x, ldx = _rows2d(x)
        M, K = x.shape
        N = weight.size(0)
        w = weight if weight.is_contiguous() else weight.contiguous()
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        ctx.rows = _linear_rows_ok(x, ldx, w, M, N, K)
        # one row per environment: one launch, bias and activation included.  (Reductions past
        # 2048 -- rgb_linear, depth_linear: 16 / 8 column strips -- are as fast split over K by the
        # general GEMM's three launches, profiles/r05_f_linear_rows_per_layer.txt; their backward
        # is not.)
        if ctx.rows and K <= 2048:
            L().linear_rows_fwd(x, ldx, w, K, bias, act, y, N, M, N, K)
        elif not _planes_gemm(x, ldx, w, y, bias, act):
            L().gemm(x, ldx, 0, w, K, 0, y, N, M, N, K, shift=bias, act=act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.dx_from = int(dx_from) if 0 < int(dx_from) < K and int(dx_from) % 4 == 0 else 0
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        lib = L()
        M, K = x.shape
        N = w.size(0)
        ldx = x.stride(0) if M > 1 else K
        dz = _f32c(dy)
        dx = dw = db = None
        if ctx.rows and not ctx.dx_from and dz.data_ptr() % 16 == 0:
            # dx, dW, db and the activation's derivative in one launch (vlnce_linear_rows_bwd)
            def new(*shape):
                return torch.empty(shape, device=dz.device, dtype=torch.float32)

            dx = new(M, K) if ctx.needs_input_grad[0] else None
            dw = new(N, K) if ctx.needs_input_grad[1] else None
            db = new(N) if ctx.has_bias and ctx.needs_input_grad[2] else None
            lib.linear_rows_bwd(x, ldx, w, K, dz, N, y, N, ctx.act, dx, dw, db, M, N, K)
            return dx, dw, db, None, None
        if ctx.act != ACT_NONE:
            dz2 = torch.empty_like(dz)
            lib.act_bwd(dz, y, dz2, dz.numel(), ctx.act)
            dz = dz2
        if ctx.needs_input_grad[0]:
            c0 = ctx.dx_from
            if c0:
                dx = torch.zeros((M, K), device=dz.device, dtype=torch.float32)
                # dx[:, c0:] += dz[M,N] * W[:, c0:]   (the zero-fill doubles as the split-K one)
                lib.gemm(dz, N, 0, w[:, c0:], K, 1, dx[:, c0:], K, M, K - c0, N, accumulate=1)
            else:
                dx = torch.empty((M, K), device=dz.device, dtype=torch.float32)
                # dx[M,K] = dz[M,N] * W[N,K]   (B stored [K'=N, N'=K]); large: as dz (W^T)^T on the
                # bf16-plane kernels (one transpose of the weights, made only when that path
                # is taken -- ADVICE r4)
                # (a gradient operand: three bf16 planes, fp32's exponent range)
                if not (_planes_eligible(dz, M, N, K)
                        and _planes_gemm(dz, N, w.t().contiguous(), dx, w_format=PLANES_BF16X6)):
                    lib.gemm(dz, N, 0, w, K, 1, dx, K, M, K, N)
        if ctx.needs_input_grad[1]:
            dw = torch.empty((N, K), device=dz.device, dtype=torch.float32)
            # dW[N,K] = dz^T[N,M] * x[M,K]
            lib.gemm(dz, N, 1, x, ldx, 1, dw, K, N, K, M)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((N,), device=dz.device, dtype=torch.float32)
            lib.colsum(dz, N, M, N, db, 0)
        return dx, dw, db, None, None


def _linear_rows_ok(x, ldx, w, M, N, K):
    """True where the skinny-linear kernels apply (csrc/linear_rows.hip): at most 128 rows, N and K
    multiples of 4, 16-byte aligned rows.  VLNCE_LINEAR_ROWS=0 keeps the general GEMM (A/B)."""
    return (M <= 128 and N % 4 == 0 and K % 4 == 0 and K >= 4 and ldx % 4 == 0
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and os.environ.get("VLNCE_LINEAR_ROWS", "1") != "0")


def linear(x, weight, bias=None, act=ACT_NONE, dx_from=0):
    lead = x.shape[:-1]
    y = LinearFn.apply(x.reshape(-1, x.size(-1)) if x.dim() != 2 else x, weight, bias, act, dx_from)
    return y if x.dim() == 2 else y.view(*lead, weight.size(0))


# ----------------------------------------------------------------- categorical action head
class ActionHeadFn(Function):
    """normalised logits of Categorical(logits = x W^T + b), i.e. z - logsumexp(z, -1), in one
    launch; the whole backward (dx, dW, db) in one launch (vlnce_action_head_fwd / _bwd).
    `nan_count` (int32 [1] or None) counts the rows holding a NaN."""

    @staticmethod
    def forward(ctx, x, weight, bias, nan_count):
        x, ldx = _rows2d(_f32c(x) if x.dtype != torch.float32 else x)
        M, K = x.shape
        A = weight.size(0)
        w = weight if weight.is_contiguous() else weight.contiguous()
        out = torch.empty((M, A), device=x.device, dtype=torch.float32)
        L().action_head_fwd(x, ldx, w, bias, M, K, A, out, nan_count)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, out)
        return out

    @staticmethod
    def backward(ctx, dn):
        x, w, out = ctx.saved_tensors
        M, K = x.shape
        A = w.size(0)
        ldx = x.stride(0) if M > 1 else K
        dn = _f32c(dn)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty((M, K), device=dn.device, dtype=torch.float32) if need_x else None
        dw = (torch.empty((A, K), device=dn.device, dtype=torch.float32)
              if need_w or need_b else None)
        db = torch.empty((A,), device=dn.device, dtype=torch.float32) if need_b else None
        L().action_head_bwd(x, ldx, w, out, dn, M, K, A, dx, dw, db)
        return dx, (dw if need_w else None), db, None


_NAN_COUNT = {}


def action_head(x, weight, bias=None, count_nans=True):
    """(normalised logits [..., A], nan counter).  The counter is one persistent int32 per device,
    zero between calls: the caller that reads it non-zero resets it (`.zero_()`) before raising."""
    cnt = None
    if count_nans:
        cnt = _NAN_COUNT.get(x.device)
        if cnt is None:
            cnt = _NAN_COUNT[x.device] = torch.zeros(1, device=x.device, dtype=torch.int32)
    lead = x.shape[:-1]
    y = ActionHeadFn.apply(x.reshape(-1, x.size(-1)) if x.dim() != 2 else x, weight, bias, cnt)
    return (y if x.dim() == 2 else y.view(*lead, weight.size(0))), cnt


# ----------------------------------------------------------------- attention
class AttnFn(Function):
    """out[B,Dv] = softmax(mask(q K^T) * scale) V ; K [B,P,Dk], V [B,P,Dv] (views allowed).
    With `index` (int64 [B]): K [U,P,Dk], V [U,P,Dv], mask [U,P] are shared by groups of queries,
    query row b attends over block index[b] (vlnce_attn_fwd_shared); the gradients of K and V are
    the per-query gradients summed per block (vlnce_segment_sum)."""

    @staticmethod
    def forward(ctx, q, K, V, mask, mask_mode, scale, index=None):
        q = _f32c(q)
        U, P, Dk = K.shape
        B = q.size(0)
        Dv = V.size(2)

        def ld_of(t):
            if t.stride(2) != 1 or t.stride(0) != P * t.stride(1):
                t = t.contiguous()
            return t, t.stride(1)

        K, ldk = ld_of(K)
        V, ldv = ld_of(V)
        if index is None:
            assert U == B, (U, B)
        else:
            index = index.contiguous()
            assert index.dtype == torch.int64 and index.numel() == B
        out = torch.empty((B, Dv), device=q.device, dtype=torch.float32)
        attn = torch.empty((B, P), device=q.device, dtype=torch.float32)
        mm = 0 if mask is None else int(mask_mode)
        L().attn_fwd(q, K, ldk, V, ldv, mask, mm, float(scale), out, attn, B, P, Dk, Dv,
                     kv_index=index)
        ctx.save_for_backward(q, K, V, attn, mask, index)
        ctx.cfg = (mm, float(scale), ldk, ldv)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, K, V, attn, mask, index = ctx.saved_tensors
        mm, scale, ldk, ldv = ctx.cfg
        U, P, Dk = K.shape
        B = q.size(0)
        Dv = V.size(2)
        dout = _f32c(dout)
        dq = torch.empty((B, Dk), device=q.device, dtype=torch.float32)
        dK = torch.empty((B, P, Dk), device=q.device, dtype=torch.float32)
        dV = torch.empty((B, P, Dv), device=q.device, dtype=torch.float32)
        L().attn_bwd(dout, q, K, ldk, V, ldv, mask, mm, scale, attn, dq, dK, Dk, dV, Dv, B, P, Dk, Dv,
                     kv_index=index)
        if index is not None:
            dKu = torch.empty((U, P, Dk), device=q.device, dtype=torch.float32)
            dVu = torch.empty((U, P, Dv), device=q.device, dtype=torch.float32)
            L().segment_sum(dK, index, B, U, P * Dk, dKu)
            L().segment_sum(dV, index, B, U, P * Dv, dVu)
            dK, dV = dKu, dVu
        return dq, dK, dV, None, None, None, None


def attention(q, K, V, mask=None, mask_mode=1, scale=1.0, index=None):
    return AttnFn.apply(q, K, V, mask, mask_mode, scale, index)


class AttnKVFn(Function):
    """AttnFn over ONE projection that holds keys and values side by side, kv [B, P, Dk + Dv]
    (the rgb_kv / depth_kv 1x1 convolutions of cma_policy.py:258-274 and
    waypoint_predictors.py:462-490, torch.split there): the kernels read K and V in place through
    their row stride, and the backward writes dK and dV into the two column ranges of one d_kv --
    slicing kv in Python cost, per attention, two zero-fills, two strided copies and an add in the
    backward graph (the autograd of the two slices)."""

    @staticmethod
    def forward(ctx, q, kv, dk, mask, mask_mode, scale):
        q = _f32c(q)
        if kv.stride(2) != 1 or kv.stride(0) != kv.size(1) * kv.stride(1) or kv.dtype != torch.float32:
            kv = _f32c(kv)
        B, P, D = kv.shape
        dv = D - dk
        ld = kv.stride(1)
        out = torch.empty((B, dv), device=q.device, dtype=torch.float32)
        attn = torch.empty((B, P), device=q.device, dtype=torch.float32)
        mm = 0 if mask is None else int(mask_mode)
        L().attn_fwd(q, kv[..., :dk], ld, kv[..., dk:], ld, mask, mm, float(scale), out, attn, B, P, dk, dv,
                     kv_index=None)
        ctx.save_for_backward(q, kv, attn, mask)
        ctx.cfg = (mm, float(scale), dk, dv, ld)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, attn, mask = ctx.saved_tensors
        mm, scale, dk, dv, ld = ctx.cfg
        B, P, D = kv.shape
        dout = _f32c(dout)
        dq = torch.empty((B, dk), device=q.device, dtype=torch.float32)
        dkv = torch.empty((B, P, D), device=q.device, dtype=torch.float32)
        L().attn_bwd(dout, q, kv[..., :dk], ld, kv[..., dk:], ld, mask, mm, scale, attn, dq,
                     dkv[..., :dk], D, dkv[..., dk:], D, B, P, dk, dv, kv_index=None)
        return dq, dkv, None, None, None, None


def attention_kv(q, kv, dk, mask=None, mask_mode=1, scale=1.0):
    """softmax(mask(q K^T) * scale) V with K = kv[..., :dk], V = kv[..., dk:]."""
    return AttnKVFn.apply(q, kv, dk, mask, mask_mode, scale)


# ----------------------------------------------------------------- row utilities
class EmbeddingFn(Function):
    """F.embedding with the backward as one atomic scatter-add launch (vlnce_embedding_bwd)."""

    @staticmethod
    def forward(ctx, tokens, weight, padding_idx):
        ctx.save_for_backward(tokens)
        ctx.padding_idx = padding_idx
        ctx.wshape = weight.shape
        return torch.nn.functional.embedding(tokens, weight, padding_idx=padding_idx)

    @staticmethod
    def backward(ctx, g):
        (tokens,) = ctx.saved_tensors
        gw = torch.zeros(ctx.wshape, device=g.device, dtype=torch.float32)
        L().embedding_bwd(tokens.contiguous(), _f32c(g).reshape(-1, ctx.wshape[1]), gw,
                          ctx.padding_idx)
        return None, gw, None


def embedding(tokens, weight, padding_idx=None):
    if weight.requires_grad and torch.is_grad_enabled() and weight.is_cuda:
        return EmbeddingFn.apply(tokens, weight, padding_idx)
    return torch.nn.functional.embedding(tokens, weight, padding_idx=padding_idx)


class MaskRowsFn(Function):
    @staticmethod
    def forward(ctx, x, mask_u8):
        x = _f32c(x)
        B, H = x.shape
        out = torch.empty_like(x)
        L().mask_rows(x, mask_u8, out, B, H)
        ctx.save_for_backward(mask_u8)
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        g = _f32c(g)
        out = torch.empty_like(g)
        L().mask_rows(g, m, out, g.size(0), g.size(1))
        return out, None


def mask_rows(x, mask_u8):
    return MaskRowsFn.apply(x, mask_u8)


class SelectRowsFn(Function):
    """out[b] = mask[b] ? a[b] : b_[b]  (either operand may be None = zeros)."""

    @staticmethod
    def forward(ctx, mask_u8, a, b_):
        ref = a if a is not None else b_
        a = _f32c(a) if a is not None else None
        b_ = _f32c(b_) if b_ is not None else None
        B, H = ref.shape
        out = torch.empty((B, H), device=ref.device, dtype=torch.float32)
        L().select_rows(mask_u8, a, b_, out, B, H)
        ctx.save_for_backward(mask_u8)
        ctx.has = (a is not None, b_ is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        g = _f32c(g)
        B, H = g.shape
        ga = gb = None
        if ctx.has[0] and ctx.needs_input_grad[1]:
            ga = torch.empty_like(g)
            L().select_rows(m, g, None, ga, B, H)
        if ctx.has[1] and ctx.needs_input_grad[2]:
            gb = torch.empty_like(g)
            L().select_rows(m, None, g, gb, B, H)
        return None, ga, gb


def select_rows(mask_u8, a, b_):
    return SelectRowsFn.apply(mask_u8, a, b_)


class MeanRowsFn(Function):
    """y[b, c] = mean_p x[b, p, c]"""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        B, P, Cc = x.shape
        y = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
        L().mean_rows(x, y, B, P, Cc)
        ctx.P = P
        return y

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.P).unsqueeze(1).expand(-1, ctx.P, -1)


def mean_rows(x):
    return MeanRowsFn.apply(x)


# ----------------------------------------------------------------- recurrent cells
class GRUGatesFn(Function):
    """h' = GRU pointwise stage; gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh  ([B,3H], r|z|n)."""

    @staticmethod
    def forward(ctx, gi, gh, h_prev):
        gi, gh, h_prev = _f32c(gi), _f32c(gh), _f32c(h_prev)
        B, H = h_prev.shape
        h = torch.empty_like(h_prev)
        gates = torch.empty((B, 3 * H), device=h.device, dtype=torch.float32)
        hn = torch.empty_like(h_prev)
        L().gru_gates_fwd(gi, gh, h_prev, None, h, gates, hn, B, H)
        ctx.save_for_backward(gates, hn, h_prev)
        return h

    @staticmethod
    def backward(ctx, dh):
        gates, hn, h_prev = ctx.saved_tensors
        B, H = h_prev.shape
        dh = _f32c(dh)
        dgi = torch.empty_like(gates)
        dgh = torch.empty_like(gates)
        dhp = torch.empty_like(h_prev)
        L().gru_gates_bwd(dh, gates, hn, h_prev, None, dgi, dgh, dhp, B, H)
        return dgi, dgh, dhp


class LSTMGatesFn(Function):
    """(h', c') = LSTM pointwise stage; gates pre-activation = gi + gh ([B,4H], i|f|g|o)."""

    @staticmethod
    def forward(ctx, gi, gh, c_prev):
        gi, gh, c_prev = _f32c(gi), _f32c(gh), _f32c(c_prev)
        B, H = c_prev.shape
        h = torch.empty_like(c_prev)
        c = torch.empty_like(c_prev)
        gates = torch.empty((B, 4 * H), device=h.device, dtype=torch.float32)
        L().lstm_gates_fwd(gi, gh, c_prev, None, h, c, gates, B, H)
        ctx.save_for_backward(gates, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c = ctx.saved_tensors
        B, H = c_prev.shape
        dh = _f32c(dh) if dh is not None else None
        dc = _f32c(dc) if dc is not None else None
        dg = torch.empty_like(gates)
        dcp = torch.empty_like(c_prev)
        L().lstm_gates_bwd(dh, dc, gates, c_prev, c, None, dg, dcp, B, H)
        return dg, dg, dcp


def gru_cell(x_gates, h_prev, w_hh, b_hh):
    gh = linear(h_prev, w_hh, b_hh)
    return GRUGatesFn.apply(x_gates, gh, h_prev)


def lstm_cell(x_gates, h_prev, c_prev, w_hh, b_hh):
    gh = linear(h_prev, w_hh, b_hh)
    return LSTMGatesFn.apply(x_gates, gh, c_prev)


_ROLLOUT_WS = {}


def _rollout_workspace(dev, N, H):
    """The exchange area of a persistent rollout launch: one per (device, stream) -- launches on
    one stream are ordered, so they can share it; the library zeroes it per call."""
    nbytes = L().gru_rollout_workspace_bytes(N, H)
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)  # lives in the graph's own pool
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    w = _ROLLOUT_WS.get(key)
    if w is None or w.numel() < nbytes:
        w = _ROLLOUT_WS[key] = torch.empty(max(nbytes, 1 << 19), dtype=torch.uint8, device=dev)
    return w


class MaskedRNNSeqFn(Function):
    """T > 1 steps of a masked GRU / LSTM state encoder as ONE autograd node (habitat
    RNNStateEncoder.seq_forward semantics: h (and c) are multiplied by the not-done mask of step
    t before step t).  A T-step rollout of N episodes -- the cached-feature DAgger batch
    (dagger_trainer.py:39-114) or a DD-PPO minibatch (rollout_storage.py:154-276) -- costs
    ONE launch forward and ONE backward for a GRU over N <= 16 episodes (gru_rollout.hip: the
    recurrent weights stay in the registers of H/16 resident workgroups, one device-scope
    barrier per step); otherwise 1-3 launches per step forward (mask, h W_hh^T, gates) and 2-3
    backward (gates', dgates W_hh, mask + skip add) instead of one autograd cell per step.  The
    recurrent weight / bias gradients are ONE GEMM / column sum over all T*N rows instead of T
    accumulated ones.

    forward(lstm, gi [T*N, G*H], h0 [N,H], c0 [N,H] | None, mask_u8 [T*N], w_hh [G*H,H], b_hh)
      -> out [T*N, H], h_T [N,H], c_T [N,H] (c_T = h_T for a GRU)."""

    @staticmethod
    def forward(ctx, lstm, gi, h0, c0, mask, w_hh, b_hh):
        lib = L()
        gi, h0, w_hh = _f32c(gi), _f32c(h0), _f32c(w_hh)
        N, H = h0.shape
        T = gi.size(0) // N
        GH = gi.size(1)
        dev = gi.device
        hp = torch.empty((T, N, H), device=dev, dtype=torch.float32)   # masked h_{t-1}
        out = torch.empty((T, N, H), device=dev, dtype=torch.float32)
        gates = torch.empty((T, N, GH), device=dev, dtype=torch.float32)
        aux = torch.empty((T, N, H), device=dev, dtype=torch.float32)  # GRU: hn; LSTM: c_t
        gh = torch.empty((N, GH), device=dev, dtype=torch.float32)
        gi3, m2 = gi.view(T, N, GH), mask.view(T, N)
        h, c = h0, (_f32c(c0) if lstm else None)
        b_hh = _f32c(b_hh)
        ctx.rollout = (not lstm) and T > 1 and lib.gru_rollout_supported(N, H)
        if ctx.rollout:  # the whole recurrence in ONE launch (gru_rollout.hip)
            lib.gru_rollout_fwd(gi3, h0, m2.contiguous(), w_hh, b_hh, hp, out, gates, aux,
                                _rollout_workspace(dev, N, H), T, N, H)
            ctx.lstm, ctx.dims = lstm, (T, N, H, GH)
            ctx.save_for_backward(hp, gates, aux, m2, w_hh, h0)
            return out.view(T * N, H), out[T - 1], out[T - 1]
        fused = lib.rnn_step_supported(N, H, lstm)
        for t in range(T):
            if fused:  # mask, h W_hh^T and the gates of this step in ONE launch
                lib.rnn_step_fwd(lstm, gi3[t], h, c, m2[t], w_hh, b_hh, hp[t], out[t], aux[t],
                                 gates[t], N, H)
                if lstm:
                    c = aux[t]
                h = out[t]
                continue
            lib.mask_rows(h, m2[t], hp[t], N, H)
            lib.gemm(hp[t], H, 0, w_hh, H, 0, gh, GH, N, GH, H, shift=b_hh)
            if lstm:
                lib.lstm_gates_fwd(gi3[t], gh, c, m2[t], out[t], aux[t], gates[t], N, H)
                c = aux[t]
            else:
                lib.gru_gates_fwd(gi3[t], gh, hp[t], None, out[t], gates[t], aux[t], N, H)
            h = out[t]
        ctx.lstm, ctx.dims = lstm, (T, N, H, GH)
        ctx.save_for_backward(hp, gates, aux, m2, w_hh, _f32c(c0) if lstm else h0)
        return out.view(T * N, H), out[T - 1], (aux[T - 1] if lstm else out[T - 1])

    @staticmethod
    def backward(ctx, dout, dh_fin, dc_fin):
        lib = L()
        hp, gates, aux, m2, w_hh, c0 = ctx.saved_tensors
        T, N, H, GH = ctx.dims
        lstm = ctx.lstm
        dev = hp.device
        if ctx.rollout:
            dgi = torch.empty((T, N, GH), device=dev, dtype=torch.float32)
            dgh = torch.empty_like(dgi)
            dh0 = torch.empty((N, H), device=dev, dtype=torch.float32)
            carry = None if dh_fin is None else _f32c(dh_fin)
            if dc_fin is not None:  # the GRU's second output aliases h_T
                carry = _f32c(dc_fin) if carry is None else carry + dc_fin
            lib.gru_rollout_bwd(None if dout is None else _f32c(dout), carry, gates, aux, hp,
                                m2.contiguous(), w_hh.t().contiguous(), dgi, dgh, dh0,
                                _rollout_workspace(dev, N, H), T, N, H)
            dw = torch.empty_like(w_hh)
            lib.gemm(dgh.view(T * N, GH), GH, 1, hp.view(T * N, H), H, 1, dw, H, GH, H, T * N)
            db = torch.empty((GH,), device=dev, dtype=torch.float32)
            lib.colsum(dgh.view(T * N, GH), GH, T * N, GH, db, 0)
            return None, dgi.view(T * N, GH), dh0, None, None, dw, db
        dout = _f32c(dout).view(T, N, H)
        mf = m2.to(torch.float32).unsqueeze(-1)                        # [T, N, 1]
        dgi = torch.empty((T, N, GH), device=dev, dtype=torch.float32)
        dgh = dgi if lstm else torch.empty_like(dgi)                   # LSTM: same pre-activation
        carry = torch.zeros((N, H), device=dev, dtype=torch.float32) if dh_fin is None \
            else _f32c(dh_fin).clone()
        if not lstm and dc_fin is not None:  # the GRU's second output aliases h_T
            carry = carry + dc_fin
        dc = None
        if lstm:
            dc = torch.zeros((N, H), device=dev, dtype=torch.float32) if dc_fin is None \
                else _f32c(dc_fin).clone()
        dh_t = torch.empty((N, H), device=dev, dtype=torch.float32)
        acc = torch.empty((N, H), device=dev, dtype=torch.float32)
        dc_prev = torch.empty((N, H), device=dev, dtype=torch.float32) if lstm else None
        fused = lib.rnn_step_supported(N, H, lstm)
        w_hh_t = w_hh.t().contiguous() if fused else None
        for t in range(T - 1, -1, -1):
            if fused:  # gate gradients, then carry <- mask * (dh*z + dgh W_hh): two launches
                c_prev = (aux[t - 1] if t > 0 else c0) if lstm else None
                lib.rnn_step_bwd(lstm, dout[t], carry, dc, gates[t], aux[t], hp[t], c_prev, m2[t],
                                 w_hh_t, dgi[t], dgh[t], acc, dc_prev, N, H)
                if lstm:
                    dc, dc_prev = dc_prev, dc
                continue
            torch.add(dout[t], carry, out=dh_t)
            if lstm:
                c_prev = aux[t - 1] if t > 0 else c0
                lib.lstm_gates_bwd(dh_t, dc, gates[t], c_prev, aux[t], m2[t], dgi[t], dc_prev,
                                   N, H)
                dc, dc_prev = dc_prev, dc
                # dh_{t-1} = m_t * (dgates W_hh)
                lib.gemm(dgi[t], GH, 0, w_hh, H, 1, acc, H, N, H, GH)
            else:
                lib.gru_gates_bwd(dh_t, gates[t], aux[t], hp[t], None, dgi[t], dgh[t], acc, N, H)
                # acc = dh_t * z; += dgh W_hh; then the step's mask
                lib.gemm(dgh[t], GH, 0, w_hh, H, 1, acc, H, N, H, GH, accumulate=1)
            torch.mul(acc, mf[t], out=carry)
        dw = torch.empty_like(w_hh)
        lib.gemm(dgh.view(T * N, GH), GH, 1, hp.view(T * N, H), H, 1, dw, H, GH, H, T * N)
        db = torch.empty((GH,), device=dev, dtype=torch.float32)
        lib.colsum(dgh.view(T * N, GH), GH, T * N, GH, db, 0)
        dc0 = dc if lstm else None
        return None, dgi.view(T * N, GH), carry, dc0, None, dw, db


# ----------------------------------------------------------------- packed-sequence RNN
class RNNLayerFn(Function):
    """A whole packed (bi)directional LSTM / GRU layer over time-major input rows as ONE autograd
    node (instruction_encoder.py:27-32,80-94: pack_padded_sequence -> nn.LSTM/GRU ->
    pad_packed_sequence), built for a short HOST path: the backward is two library calls
    (vlnce_rnn_seq_bwd2, vlnce_rnn_seq_wgrad) where the chain linear -> RNNSeqFn -> cat issued
    ~45 launches from Python, eagerly, behind the tail's backward graph.

    forward(kind, save, lengths_i32, dirs, B, L, x_tm [L*B, E], then per direction
            w_ih [G*H,E], b_ih, w_hh [G*H,H], b_hh)
    returns seq [B, L, dirs*H] (zeros past each length: the consumer's row layout, written by
    the kernel itself) then hfin_0[, hfin_1] ([B,H])."""

    @staticmethod
    def forward(ctx, kind, save, lengths, dirs, B, Lm, x_tm, *params):
        x_tm, ldx = _rows2d(_f32c(x_tm) if x_tm.dtype != torch.float32 else x_tm)
        E = x_tm.size(1)
        w_ih = [_f32c(params[4 * d]) for d in range(dirs)]
        b_ih = [params[4 * d + 1] for d in range(dirs)]
        w_hh = [_f32c(params[4 * d + 2]) for d in range(dirs)]
        b_hh = [_f32c(params[4 * d + 3]) for d in range(dirs)]
        GH, H = w_hh[0].shape
        dev = x_tm.device
        lib = L()
        rows = Lm * B
        gis = []
        for d in range(dirs):
            gi = torch.empty((rows, GH), device=dev, dtype=torch.float32)
            if not _planes_gemm(x_tm, ldx, w_ih[d], gi, b_ih[d]):
                lib.gemm(x_tm, ldx, 0, w_ih[d], E, 0, gi, GH, rows, GH, E, shift=b_ih[d])
            gis.append(gi.view(Lm, B, GH))

        def new(*shape):
            return torch.empty(shape, device=dev, dtype=torch.float32)

        out_tm = [new(Lm, B, H) for _ in range(dirs)]
        seq = new(B, Lm, dirs * H)
        hfin = [new(B, H) for _ in range(dirs)]
        gates = aux = None
        if save:
            gates = [new(Lm, B, GH) for _ in range(dirs)]
            aux = [new(Lm, B, H) for _ in range(dirs)]
        lib.rnn_seq_fwd2(kind, dirs, gis, w_hh, b_hh, lengths, out_tm, seq, dirs * H, Lm * dirs * H,
                         hfin, gates, aux, B, Lm, H)
        ctx.cfg = (kind, dirs, Lm, B, H, GH, E, ldx)
        ctx.set_materialize_grads(False)   # unused outputs (the final states) arrive as None, not zeros
        if save:
            ctx.save_for_backward(lengths, x_tm, *w_ih, *w_hh, *out_tm, *gates, *aux)
        return (seq, *hfin)

    @staticmethod
    def backward(ctx, dseq, *dhf):
        kind, dirs, Lm, B, H, GH, E, ldx = ctx.cfg
        sv = ctx.saved_tensors
        lengths, x_tm = sv[0], sv[1]
        w_ih, w_hh = sv[2:2 + dirs], sv[2 + dirs:2 + 2 * dirs]
        out_tm = sv[2 + 2 * dirs:2 + 3 * dirs]
        gates = sv[2 + 3 * dirs:2 + 4 * dirs]
        aux = sv[2 + 4 * dirs:2 + 5 * dirs]
        dev = x_tm.device
        st = sb = 0
        if dseq is not None:
            #  an EXPANDED upstream gradient -- seq.mean(dim=1), a broadcast add -- has a
            # unit inner stride and a zero batch / time stride, which vlnce_rnn_seq_bwd2 rejects
            if (dseq.dtype != torch.float32 or dseq.stride(2) != 1 or dseq.stride(0) <= 0
                    or dseq.stride(1) <= 0):
                dseq = _f32c(dseq)
            sb, st = dseq.stride(0), dseq.stride(1)
        dhf = [(_f32c(g) if g is not None else None) for g in dhf[:dirs]]

        def new(*shape):
            return torch.empty(shape, device=dev, dtype=torch.float32)

        dgi = [new(Lm, B, GH) for _ in range(dirs)]
        dgh = [new(Lm, B, GH) for _ in range(dirs)] if kind == 1 else None
        lib = L()
        ws = new(dirs, Lm, B, H) if dseq is not None else None
        lib.rnn_seq_bwd2(kind, dirs, list(w_hh), lengths, list(out_tm), list(gates), list(aux), dseq,
                         st, sb, ws, dhf, dgi, dgh, B, Lm, H)
        dw_ih = [new(GH, E) for _ in range(dirs)]
        dw_hh = [new(GH, H) for _ in range(dirs)]
        db_ih = [new(GH) for _ in range(dirs)]
        db_hh = [new(GH) for _ in range(dirs)]
        need_dx = ctx.needs_input_grad[6]
        dx = new(Lm * B, E) if need_dx else None
        from .streams import BranchStreams

        if (dirs == 2 and Lm * B >= 2048 and BranchStreams.enabled(dev)
                and os.environ.get("VLNCE_RNN_WGRAD_STREAMS", "1") != "0"):
            # the two directions' parameter gradients (10 launches each, ~80 us at 64 x 80 rows) on
            # two streams: this is the tail end of a step's backward and nothing else is running
            # (9.58 -> 9.47 ms/step, profiles/r05_g_*).  Not for the few distinct instructions of a
            # sequence-mode batch (5 x 200 rows): there the second stream's launches run beside the
            # state encoders' one-launch rollout backward and the update got 1 ms SLOWER
            # (7.97 -> 8.9 ms, profiles/r05_i_same_box_round4_vs_round5.txt)
            cur = torch.cuda.current_stream(dev)
            side = BranchStreams()._stream(1, dev)
            if side is cur:
                side = BranchStreams()._stream(2, dev)
            dx1 = new(Lm * B, E) if need_dx else None
            fork = torch.cuda.Event()
            fork.record(cur)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                lib.rnn_seq_wgrad(kind, 1, dgi[1:], dgh[1:] if dgh else None, list(out_tm[1:]), x_tm, ldx, E,
                                  list(w_ih[1:]), dw_ih[1:], dw_hh[1:], db_ih[1:], db_hh[1:], dx1, B, Lm,
                                  H, first_dir=1)
                done = torch.cuda.Event()
                done.record(side)
            lib.rnn_seq_wgrad(kind, 1, dgi[:1], dgh[:1] if dgh else None, list(out_tm[:1]), x_tm, ldx, E,
                              list(w_ih[:1]), dw_ih[:1], dw_hh[:1], db_ih[:1], db_hh[:1], dx, B, Lm, H)
            cur.wait_event(done)
            if need_dx:
                dx.add_(dx1)
        else:
            lib.rnn_seq_wgrad(kind, dirs, dgi, dgh, list(out_tm), x_tm, ldx, E, list(w_ih), dw_ih, dw_hh,
                              db_ih, db_hh, dx, B, Lm, H)
        res = [None, None, None, None, None, None, dx]
        for d in range(dirs):
            res += [dw_ih[d], db_ih[d], dw_hh[d], db_hh[d]]
        return tuple(res)


def rnn_layer(kind, lengths_i32, x_tm, B, Lm, per_direction, need_grad):
    """per_direction: list of (w_ih, b_ih, w_hh, b_hh).  Returns (seq [B, L, dirs*H], [hfin_d])."""
    flat = [t for quad in per_direction for t in quad]
    res = RNNLayerFn.apply(kind, bool(need_grad), lengths_i32, len(per_direction), B, Lm, x_tm, *flat)
    return res[0], list(res[1:])
