"""Autograd for the visual trunks when MODEL.{RGB,DEPTH}_ENCODER.trainable=True.

The frozen (default) trunks run a fused forward-only plan (resnet_encoders.py).
With trainable encoders the trunk runs layer by layer, keeps the raw conv
outputs and the normalised activations, and this module's `TrunkFn` replays the
layers in reverse on the hand-written backward kernels:

  conv  : data gradient  = vlnce_conv2d_fwd with the flipped/transposed weights
                           (stride-2 3x3: zero-inserted dY; stride-2 1x1: strided scatter)
          weight gradient = vlnce_conv2d_wgrad (dY^T x im2col(X), split-K over pixels)
  BatchNorm / GroupNorm (+ReLU, +residual) : vlnce_bn_bwd / vlnce_gn_bwd
  max-pool (arg-max tap saved in forward), adaptive average pool.
"""
import os

import torch
from torch.autograd import Function

from .. import ops


def L():
    return ops.L()


_SEPARATE_ADD = os.environ.get("VLNCE_DGRAD_ADD", "1") == "0"


# ------------------------------------------------------------------ layer primitives
def conv_raw(x, w_ohwi, stride, pad):
    return ops.conv2d_nhwc(x, w_ohwi, stride, pad)


def conv_backward(x, w_ohwi, dy, stride, pad, need_dx, add=None, pow2=None):
    """returns (dx | None, dW in OIHW).  `add` (the gradient arriving at the same tensor over the
    block's other branch) is summed into dx by the data-gradient convolution's epilogue instead of
    a separate pass over two block-input-sized tensors.  `pow2` ([2, P >= max(Cin, Cout)], from
    bn_backward / gn_backward): the power of two at which dy fits the fp16 planes -- with it the
    data gradient runs in plane format 2 (three plane products), dy scaled in the prologue and the
    result scaled back in the epilogue, both exactly."""
    lib = L()
    g = ops.conv_geometry(x, w_ohwi, stride, pad)
    Cout, KH, KW, Cin = w_ohwi.shape
    dy = dy.contiguous()
    # (one zeroed arena for all of a trunk's dW + `accumulate`, instead of a fill in front of every
    # split launch: measured equal, 30.6-31.6 vs 30.6-30.7 ms, and dropped -- profiles/r06_o_*)
    dw = torch.empty_like(w_ohwi)
    lib.conv2d_wgrad(x, dy, dw, g, pow2)
    dw_oihw = dw.permute(0, 3, 1, 2).contiguous()
    if not need_dx:
        return None, dw_oihw
    N, H, W, _ = x.shape
    # flipped taps, in/out channels swapped: [Cin, KH, KW, Cout].  Gradient operands: three bf16
    # planes (fp32's exponent range); the fp16 planes of the forward would flush them
    # (a trainable trunk's one-launch weight preparation leaves that bank, with its planes and
    # fragments, on the forward tensor: _WeightCache.prepare_trainable)
    wt = getattr(w_ohwi, "_vlnce_dgrad", None)
    if wt is None:
        wt = w_ohwi.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    if add is not None:
        if _SEPARATE_ADD:   # A/B knob (VLNCE_DGRAD_ADD=0): the sum as its own pass
            dx, dw_oihw = conv_backward(x, w_ohwi, dy, stride, pad, True, None, pow2)
            return dx + add, dw_oihw
        add = add.contiguous()
    if pow2 is not None:
        zero = ops.zeros_vec(x.device, max(Cin, Cout))
        how = dict(in_scale=pow2[0, :Cout], in_shift=zero[:Cout], scale=pow2[1, :Cin],
                   shift=zero[:Cin], w_format=ops.PLANES_F16X3)
    else:
        how = dict(w_format=ops.PLANES_BF16X6)
    if stride == 1:
        dx = ops.conv2d_nhwc(dy, wt, 1, KH - 1 - pad, residual=add, **how)
    elif KH == 1:
        # 1x1 / stride s: only the sampled pixels receive gradient
        small = ops.conv2d_nhwc(dy, wt, 1, 0, **how)
        if add is not None:
            dx = add.clone()
            dx[:, ::stride, ::stride] += small
        else:
            dx = torch.zeros((N, H, W, Cin), device=x.device, dtype=torch.float32)
            dx[:, ::stride, ::stride] = small
    else:
        # general stride: dY zero-inserted on the stride grid, then a stride-1 correlation
        # with the flipped taps and padding KH-1-pad lands exactly on the input pixels
        Hu, Wu = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
        up = torch.zeros((N, Hu, Wu, Cout), device=x.device, dtype=torch.float32)
        up[:, ::stride, ::stride] = dy
        dx = ops.conv2d_nhwc(up, wt, 1, KH - 1 - pad, residual=add, **how)
    return dx, dw_oihw


class BNSaved:
    __slots__ = ("raw", "y", "mean", "rstd", "gamma", "relu", "batch_stats", "has_res")


def bn_forward(raw, bn, relu, residual, stats, touched):
    """normalise+activate the raw conv output, keeping what backward needs."""
    lib = L()
    Cc = raw.size(-1)
    M = raw.numel() // Cc
    dev = raw.device
    sv = BNSaved()
    sv.raw, sv.relu, sv.has_res, sv.gamma = raw, relu, residual is not None, bn.weight
    scale = torch.empty(Cc, device=dev, dtype=torch.float32)
    shift = torch.empty_like(scale)
    sv.mean = torch.empty_like(scale)
    sv.rstd = torch.empty_like(scale)
    if bn.training:
        partial, tiles_m, tile_rows = stats
        wb = lib.bn_finalize_workspace_bytes(tiles_m, Cc)
        ws = torch.empty(wb // 8, device=dev, dtype=torch.float64) if wb else None
        lib.bn_finalize(partial, tiles_m, tile_rows, M, Cc, bn.weight, bn.bias, float(bn.eps),
                        float(bn.momentum), bn.running_mean, bn.running_var, scale, shift,
                        sv.mean, sv.rstd, workspace=ws)
        touched.append(bn.num_batches_tracked)
        sv.batch_stats = True
        # the reference's arithmetic: (x - mean) * (gamma*rstd) + beta
        sv.y = ops.scale_shift_act(raw, scale, bn.bias.detach(), center=sv.mean,
                                   residual=residual,
                                   act=ops.ACT_RELU if relu else ops.ACT_NONE)
        return sv.y, sv
    else:
        sv.mean.copy_(bn.running_mean)
        sv.rstd.copy_(torch.rsqrt(bn.running_var + bn.eps))
        scale = (bn.weight.detach() * sv.rstd).contiguous()
        shift = (bn.bias.detach() - sv.mean * scale).contiguous()
        sv.batch_stats = False
    sv.y = ops.scale_shift_act(raw, scale, shift, residual=residual,
                               act=ops.ACT_RELU if relu else ops.ACT_NONE)
    return sv.y, sv


def _pow2_buffer(raw, P):
    """[2, P] for the norm-backward kernels' power-of-two output, or None where the backward
    convolutions keep format 1 (VLNCE_GRAD_PLANES=bf16, the CPU simulator's tensors, C % 4 != 0)"""
    if (not P or ops.GRAD_PLANES != ops.PLANES_F16X3 or raw.size(-1) % 4 != 0
            or L().plane_format() != ops.PLANES_F16X3):
        return None
    return torch.empty((2, P), device=raw.device, dtype=torch.float32)


def bn_backward(dy, sv, P=0):
    """returns (d raw, d residual | None, dgamma, dbeta, pow2 | None); P = the longer channel count
    of the convolution whose backward reads d raw (0: no power of two wanted)."""
    Cc = sv.raw.size(-1)
    M = sv.raw.numel() // Cc
    dev = sv.raw.device
    dx = torch.empty_like(sv.raw)
    dres = torch.empty_like(sv.raw) if sv.has_res else None
    dg = torch.empty(Cc, device=dev, dtype=torch.float32)
    db = torch.empty_like(dg)
    lib = L()
    ws = torch.empty(max(lib.bn_bwd_workspace_floats(M, Cc), 1), device=dev, dtype=torch.float32)
    pow2 = _pow2_buffer(sv.raw, P)
    lib.bn_bwd(dy.contiguous(), sv.y, sv.raw, sv.mean, sv.rstd, sv.gamma.detach(), M, Cc, sv.relu,
               sv.batch_stats, dx, dres, dg, db, ws, pow2)
    return dx, dres, dg, db, pow2


class GNSaved:
    __slots__ = ("raw", "y", "mean", "rstd", "gamma", "groups", "relu", "has_res")


def gn_forward(raw, gn, relu, residual):
    lib = L()
    N, H, W, Cc = raw.shape
    HW = H * W
    dev = raw.device
    sv = GNSaved()
    sv.raw, sv.relu, sv.has_res, sv.gamma, sv.groups = raw, relu, residual is not None, gn.weight, \
        gn.num_groups
    chunks = lib.gn_chunks(HW)
    partial = torch.empty((N, chunks, Cc, 2), device=dev, dtype=torch.float32)
    lib.gn_partial(raw, N, HW, Cc, partial)
    scale = torch.empty((N, Cc), device=dev, dtype=torch.float32)
    shift = torch.empty_like(scale)
    center = torch.empty_like(scale)
    sv.mean = torch.empty((N, gn.num_groups), device=dev, dtype=torch.float32)
    sv.rstd = torch.empty_like(sv.mean)
    lib.gn_finalize(partial, N, HW, Cc, gn.num_groups, gn.weight, gn.bias, float(gn.eps), scale,
                    shift, sv.mean, sv.rstd, center_out=center)
    sv.y = ops.scale_shift_act(raw, scale, shift, center=center, rows_per_sample=HW,
                               residual=residual, act=ops.ACT_RELU if relu else ops.ACT_NONE)
    return sv.y, sv


def gn_backward(dy, sv, P=0):
    lib = L()
    N, H, W, Cc = sv.raw.shape
    dev = sv.raw.device
    dx = torch.empty_like(sv.raw)
    dres = torch.empty_like(sv.raw) if sv.has_res else None
    dg = torch.empty(Cc, device=dev, dtype=torch.float32)
    db = torch.empty_like(dg)
    ws = torch.empty(lib.gn_bwd_workspace_floats(N, H * W, Cc, sv.groups), device=dev,
                     dtype=torch.float32)
    pow2 = _pow2_buffer(sv.raw, P)
    lib.gn_bwd(dy.contiguous(), sv.y, sv.raw, sv.mean, sv.rstd, sv.gamma.detach(), N, H * W, Cc,
               sv.groups, sv.relu, dx, dres, dg, db, ws, pow2)
    return dx, dres, dg, db, pow2


def maxpool_forward(x):
    N, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    arg = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=torch.uint8)
    L().maxpool3x3s2_argmax(x, y, arg, N, H, W, Cc, Ho, Wo)
    return y, (arg, x.shape)


def maxpool_backward(dy, saved):
    arg, (N, H, W, Cc) = saved
    dx = torch.empty((N, H, W, Cc), device=dy.device, dtype=torch.float32)
    L().maxpool3x3s2_bwd(dy.contiguous(), arg, dx, N, H, W, Cc, dy.size(1), dy.size(2))
    return dx


def avgpool_backward(dy, in_shape, out_hw):
    N, H, W, Cc = in_shape
    dx = torch.empty(in_shape, device=dy.device, dtype=torch.float32)
    L().adaptive_avgpool_bwd(dy.contiguous(), dx, N, H, W, Cc, out_hw[0], out_hw[1])
    return dx


# ------------------------------------------------------------------ the trunk Function
class TrunkFn(Function):
    """forward(plan, x, *params): plan.run(x) -> (out, tape); backward replays the tape.
    `plan` is the owning trunk module (it knows its layer structure); `params` are its
    trainable parameters in `plan.trainable_params()` order so autograd routes the
    returned gradients to them."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        out, tape = plan.run_recording(x)
        ctx.plan = plan
        ctx.tape = tape
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = ctx.plan.backward_from_tape(ctx.tape, dout.contiguous())
        ctx.tape = None
        plist = ctx.plan.trainable_params()
        return (None, None) + tuple(grads.get(id(p)) for p in plist)
