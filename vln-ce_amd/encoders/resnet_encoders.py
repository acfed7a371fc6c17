"""RGB / depth visual encoders of the VLN-CE policies on the HIP kernels.

Mirrors vlnce_baselines/models/encoders/resnet_encoders.py:17-229 (module and
parameter names, constructor arguments, cached-feature bypass, output shapes)
plus the two third-party trunks it instantiates (torchvision resnet18/50 and
habitat-lab v0.1.7 ResNetEncoder).  nn.Conv2d / nn.BatchNorm2d
/ nn.GroupNorm objects are kept ONLY as parameter containers so state_dict keys
match published checkpoints; their torch forward is never called.  The trunk
runs channels-last on libvlnce_hip.so: implicit-GEMM fp32-MFMA convolutions
with the norm / ReLU / residual fused into the epilogue (eval BatchNorm) or a
statistics epilogue + one apply pass (train BatchNorm, GroupNorm).

Visual feature maps are returned as *logical* NCHW tensors whose memory is
NHWC (a permuted view), so forward hooks on `.cnn` / `.visual_encoder`
(dagger_trainer.py:300-314) see the reference's shapes while downstream HIP
ops get channels-last rows without a copy.
"""
import os
import warnings
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..config import Box, Dict
from ..streams import DropsGraphsOnApply, capture_guard, wait_ready
from . import trunk_backward as tb


# ------------------------------------------------------------------ helpers
class _WeightCache:
    """OIHW nn.Conv2d weights repacked to the kernels' OHWI layout, refreshed
    when the parameter is updated in place (optimizer step / load_state_dict)
    or moved to another device."""

    def __init__(self):
        self._packed = {}
        self._folded = {}
        self._prep = None

    @staticmethod
    def _key(*ts):
        return tuple((t.data_ptr(), t._version, t.device) for t in ts)

    def conv(self, conv):
        k = self._key(conv.weight)
        hit = self._packed.get(id(conv))
        if hit is None or hit[0] != k:
            w = conv.weight.detach().permute(0, 2, 3, 1).contiguous()
            if not conv.weight.requires_grad:   # a frozen bank outside the fp16 planes' range -> bf16 planes
                ops.check_weight_range(w)
            hit = (k, w)
            self._packed[id(conv)] = hit
        return hit[1]

    def prepare_trainable(self, convs):
        """Trainable trunks: every weight image the step needs -- OHWI fp32, planes and fragments in
        the forward format, and the data-gradient bank (taps reversed, channels swapped) as fp32,
        planes and fragments in the backward format (ops.GRAD_PLANES) -- written by ONE launch (vlnce_conv2d_prepare_weights)
        into persistent buffers, and handed to conv() / ops.split_weights / ops.pack_weights /
        trunk_backward.conv_backward through the caches those already consult.  Convolutions the
        kernel does not take (the stems: Cin = 1 or 3) keep the per-tensor path."""
        convs = [c for c in convs if c.weight.is_cuda and c.weight.is_contiguous()
                 and c.weight.size(0) % 32 == 0 and c.weight.size(1) % 32 == 0]
        if not convs or os.environ.get("VLNCE_WEIGHT_PREP", "1") == "0":
            return
        lib = ops.L()
        fmt = ops.plane_format()
        gfmt = ops.GRAD_PLANES if fmt == ops.PLANES_F16X3 else ops.PLANES_BF16X6
        sig = (fmt, gfmt) + tuple((id(c), c.weight.data_ptr()) for c in convs)
        st = self._prep
        if st is None or st["sig"] != sig:
            st = dict(sig=sig, key=None, rows=[], plan=None)
            jobs = []
            for c in convs:
                p = c.weight
                Cout, Cin, KH, KW = p.shape
                dev = p.device

                def i16(*shape):
                    return torch.empty(shape, device=dev, dtype=torch.int16)
                row = dict(conv=c,
                           w=(torch.empty((Cout, KH, KW, Cin), device=dev) if KH * KW > 1 else None),
                           split=i16(3, p.numel()), frag=i16(p.numel() * 3),
                           wt=torch.empty((Cin, KH, KW, Cout), device=dev),
                           split_t=i16(3, p.numel()), frag_t=i16(p.numel() * 3))
                row["split"]._vlnce_fmt = row["frag"]._vlnce_fmt = fmt
                row["split_t"]._vlnce_fmt = row["frag_t"]._vlnce_fmt = gfmt
                if row["w"] is not None:
                    jobs.append((p, row["w"], lib.WP_F32, 0, 0))
                jobs += [(p, row["split"], lib.WP_PLANES, 0, fmt),
                         (p, row["frag"], lib.WP_FRAGMENTS, 0, fmt),
                         (p, row["wt"], lib.WP_F32, 1, 0),
                         (p, row["split_t"], lib.WP_PLANES, 1, gfmt),
                         (p, row["frag_t"], lib.WP_FRAGMENTS, 1, gfmt)]
                st["rows"].append(row)
            st["plan"] = lib.weight_prep_plan(jobs)
            self._prep = st
        key = tuple(c.weight._version for c in convs)
        if st["key"] == key:
            return
        lib.conv2d_prepare_weights(st["plan"])
        st["key"] = key
        for row in st["rows"]:
            c = row["conv"]
            w = row["w"] if row["w"] is not None else c.weight.detach().permute(0, 2, 3, 1)
            wt = row["wt"]
            w.__dict__["_vlnce_split"] = {fmt: (w._version, row["split"])}
            w.__dict__["_vlnce_frag"] = {fmt: (w._version, row["frag"])}
            wt.__dict__["_vlnce_split"] = {gfmt: (wt._version, row["split_t"])}
            wt.__dict__["_vlnce_frag"] = {gfmt: (wt._version, row["frag_t"])}
            w._vlnce_dgrad = wt
            self._packed[id(c)] = (self._key(c.weight), w)

    def stem_s2d(self, conv):
        """7x7/s2/p3 stem taps regrouped for the space-to-depth formulation (ops.stem_weight_s2d)."""
        k = self._key(conv.weight)
        hit = self._packed.get(("s2d", id(conv)))
        if hit is None or hit[0] != k:
            hit = (k, ops.stem_weight_s2d(self.conv(conv)))
            self._packed[("s2d", id(conv))] = hit
        return hit[1]

    def stem7(self, conv):
        """the 7x7 stem filters as the B fragments of vlnce_stem7_fwd (ops.stem7_pack_weights) in
        the launch's plane format; returns (fragments, format)"""
        fmt = ops.plane_format(None, self.conv(conv))
        k = self._key(conv.weight) + (fmt,)
        hit = self._packed.get(("stem7", id(conv)))
        if hit is None or hit[0] != k:
            hit = (k, ops.stem7_pack_weights(self.conv(conv), fmt))
            self._packed[("stem7", id(conv))] = hit
        return hit[1], fmt

    def bn_eval(self, bn, gen=0):
        # `gen` counts train-mode forwards: the HIP kernels update the running statistics
        # through raw pointers, which torch's version counters do not see
        k = self._key(bn.weight, bn.bias, bn.running_mean, bn.running_var) + (gen,)
        hit = self._folded.get(id(bn))
        if hit is None or hit[0] != k:
            scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias.detach() - bn.running_mean * scale
            hit = (k, scale.contiguous(), shift.contiguous())
            self._folded[id(bn)] = hit
        return hit[1], hit[2]


class _GraphRunner:
    """Replays a frozen trunk forward as a captured HIP graph (hipGraph through
    torch.cuda.CUDAGraph).  A trunk is ~110-210 short kernel launches whose
    host-side issue cost exceeds the GPU time of the small layers; the graph
    removes it.  Protocol per key (input shape + norm modes + parameter
    versions): 1st call runs eagerly (real forward, warms kernel attributes and
    the weight caches), 2nd call captures and replays, later calls replay.
    BatchNorm running-stat side effects happen exactly once per call either way.
    Set VLNCE_HIP_GRAPHS=0 to disable."""

    MAX_GRAPHS = 6

    def __init__(self, fn):
        self.fn = fn
        self.entries = OrderedDict()

    @staticmethod
    def enabled():
        return os.environ.get("VLNCE_HIP_GRAPHS", "1") != "0"

    def captured(self, key):
        return not self.enabled() or isinstance(self.entries.get(key), list)

    @staticmethod
    def _parts(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x,)

    def __call__(self, x, key):
        """x: a tensor or a tuple of tensors / None (frame stack, extra frame, mask)."""
        parts = self._parts(x)
        probe = parts[0]
        if (not self.enabled() or not probe.is_cuda or torch.cuda.is_current_stream_capturing()):
            return self.fn(x)
        ent = self.entries.get(key)
        if ent is None:
            while len(self.entries) >= self.MAX_GRAPHS:
                self.entries.popitem(last=False)
            self.entries[key] = "seen"
            return self.fn(x)
        self.entries.move_to_end(key)
        cur = torch.cuda.current_stream(probe.device)
        if ent == "seen":
            # static inputs keep the STORAGE dtype (uint8 frames stay uint8); clone() densifies
            # a centre-crop view, so inside the graph the crop is part of this copy
            static_in = tuple(None if t is None else t.clone() for t in parts)
            graph = torch.cuda.CUDAGraph()
            with capture_guard(), torch.cuda.graph(graph):
                static_out = self.fn(static_in if isinstance(x, (tuple, list)) else static_in[0])
            ent = [graph, static_in, static_out, None]
            self.entries[key] = ent
        else:
            # one graph = one set of static buffers: a replay issued from another stream (the
            # run-ahead path next to an inline call) must wait for the previous one to be
            # done with them
            if ent[3] is not None:
                cur.wait_event(ent[3])
            for dst, src in zip(ent[1], parts):
                if dst is not None:
                    dst.copy_(src)
        ent[0].replay()
        out = ent[2].clone()
        ent[3] = torch.cuda.Event()
        ent[3].record(cur)
        return out


def _stem_is_s2d(conv, H, W=None):
    """the frozen trunks run a 7x7/stride-2/pad-3 stem as a 4x4 convolution over 2x2
    space-to-depth blocks (contiguous 16-byte operand loads instead of a 3- or 1-channel gather).
    `H` may be a [N,H,W,C] tensor."""
    if W is None:
        H, W = H.size(1), H.size(2)
    return (conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and H % 2 == 0 and W % 2 == 0 and os.environ.get("VLNCE_STEM_S2D", "1") != "0")


def _as_nhwc(t_nchw_logical):
    """logical NCHW tensor -> contiguous NHWC tensor (no copy when already channels-last)."""
    return t_nchw_logical.permute(0, 2, 3, 1).contiguous()


def _grid_embedding_nhwc(emb, b, h, w):
    # reference: embeddings(arange(n)).view(1, -1, h, w).expand(b, ...) -- a raw
    # reinterpretation of the [h*w, 64] table as [64, h, w] (resnet_encoders.py:97-111,201-215)
    e = emb.weight.view(1, emb.embedding_dim, h, w).permute(0, 2, 3, 1)
    return e.expand(b, h, w, emb.embedding_dim)


# ------------------------------------------------------------------ torchvision trunk
def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


def _c1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _c3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def stages(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _c3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _c1(planes, planes * 4)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def stages(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]


class _GlobalAvgPool(nn.Module):
    out_hw = (1, 1)


class SpatialAvgPool(nn.Module):
    out_hw = (4, 4)


class HipResNetTrunk(DropsGraphsOnApply, nn.Sequential):
    """torchvision ResNet children[:-1] (conv1, bn1, relu, maxpool, layer1..4,
    [avgpool]) with indices/keys preserved; forward = fused HIP plan."""

    def __init__(self, block, layers):
        inplanes = 64
        mods = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
                nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1)]
        for planes, n, stride in zip((64, 128, 256, 512), layers, (1, 2, 2, 2)):
            down = None
            if stride != 1 or inplanes != planes * block.expansion:
                down = nn.Sequential(_c1(inplanes, planes * block.expansion, stride),
                                     nn.BatchNorm2d(planes * block.expansion))
            blocks = [block(inplanes, planes, stride, down)]
            inplanes = planes * block.expansion
            blocks += [block(inplanes, planes) for _ in range(1, n)]
            mods.append(nn.Sequential(*blocks))
        mods.append(_GlobalAvgPool())
        super().__init__(*mods)
        self.final_channels = inplanes
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._cache = _WeightCache()
        self.input_scale = None  # set by the owning encoder: per-channel (scale, shift)
        self._bn_gen = 0
        self._graphs = _GraphRunner(self._forward_impl)
        self._norms = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]

    # ---- eval mode: BatchNorm folded to a per-channel scale/shift in the conv epilogue
    def _conv_bn_eval(self, x, conv, bn, relu, residual=None, prologue=None, s2d=False):
        pro = {} if prologue is None else dict(in_scale=prologue[0], in_shift=prologue[1])
        scale, shift = self._cache.bn_eval(bn, self._bn_gen)
        w, stride, pad = ((self._cache.stem_s2d(conv), 1, 0) if s2d else
                          (self._cache.conv(conv), conv.stride[0], conv.padding[0]))
        return ops.conv2d_nhwc(x, w, stride, pad, scale=scale, shift=shift, residual=residual,
                               act=ops.ACT_RELU if relu else ops.ACT_NONE, **pro)

    # ---- train mode: conv writes the RAW output + per-tile moments; the finalize kernel
    # turns them into (scale, shift) and updates the running statistics.  The pending
    # normalisation (+ReLU) is then applied by whoever consumes the raw tensor: the next
    # conv's operand loader, the max-pool, or the block-end add pass.
    def _conv_stats(self, x, conv, bn, touched, prologue=None, in_relu=False, s2d=False,
                    dual=None):
        pro = {}
        if prologue is not None:
            pro = dict(in_scale=prologue[0], in_shift=prologue[1], in_relu=in_relu,
                       in_center=prologue[2] if len(prologue) > 2 else None)
        if dual is not None:  # (x2, pending norm of x2 | None, side_out)
            x2, p2, side = dual
            pro.update(x2=x2, side_out=side)
            if p2 is not None:
                pro.update(in2_scale=p2[0], in2_shift=p2[1], in2_center=p2[2])
        w, stride, pad = ((self._cache.stem_s2d(conv), 1, 0) if s2d else
                          (self._cache.conv(conv), conv.stride[0], conv.padding[0]))
        assert bn.momentum is not None
        if os.environ.get("VLNCE_BN_FUSED", "1") == "0":   # A/B: tile moments + finalize launch(es)
            y, stats = ops.conv2d_nhwc(x, w, stride, pad, want_stats=True, **pro)
            pend = ops.bn_finalize(stats, y.numel() // y.size(-1), bn.weight, bn.bias, bn.eps,
                                   bn.momentum, bn.running_mean, bn.running_var)
            touched.append(bn.num_batches_tracked)
            return y, pend
        # raw output + batch statistics from the convolution itself (vlnce_bn_sums: fp64 atomic
        # column sums) + a one-workgroup finalize behind it, instead of tile moments + finalize
        y, pend = ops.conv2d_bn_train(x, w, stride, pad, bn, **pro)
        touched.append(bn.num_batches_tracked)
        return y, pend  # (scale, shift, center)

    # A block's output relu(bn3(raw3) + skip) is kept PENDING as (raw3, pend3, skip, pend_skip)
    # and evaluated inside the next block's first 1x1 convolution (dual-input operand loader,
    # which also writes the materialised value once for the skip path): the separate
    # read-2-write-1 pass per block disappears.  Blocks that cannot take it (BasicBlock's 3x3
    # first conv, Cin not a multiple of 32) and the trunk's last block use _materialise.
    @staticmethod
    def _materialise(pending):
        raw, pend, skip, pskip = pending
        if pskip is not None:
            return ops.scale_shift_add_act(raw, pend[0], pend[1], skip, pskip[0], pskip[1],
                                           act=ops.ACT_RELU, out=raw, c1=pend[2], c2=pskip[2])
        return ops.scale_shift_act(raw, pend[0], pend[1], center=pend[2], residual=skip,
                                   act=ops.ACT_RELU, out=raw)

    @staticmethod
    def _takes_pending(blk, pending):
        conv = blk.stages()[0][0]
        return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
                and conv.in_channels % 32 == 0 and pending[0].numel() * 4 < (1 << 31)
                and os.environ.get("VLNCE_FUSE_BLOCK_END", "1") != "0")

    def _block_train(self, x, pending, blk, touched):
        """x: materialised block input or None when `pending` holds it; returns the new pending."""
        st = blk.stages()
        if pending is not None and self._takes_pending(blk, pending):
            raw3, pend3, skip, pskip = pending
            x = torch.empty_like(raw3)
            raw, pend = self._conv_stats(raw3, st[0][0], st[0][1], touched, prologue=pend3,
                                         in_relu=True, dual=(skip, pskip, x))
        else:
            if pending is not None:
                x = self._materialise(pending)
            raw, pend = self._conv_stats(x, st[0][0], st[0][1], touched)
        for conv, bn in st[1:]:
            raw, pend = self._conv_stats(raw, conv, bn, touched, prologue=pend, in_relu=True)
        if blk.downsample is not None:
            rd, pd = self._conv_stats(x, blk.downsample[0], blk.downsample[1], touched)
            return raw, pend, rd, pd
        return raw, pend, x, None

    def _block_eval(self, x, blk):
        identity = x
        if blk.downsample is not None:
            identity = self._conv_bn_eval(x, blk.downsample[0], blk.downsample[1], False)
        st = blk.stages()
        for conv, bn in st[:-1]:
            x = self._conv_bn_eval(x, conv, bn, True)
        return self._conv_bn_eval(x, st[-1][0], st[-1][1], True, residual=identity)

    def forward(self, x_nhwc_raw):
        """x: [B,H,W,3] pixel values 0..255 -- uint8 as the simulator delivers them or fp32 as
        habitat's batch_obs casts them -- channels-last, possibly a centre-crop view; or the
        tuple (frames [B,F,H,W,3], extra frame [B,H,W,3], mask [B]) of ops.frames.  Returns
        logical NCHW features."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._plist()):
            # trainable encoder: layer-by-layer forward that records what backward needs
            x = ops.frames_f32(ops.frames(x_nhwc_raw))
            return tb.TrunkFn.apply(self, x, *self.trainable_params()).permute(0, 3, 1, 2)
        key, train = self._graph_key(ops.frames_signature(x_nhwc_raw))
        y = self._graphs(x_nhwc_raw, key)
        if train:
            self._bn_gen += 1
        return y.permute(0, 3, 1, 2)

    def _graph_key(self, signature):
        modes = tuple(m.training for m in self._norms)
        if any(modes) and not all(modes):
            raise NotImplementedError("mixed train/eval BatchNorm modes inside one trunk")
        key = (signature, modes[0], tuple(p._version for p in self._plist(checked=False)),
               id(self.input_scale[0]), len(self._modules),
               0 if modes[0] else self._bn_gen, ops.plane_format())
        return key, modes[0]

    def graph_ready(self, x):
        """True when a forward with this input would only replay a captured graph."""
        return self._graphs.captured(self._graph_key(ops.frames_signature(x))[0])

    def _forward_impl(self, x):
        with torch.no_grad():
            kids = list(self.children())
            train = kids[1].training
            touched = []
            fr = ops.frames(x)  # crop window / frame stack / uint8 -> read by the ingest kernel
            s2d = _stem_is_s2d(kids[0], fr["H"], fr["W"])
            pro = self.input_scale
            stem7 = (s2d and fr["C"] == 3 and kids[0].out_channels in (32, 64)
                     and os.environ.get("VLNCE_STEM7", "1") != "0")
            if stem7:
                # the 7x7 / stride-2 stem straight from the frames on the bf16 matrix pipe
                # (vlnce_stem7_fwd): no regrouped copy of the frames, no fp32-MFMA convolution
                wf, wfmt = self._cache.stem7(kids[0])
                if train:
                    bn = kids[1]
                    acc = ops._bn_state(bn)
                    raw = ops.stem7(fr, wf, kids[0].out_channels, pro[0], pro[1], bn_acc=acc,
                                    w_format=wfmt)
                    pend = ops.bn_finalize_sums(acc, raw.numel() // raw.size(-1), bn)
                    touched.append(bn.num_batches_tracked)
                    x = ops.maxpool3x3s2(raw, pend[0], pend[1], in_relu=True, in_center=pend[2])
                else:
                    sc, sh = self._cache.bn_eval(kids[1], self._bn_gen)
                    x = ops.stem7(fr, wf, kids[0].out_channels, pro[0], pro[1], scale=sc, shift=sh,
                                  act=ops.ACT_RELU, w_format=wfmt)
                    x = ops.maxpool3x3s2(x)
            else:
                if s2d:  # /255 (+mean/std) applied while regrouping, before the zero border
                    x = ops.frames_s2d(fr, 2, 1, self.input_scale[0], self.input_scale[1])
                    pro = None
                else:
                    x = ops.frames_f32(fr)
                if train:
                    raw, pend = self._conv_stats(x, kids[0], kids[1], touched, prologue=pro, s2d=s2d)
                    x = ops.maxpool3x3s2(raw, pend[0], pend[1], in_relu=True, in_center=pend[2])
                else:
                    x = self._conv_bn_eval(x, kids[0], kids[1], True, prologue=pro, s2d=s2d)
                    x = ops.maxpool3x3s2(x)
            pending = None
            for stage in kids[4:8]:
                for blk in stage:
                    if train:
                        pending = self._block_train(x, pending, blk, touched)
                        x = None
                    else:
                        x = self._block_eval(x, blk)
            if pending is not None:
                x = self._materialise(pending)
            for pool in kids[8:]:
                x = ops.adaptive_avgpool(x, *pool.out_hw)
            if touched:
                torch._foreach_add_(touched, 1)
        return x


    # ------------------------------------------------------------------ trainable mode
    def trainable_params(self):
        return [p for p in self.parameters() if p.requires_grad]

    def _all_convs(self):
        if getattr(self, "_conv_list", None) is None:
            self._conv_list = [m for m in self.modules() if isinstance(m, nn.Conv2d)]
        return self._conv_list

    def _rec_conv_bn(self, x, conv, bn, relu, residual, tape, touched):
        w = self._cache.conv(conv)
        stride, pad = conv.stride[0], conv.padding[0]
        if bn.training:
            raw, stats = ops.conv2d_nhwc(x, w, stride, pad, want_stats=True)
        else:
            raw, stats = ops.conv2d_nhwc(x, w, stride, pad), None
        y, sv = tb.bn_forward(raw, bn, relu, residual, stats, touched)
        tape.append(("conv_bn", x, conv, bn, w, sv))
        return y

    def run_recording(self, x):
        kids = list(self.children())
        tape, touched = [], []
        self._cache.prepare_trainable(self._all_convs())
        sc, sh = self.input_scale
        x = ops.scale_shift_act(x, sc, sh)  # /255 (+mean/std) materialised: wgrad reads it
        x = self._rec_conv_bn(x, kids[0], kids[1], True, None, tape, touched)
        x, saved = tb.maxpool_forward(x)
        tape.append(("maxpool", saved))
        for stage in kids[4:8]:
            for blk in stage:
                block_in = x
                identity = x
                tape.append(("block_begin", blk.downsample is not None))
                if blk.downsample is not None:
                    identity = self._rec_conv_bn(block_in, blk.downsample[0], blk.downsample[1],
                                                 False, None, tape, touched)
                tape.append(("main_begin",))
                st = blk.stages()
                cur = block_in
                for conv, bn in st[:-1]:
                    cur = self._rec_conv_bn(cur, conv, bn, True, None, tape, touched)
                x = self._rec_conv_bn(cur, st[-1][0], st[-1][1], True, identity, tape, touched)
                tape.append(("block_end",))
        for pool in kids[8:]:
            tape.append(("avgpool", tuple(x.shape), pool.out_hw))
            x = ops.adaptive_avgpool(x, *pool.out_hw)
        if touched:
            torch._foreach_add_(touched, 1)
        if any(m.training for m in self._norms):
            self._bn_gen += 1
        return x, tape

    def backward_from_tape(self, tape, dout):
        """Replays the tape in reverse.  Block structure on the tape:
        block_begin, [downsample conv_bn], main_begin, conv_bn*, block_end."""
        grads = {}

        def conv_bn_back(entry, dy, need_dx=True, add=None):
            _, x_in, conv, bn, w, sv = entry
            draw, dres, dg, db, pow2 = tb.bn_backward(
                dy, sv, max(conv.in_channels, conv.out_channels) if need_dx else 4)
            dx, dw = tb.conv_backward(x_in, w, draw, conv.stride[0], conv.padding[0], need_dx, add,
                                      pow2)
            for prm, g in ((conv.weight, dw), (bn.weight, dg), (bn.bias, db)):
                if prm.requires_grad:
                    grads[id(prm)] = g
            return dx, dres

        i = len(tape) - 1
        d = dout
        while i >= 0:
            kind = tape[i][0]
            if kind == "avgpool":
                d = tb.avgpool_backward(d, tape[i][1], tape[i][2])
                i -= 1
            elif kind == "block_end":
                # main path convs back to main_begin
                i -= 1
                d_main, d_skip = conv_bn_back(tape[i], d)  # last conv: residual gradient
                i -= 1
                # the two branches' gradients meet at the block input: the LAST data-gradient
                # convolution issued for the block adds the other branch in its epilogue
                while tape[i][0] != "main_begin":
                    first = tape[i - 1][0] == "main_begin"
                    plain_skip = first and tape[i - 2][0] != "conv_bn"
                    d_main, _ = conv_bn_back(tape[i], d_main, add=d_skip if plain_skip else None)
                    i -= 1
                i -= 1  # past main_begin
                if tape[i][0] == "conv_bn":  # downsample branch
                    d, _ = conv_bn_back(tape[i], d_skip, add=d_main)
                    i -= 1
                else:
                    d = d_main
                assert tape[i][0] == "block_begin"
                i -= 1
            elif kind == "maxpool":
                d = tb.maxpool_backward(d, tape[i][1])
                i -= 1
            elif kind == "conv_bn":  # the stem: its input is the image, no data gradient
                conv_bn_back(tape[i], d, need_dx=False)
                i -= 1
            else:
                raise AssertionError(kind)
        return grads


_TV_CHILD_INDEX = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6",
                   "layer4": "7"}


def torchvision_trunk_state_dict(sd):
    """torchvision `resnet18/50().state_dict()` (keys conv1.*, bn1.*, layer1.0.conv1.* ..., fc.*)
    -> the keys of `nn.Sequential(*children[:-1])` that the reference stores the trunk under
    (resnet_encoders.py:136-139): conv1 -> 0, bn1 -> 1, layer1..4 -> 4..7; the
    classifier `fc.*` is dropped.  A dict that already uses the Sequential keys passes through."""
    out = {}
    for k, v in sd.items():
        head, _, rest = k.partition(".")
        if head == "fc":
            continue
        out[(_TV_CHILD_INDEX.get(head, head) + "." + rest) if rest else k] = v
    return out


class TorchVisionResNet(nn.Module):
    """resnet_encoders.py:118-219.

    The reference builds `models.resnet18/50(pretrained=True)`, i.e. an ImageNet-initialised
    trunk that is frozen by default.  There is no network here, so the ImageNet weights come
    from a local file: `pretrained_weights` (config key MODEL.RGB_ENCODER.pretrained_weights,
    or the VLNCE_TORCHVISION_WEIGHTS environment variable) names a torchvision resnet
    state_dict (`resnet50-*.pth`); it is loaded strictly.  Without it the trunk keeps its
    random initialisation -- fine when a full policy checkpoint is restored afterwards or for
    synthetic benchmarks, WRONG for training from scratch with a frozen trunk, hence the
    warning."""

    def __init__(self, output_size, resnet_version="resnet50", normalize_visual_inputs=False,
                 trainable=False, spatial_output=False, single_spatial_filter=True,
                 pretrained_weights=None):
        super().__init__()
        self.normalize_visual_inputs = normalize_visual_inputs
        self.spatial_output = spatial_output
        if resnet_version == "resnet50":
            self.cnn = HipResNetTrunk(Bottleneck, [3, 4, 6, 3])
        elif resnet_version == "resnet18":
            self.cnn = HipResNetTrunk(BasicBlock, [2, 2, 2, 2])
        else:
            raise ValueError(resnet_version)
        self.resnet_layer_size = self.cnn.final_channels
        pretrained_weights = pretrained_weights or os.environ.get("VLNCE_TORCHVISION_WEIGHTS")
        if pretrained_weights and pretrained_weights != "NONE":
            self.load_torchvision_weights(pretrained_weights)
        elif not trainable:
            warnings.warn(
                f"TorchVisionResNet({resnet_version}): the frozen RGB trunk is RANDOMLY initialised "
                "(the reference uses ImageNet weights, pretrained=True).  Set "
                "MODEL.RGB_ENCODER.pretrained_weights / VLNCE_TORCHVISION_WEIGHTS to a torchvision "
                "state_dict, or restore a full policy checkpoint before training.", stacklevel=2)
        for p in self.cnn.parameters():
            p.requires_grad_(trainable)
        self.cnn.train(trainable)
        if not spatial_output:
            self.output_shape = (output_size,)
            self.fc = nn.Sequential(nn.Flatten(), nn.Linear(self.resnet_layer_size, output_size),
                                    nn.ReLU())
        else:
            if single_spatial_filter:
                del self.cnn[8]  # drop the global average pool
            self.cnn.avgpool = SpatialAvgPool()
            self.spatial_embeddings = nn.Embedding(4 * 4, 64)
            self.output_shape = (self.resnet_layer_size + 64, 4, 4)
        self._in_cache = None

    def load_torchvision_weights(self, path_or_state_dict):
        """strict load of a torchvision resnet state_dict into the trunk (before `del cnn[8]`
        or after: the pools hold no parameters)."""
        sd = path_or_state_dict
        if not isinstance(sd, dict):
            sd = torch.load(sd, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        self.cnn.load_state_dict(torchvision_trunk_state_dict(sd), strict=True)

    def _input_transform(self, device):
        # /255 then optional ImageNet mean/std (:171-192), fused into the stem's loader
        if self._in_cache is None or self._in_cache[0].device != device:
            if self.normalize_visual_inputs:
                mean = torch.tensor([0.485, 0.456, 0.406], device=device)
                std = torch.tensor([0.229, 0.224, 0.225], device=device)
                sc, sh = 1.0 / (255.0 * std), -mean / std
            else:
                sc = torch.full((3,), 1.0 / 255.0, device=device)
                sh = torch.zeros(3, device=device)
            self._in_cache = (sc.contiguous(), sh.contiguous())
        return self._in_cache

    def trunk_features(self, observations):
        """output of the torchvision trunk (what dagger_trainer.py:300-314 caches as
        `rgb_features`): logical [B, C, h, w]."""
        rgb = observations["rgb"]
        self.cnn.input_scale = self._input_transform(_GraphRunner._parts(rgb)[0].device)
        return self.cnn(rgb)

    def trunk_parameters(self):
        return self.cnn._plist()

    def trunk_ready(self, observations):
        self.cnn.input_scale = self._input_transform(_GraphRunner._parts(observations["rgb"])[0].device)
        return self.cnn.graph_ready(observations["rgb"])

    def forward(self, observations):
        if "rgb_features" in observations:
            feats = wait_ready(observations["rgb_features"], observations, "rgb_features")
        else:
            feats = self.trunk_features(observations)
        if not self.spatial_output:
            return ops.linear(feats.reshape(feats.size(0), -1), self.fc[1].weight, self.fc[1].bias,
                              ops.ACT_RELU)
        b, _, h, w = feats.shape
        x = torch.cat([_as_nhwc(feats), _grid_embedding_nhwc(self.spatial_embeddings, b, h, w)],
                      dim=3)
        return x.permute(0, 3, 1, 2)


class TorchVisionResNet50(TorchVisionResNet):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, resnet_version="resnet50", **kwargs)


class TorchVisionResNet18(TorchVisionResNet):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, resnet_version="resnet18", **kwargs)


# ------------------------------------------------------------------ habitat GroupNorm trunk
class GNBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None):
        super().__init__()
        self.convs = nn.Sequential(_c3(inplanes, planes, stride), nn.GroupNorm(ngroups, planes),
                                   nn.ReLU(True), _c3(planes, planes),
                                   nn.GroupNorm(ngroups, planes))
        self.downsample = downsample
        self.relu = nn.ReLU(True)


class GNBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None):
        super().__init__()
        self.convs = nn.Sequential(
            _c1(inplanes, planes), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
            _c3(planes, planes, stride), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
            _c1(planes, planes * 4), nn.GroupNorm(ngroups, planes * 4))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample


class GNResNet(nn.Module):
    """habitat_baselines.rl.ddppo.policy.resnet.ResNet parameter tree."""

    def __init__(self, in_channels, base_planes, ngroups, block, layers):
        super().__init__()
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_channels, base_planes, 7, stride=2, padding=3, bias=False),
            nn.GroupNorm(ngroups, base_planes), nn.ReLU(True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = base_planes
        self.layer1 = self._stage(block, ngroups, base_planes, layers[0], 1)
        self.layer2 = self._stage(block, ngroups, base_planes * 2, layers[1], 2)
        self.layer3 = self._stage(block, ngroups, base_planes * 4, layers[2], 2)
        self.layer4 = self._stage(block, ngroups, base_planes * 8, layers[3], 2)
        self.final_channels = self.inplanes
        self.final_spatial_compress = 1.0 / (2 ** 5)

    def _stage(self, block, ngroups, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(_c1(self.inplanes, planes * block.expansion, stride),
                                 nn.GroupNorm(ngroups, planes * block.expansion))
        blocks = [block(self.inplanes, planes, ngroups, stride, down)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes, ngroups) for _ in range(1, n)]
        return nn.Sequential(*blocks)


def resnet18(in_channels, base_planes, ngroups):
    return GNResNet(in_channels, base_planes, ngroups, GNBasicBlock, [2, 2, 2, 2])


def resnet50(in_channels, base_planes, ngroups):
    return GNResNet(in_channels, base_planes, ngroups, GNBottleneck, [3, 4, 6, 3])


class HipResNetEncoder(DropsGraphsOnApply, nn.Module):
    """habitat ResNetEncoder (depth-only use in VLN-CE): avg_pool2d(2) ->
    GroupNorm ResNet -> 3x3 compression conv + GroupNorm(1, C) + ReLU."""

    def __init__(self, observation_space, baseplanes=32, ngroups=32, spatial_size=128,
                 make_backbone=None, normalize_visual_inputs=False):
        super().__init__()
        sp = observation_space.spaces
        assert "rgb" not in sp, "VLN-CE builds the habitat encoder for depth only"
        self._n_input_rgb = 0
        self._n_input_depth = sp["depth"].shape[2]
        spatial_size = sp["depth"].shape[0] // 2
        assert not normalize_visual_inputs
        self.running_mean_and_var = nn.Sequential()
        self.backbone = make_backbone(self._n_input_depth, baseplanes, ngroups)
        final_spatial = int(spatial_size * self.backbone.final_spatial_compress)
        ncomp = int(round(2048 / (final_spatial ** 2)))
        self.compression = nn.Sequential(
            nn.Conv2d(self.backbone.final_channels, ncomp, 3, padding=1, bias=False),
            nn.GroupNorm(1, ncomp), nn.ReLU(True))
        self.output_shape = (ncomp, final_spatial, final_spatial)
        for layer in self.modules():
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
        self._cache = _WeightCache()
        self._graphs = _GraphRunner(self._forward_impl)

    @property
    def is_blind(self):
        return self._n_input_rgb + self._n_input_depth == 0

    def _conv_gn(self, x, conv, gn, relu, residual=None):
        if _stem_is_s2d(conv, x):
            x, w, stride, pad = ops.space_to_depth2(x, 2, 1), self._cache.stem_s2d(conv), 1, 0
        else:
            w, stride, pad = self._cache.conv(conv), conv.stride[0], conv.padding[0]
        return ops.conv_group_norm_act(x, w, stride, pad, gn.num_groups, gn.weight, gn.bias, gn.eps,
                                       residual=residual,
                                       act=ops.ACT_RELU if relu else ops.ACT_NONE)

    def _run_convs(self, x, seq, residual):
        """block.convs = [conv, GN, ReLU]* + [conv, GN]; the last GroupNorm output gets
        the skip connection added and then the block's ReLU."""
        mods, pairs, i = list(seq), [], 0
        while i < len(mods):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            pairs.append((mods[i], mods[i + 1], relu))
            i += 3 if relu else 2
        for j, (conv, gn, relu) in enumerate(pairs):
            if j == len(pairs) - 1:
                x = self._conv_gn(x, conv, gn, True, residual=residual)
            else:
                x = self._conv_gn(x, conv, gn, relu)
        return x

    def forward(self, observations):
        x = observations["depth"]  # [B,H,W,1] channels-last already; or the ops.frames tuple
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._plist()):
            x = ops.frames_f32(ops.frames(x))
            return tb.TrunkFn.apply(self, x, *self.trainable_params()).permute(0, 3, 1, 2)
        return self._graphs(x, self._graph_key(ops.frames_signature(x))).permute(0, 3, 1, 2)

    def _graph_key(self, signature):
        return (signature, tuple(p._version for p in self._plist(checked=False)), ops.plane_format())

    def graph_ready(self, x):
        return self._graphs.captured(self._graph_key(ops.frames_signature(x)))

    def _forward_impl(self, x):
        with torch.no_grad():
            x = ops.frames_avgpool2(ops.frames(x))
            bb = self.backbone
            x = self._conv_gn(x, bb.conv1[0], bb.conv1[1], True)
            x = ops.maxpool3x3s2(x)
            for stage in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
                for blk in stage:
                    identity = x
                    if blk.downsample is not None:
                        identity = self._conv_gn(x, blk.downsample[0], blk.downsample[1], False)
                    x = self._run_convs(x, blk.convs, identity)
            x = self._conv_gn(x, self.compression[0], self.compression[1], True)
        return x


# ---- trainable mode of HipResNetEncoder (methods attached below the class for readability)
def _depth_trainable_params(self):
    return [p for p in self.parameters() if p.requires_grad]


def _depth_rec_conv_gn(self, x, conv, gn, relu, residual, tape):
    w = self._cache.conv(conv)
    raw = ops.conv2d_nhwc(x, w, conv.stride[0], conv.padding[0])
    y, sv = tb.gn_forward(raw, gn, relu, residual)
    tape.append(("conv_gn", x, conv, gn, w, sv))
    return y


def _depth_run_recording(self, x):
    tape = []
    if getattr(self, "_conv_list", None) is None:
        self._conv_list = [m for m in self.modules() if isinstance(m, nn.Conv2d)]
    self._cache.prepare_trainable(self._conv_list)
    x = ops.avgpool2x2(x)
    bb = self.backbone
    x = self._rec_conv_gn(x, bb.conv1[0], bb.conv1[1], True, None, tape)
    x, saved = tb.maxpool_forward(x)
    tape.append(("maxpool", saved))
    for stage in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
        for blk in stage:
            block_in, identity = x, x
            tape.append(("block_begin",))
            if blk.downsample is not None:
                identity = self._rec_conv_gn(block_in, blk.downsample[0], blk.downsample[1], False,
                                             None, tape)
            tape.append(("main_begin",))
            mods, pairs, i = list(blk.convs), [], 0
            while i < len(mods):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                pairs.append((mods[i], mods[i + 1]))
                i += 3 if relu else 2
            cur = block_in
            for conv, gn in pairs[:-1]:
                cur = self._rec_conv_gn(cur, conv, gn, True, None, tape)
            x = self._rec_conv_gn(cur, pairs[-1][0], pairs[-1][1], True, identity, tape)
            tape.append(("block_end",))
    x = self._rec_conv_gn(x, self.compression[0], self.compression[1], True, None, tape)
    return x, tape


def _depth_backward_from_tape(self, tape, dout):
    grads = {}

    def back(entry, dy, need_dx=True, add=None):
        _, x_in, conv, gn, w, sv = entry
        draw, dres, dg, db, pow2 = tb.gn_backward(
            dy, sv, max(conv.in_channels, conv.out_channels) if need_dx else 4)
        dx, dw = tb.conv_backward(x_in, w, draw, conv.stride[0], conv.padding[0], need_dx, add, pow2)
        for prm, g in ((conv.weight, dw), (gn.weight, dg), (gn.bias, db)):
            if prm.requires_grad:
                grads[id(prm)] = g
        return dx, dres

    i = len(tape) - 1
    d = dout
    while i >= 0:
        kind = tape[i][0]
        if kind == "block_end":
            i -= 1
            d_main, d_skip = back(tape[i], d)
            i -= 1
            while tape[i][0] != "main_begin":   # (block-input sum in an epilogue: see the RGB trunk)
                first = tape[i - 1][0] == "main_begin"
                plain_skip = first and tape[i - 2][0] != "conv_gn"
                d_main, _ = back(tape[i], d_main, add=d_skip if plain_skip else None)
                i -= 1
            i -= 1
            if tape[i][0] == "conv_gn":
                d, _ = back(tape[i], d_skip, add=d_main)
                i -= 1
            else:
                d = d_main
            assert tape[i][0] == "block_begin"
            i -= 1
        elif kind == "maxpool":
            d = tb.maxpool_backward(d, tape[i][1])
            i -= 1
        elif kind == "conv_gn":
            # compression conv (has a data gradient) or the stem (input is the depth image)
            is_stem = i == 0
            d, _ = back(tape[i], d, need_dx=not is_stem)
            i -= 1
        else:
            raise AssertionError(kind)
    return grads


HipResNetEncoder.trainable_params = _depth_trainable_params
HipResNetEncoder._rec_conv_gn = _depth_rec_conv_gn
HipResNetEncoder.run_recording = _depth_run_recording
HipResNetEncoder.backward_from_tape = _depth_backward_from_tape


def single_frame_box_shape(box):
    """vlnce_baselines/common/utils.py:32-42."""
    if len(box.shape) < 4:
        return box
    return Box(float(np.min(box.low)), float(np.max(box.high)), box.shape[1:], box.high.dtype)


class VlnResnetDepthEncoder(nn.Module):
    """resnet_encoders.py:17-115."""

    def __init__(self, observation_space, output_size=128, checkpoint="NONE", backbone="resnet50",
                 resnet_baseplanes=32, normalize_visual_inputs=False, trainable=False,
                 spatial_output=False):
        super().__init__()
        self.visual_encoder = HipResNetEncoder(
            Dict({"depth": single_frame_box_shape(observation_space.spaces["depth"])}),
            baseplanes=resnet_baseplanes, ngroups=resnet_baseplanes // 2,
            make_backbone={"resnet18": resnet18, "resnet50": resnet50}[backbone],
            normalize_visual_inputs=normalize_visual_inputs)
        for p in self.visual_encoder.parameters():
            p.requires_grad_(trainable)
        if checkpoint != "NONE":
            # the published DD-PPO files (gibson-2plus-resnet50.pth ...) carry a pickled config
            # object next to "state_dict": a trusted local file, loaded as upstream does
            ddppo_weights = torch.load(checkpoint, map_location="cpu", weights_only=False)
            prefix = "actor_critic.net.visual_encoder."
            sd = {k[len(prefix):]: v for k, v in ddppo_weights["state_dict"].items()
                  if k.startswith(prefix)}
            del ddppo_weights
            self.visual_encoder.load_state_dict(sd, strict=True)
        self.spatial_output = spatial_output
        c, fh, fw = self.visual_encoder.output_shape
        if not spatial_output:
            self.output_shape = (output_size,)
            self.visual_fc = nn.Sequential(nn.Flatten(), nn.Linear(c * fh * fw, output_size),
                                           nn.ReLU(True))
            self._fc_cache = None
        else:
            self.spatial_embeddings = nn.Embedding(fh * fw, 64)
            self.output_shape = (c + 64, fh, fw)

    def _fc_weight_nhwc(self, c, h, w):
        """visual_fc consumes the NCHW flattening (index c*h*w + p); the features
        are stored NHWC (index p*c + ch), so the weight's columns are permuted
        once (view ops: autograd routes the gradient back to the parameter)."""
        wt = self.visual_fc[1].weight
        return wt.view(wt.size(0), c, h * w).permute(0, 2, 1).reshape(wt.size(0), h * w * c)

    def trunk_features(self, observations):
        """output of the habitat ResNetEncoder (cached upstream as `depth_features`)."""
        return self.visual_encoder(observations)

    def trunk_parameters(self):
        return self.visual_encoder._plist()

    def trunk_ready(self, observations):
        return self.visual_encoder.graph_ready(observations["depth"])

    def forward(self, observations):
        if "depth_features" in observations:
            x = wait_ready(observations["depth_features"], observations, "depth_features")
        else:
            x = self.trunk_features(observations)
        b, c, h, w = x.shape
        if self.spatial_output:
            y = torch.cat([_as_nhwc(x), _grid_embedding_nhwc(self.spatial_embeddings, b, h, w)],
                          dim=3)
            return y.permute(0, 3, 1, 2)
        return ops.linear(_as_nhwc(x).reshape(b, -1), self._fc_weight_nhwc(c, h, w),
                          self.visual_fc[1].bias, ops.ACT_RELU)
