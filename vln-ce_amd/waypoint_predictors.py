"""Waypoint prediction network (reference: vlnce_baselines/models/
waypoint_predictors.py:29-625): 12 panorama frames + 1 history frame through
the RGB (ResNet-18) / depth encoders, visual-history GRU, instruction
attention, per-frame spatial attention, panorama multi-head attention, main
GRU and the pano / offset / distance heads.  Module and parameter names match
the reference; all dense math runs on the HIP kernels with features kept as
[batch, positions, channels] rows."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .cma_policy import nchw_flat_weight, rows_of
from .encoders.instruction_encoder import InstructionEncoder
from .net_parts import build_depth_encoder, build_rgb_encoder, relu_fc
from .policy import Net
from .rnn_state_encoder import build_rnn_state_encoder
from .streams import BranchStreams, GraphedTail
from .utils import (CustomFixedCategorical, DotProductAttention, MultiHeadDotProductAttention,
                    TemperatureTanh)

PREV_ACTION_DIM = 4
PANO_ATTN_KEY_DIM = 128
ANGLE_FEATURE_SIZE = 4


def _lin(module, x, act=ops.ACT_NONE):
    return ops.linear(x, module.weight, module.bias, act)


class _WaypointTail(nn.Module):
    """Everything of WaypointPredictionNet.forward downstream of the three encoders, as a
    tensor-only callable with static shapes so that its forward AND backward replay as HIP
    graphs (streams.GraphedTail): ~150 small launches forward whose host-side issue cost is
    otherwise on the critical path.  It references the parent's sub-modules and is NOT one of
    its children.  Heads that the configuration does not have come back as empty tensors."""

    def __init__(self, net):
        super().__init__()
        object.__setattr__(self, "net", net)
        # the sub-modules it uses are (shared) children: make_graphed_callables takes the
        # parameters to differentiate from module.parameters()
        for name in ("rgb_pool_linear", "rgb_hist_linear", "depth_hist_linear", "visual_rnn",
                     "inst_attn_q", "inst_attn_k", "inst_attn", "text_q_linear", "rgb_kv_spatial",
                     "rgb_spatial_attn", "depth_kv_spatial", "depth_spatial_attn", "pano_attn",
                     "main_state_compress", "main_state_encoder", "stop_linear",
                     "compress_x_linear", "distance_linear", "distance_var_linear",
                     "offset_linear", "offset_var_linear"):
            if hasattr(net, name):
                setattr(self, name, getattr(net, name))

    def forward(self, ins, rgb, rgb_hist, dep, dep_hist, pa, angle_features, rnn_states, masks):
        n = self.net
        mc, wc = n.model_config, n.wypt_cfg
        P = n._num_panos
        hs = n._hidden_size
        half = hs // 2
        B = rgb.shape[0]
        Pr, Cr = rgb.shape[2:]
        Pd, Cd = dep.shape[2:]

        # visual history GRU (:275-284, 400-427)
        rl = n.rgb_encoder.resnet_layer_size
        pooled = ops.mean_rows(rgb.reshape(B * P, Pr, Cr))[:, :rl]  # mean over positions
        pooled = ops.mean_rows(_lin(n.rgb_pool_linear, pooled).view(B, P, -1))
        rgb_h = _lin(n.rgb_hist_linear[2], ops.mean_rows(rgb_hist.contiguous()), ops.ACT_RELU)
        dep_h = ops.linear(dep_hist.reshape(B, Pd * Cd),
                           nchw_flat_weight(n.depth_hist_linear[1], Cd, Pd),
                           n.depth_hist_linear[1].bias, ops.ACT_RELU)
        nv = n.visual_rnn.num_recurrent_layers
        vis, h1 = n.visual_rnn(torch.cat([pooled, pa, rgb_h, dep_h], dim=1),
                                  rnn_states[:, 0:nv], masks)

        # instruction attention -- multiplicative mask on PAD (:433-438, App. B-3)
        q = _lin(n.inst_attn_q[0], vis, ops.ACT_RELU)
        k = ops.linear(ins, n.inst_attn_k.weight.view(half, -1), n.inst_attn_k.bias)
        text = ops.attention(q, k, ins, ops.rowzero_mask(ins.detach()), 2,
                             n.inst_attn._scale_f)

        # spatial attention per pano frame; repeat_interleave WITHOUT dim repeats elements
        # (:456-462, App. B-4) -- reproduced verbatim
        tq = _lin(n.text_q_linear, text)
        tq = tq.repeat_interleave(P).view(B * P, tq.shape[1])
        rgb_kv = ops.linear(rgb.reshape(B * P, Pr, Cr), n.rgb_kv_spatial.weight.view(-1, Cr),
                            n.rgb_kv_spatial.bias)
        dep_kv = ops.linear(dep.reshape(B * P, Pd, Cd), n.depth_kv_spatial.weight.view(-1, Cd),
                            n.depth_kv_spatial.bias)
        att_rgb = ops.attention_kv(tq, rgb_kv, half, None, 0,
                                   n.rgb_spatial_attn._scale_f).view(B, P, -1)
        att_dep = ops.attention_kv(tq, dep_kv, half, None, 0,
                                   n.depth_spatial_attn._scale_f).view(B, P, -1)

        vis_feats = torch.cat([att_rgb, att_dep, angle_features], dim=2)  # [B,12,d]
        shared = vis_feats.permute(0, 2, 1)  # logical [B, d, 12]
        pano = n.pano_attn(Q=text, K=shared, V=shared)

        x = _lin(n.main_state_compress[0], torch.cat([text, pano, vis, pa], dim=1), ops.ACT_RELU)
        x, h2 = n.main_state_encoder(x, rnn_states[:, nv:], masks)
        rnn_states_out = torch.cat([h1, h2], dim=1)

        # heads (:549-625)
        x_small = _lin(n.compress_x_linear[0], x, ops.ACT_RELU).unsqueeze(1)
        dotted = (vis_feats * x_small).sum(2)
        pano_stop_logits = torch.cat([dotted, _lin(n.stop_linear, x)], dim=1)
        catted = torch.cat([vis_feats, x.unsqueeze(1).expand(-1, P, -1)], dim=2)

        if wc.continuous_distance:
            d1 = _lin(n.distance_linear[0], catted, ops.ACT_SIGMOID).squeeze(2)
            d1 = (wc.max_distance_prediction - wc.min_distance_prediction) * d1 \
                + wc.min_distance_prediction
            d2 = (wc.max_distance_var - wc.min_distance_var) * _lin(
                n.distance_var_linear[0], catted, ops.ACT_SIGMOID).squeeze(2) \
                + wc.min_distance_var
        else:
            d1, d2 = _lin(n.distance_linear, catted).squeeze(2), None
        if wc.continuous_offset:
            o1 = n.offset_scale * n.offset_linear[1](
                _lin(n.offset_linear[0], catted)).squeeze(2)
            o2 = (wc.max_offset_var - wc.min_offset_var) * _lin(
                n.offset_var_linear[0], catted, ops.ACT_SIGMOID).squeeze(2) + wc.min_offset_var
        else:
            o1, o2 = _lin(n.offset_linear, catted).squeeze(2), None
        empty = x.new_zeros(0)
        return (pano_stop_logits, o1, o2 if o2 is not None else empty, d1,
                d2 if d2 is not None else empty, x, rnn_states_out)


class WaypointPredictionNet(Net):
    def __init__(self, observation_space, model_config):
        super().__init__()
        self.model_config = model_config
        self.wypt_cfg = model_config.WAYPOINT
        self._hidden_size = model_config.STATE_ENCODER.hidden_size
        self._num_panos = model_config.num_panos
        hs = self._hidden_size
        r_out = model_config.RGB_ENCODER.output_size
        d_out = model_config.DEPTH_ENCODER.output_size

        # attribute order below = parameter / state_dict order of the reference net
        self.instruction_encoder = InstructionEncoder(model_config.INSTRUCTION_ENCODER)
        ins = self.instruction_encoder.output_size
        self.depth_encoder = build_depth_encoder(observation_space, model_config,
                                                 spatial_output=True, trainable=False)
        self.rgb_encoder = build_rgb_encoder(model_config, spatial_output=True,
                                             single_spatial_filter=False, trainable=False)
        rnn_type = model_config.STATE_ENCODER.rnn_type
        rgb_c = self.rgb_encoder.output_shape[0]
        depth_c = self.depth_encoder.output_shape[0]
        half = hs // 2
        pano_width = r_out + d_out + ANGLE_FEATURE_SIZE      # one panorama slot's feature

        # visual history (the frame the agent looked at last)
        self.visual_rnn = build_rnn_state_encoder(
            input_size=r_out + PREV_ACTION_DIM + d_out + r_out, hidden_size=hs, rnn_type=rnn_type,
            num_layers=1)
        self.rgb_pool_linear = nn.Linear(self.rgb_encoder.resnet_layer_size, r_out)
        self.rgb_hist_linear = relu_fc(rgb_c, r_out, nn.AdaptiveAvgPool1d(1), nn.Flatten())
        self.depth_hist_linear = relu_fc(int(np.prod(self.depth_encoder.output_shape)), d_out,
                                         nn.Flatten())
        # instruction attention, then spatial attention inside every panorama frame
        self.inst_attn_q = relu_fc(hs, half)
        self.inst_attn_k = nn.Conv1d(ins, half, 1)
        self.inst_attn = DotProductAttention(half)
        self.text_q_linear = nn.Linear(ins, half)
        self.rgb_kv_spatial = nn.Conv1d(rgb_c, half + r_out, 1)
        self.rgb_spatial_attn = DotProductAttention(half)
        self.depth_kv_spatial = nn.Conv1d(depth_c, half + d_out, 1)
        self.depth_spatial_attn = DotProductAttention(half)
        # attention over the 12 panorama slots, main recurrent state, heads
        self.pano_attn = MultiHeadDotProductAttention(
            d_q_in=ins, d_k_in=pano_width, d_v_in=pano_width, d_qk=PANO_ATTN_KEY_DIM,
            d_v=PANO_ATTN_KEY_DIM, num_heads=1, d_out=pano_width)
        self.main_state_compress = relu_fc(ins + pano_width + hs + PREV_ACTION_DIM, hs)
        self.main_state_encoder = build_rnn_state_encoder(
            input_size=hs, hidden_size=hs, rnn_type=rnn_type, num_layers=1)
        self.stop_linear = nn.Linear(hs, 1)
        nn.init.constant_(self.stop_linear.bias, 0)
        self.compress_x_linear = relu_fc(hs, pano_width)
        self._build_component_heads(hs + pano_width)
        # kept out of the module tree (it shares our sub-modules): see _WaypointTail
        object.__setattr__(self, "_tail", GraphedTail(lambda: _WaypointTail(self)))
        self._branches = BranchStreams()
        self.train()

    # ---- class index -> metres / radians (waypoint_predictors.py:184-215)
    @staticmethod
    def _bin_centre(index, low, high, bins):
        """value of class `index` when [low, high] is divided into `bins` classes, ends included"""
        return low + index * ((high - low) / (bins - 1))

    def distance_to_continuous(self, distance):
        wc = self.wypt_cfg
        if wc.continuous_distance:
            return distance
        return self._bin_centre(distance, wc.min_distance_prediction, wc.max_distance_prediction,
                                wc.discrete_distances)

    def offset_to_continuous(self, offset):
        if self.wypt_cfg.continuous_offset:
            return offset
        half_slot = np.pi / self._num_panos
        return self._bin_centre(offset, -half_slot, half_slot, self.wypt_cfg.discrete_offsets)

    @property
    def num_recurrent_layers(self):
        return (self.main_state_encoder.num_recurrent_layers
                + self.visual_rnn.num_recurrent_layers)

    @property
    def is_blind(self):
        return self.rgb_encoder.is_blind and self.depth_encoder.is_blind

    @property
    def output_size(self):
        return self._hidden_size

    def _build_component_heads(self, width):
        """distance / offset heads over [panorama slot feature, recurrent state] rows of `width`
        (waypoint_predictors.py:217-264): a continuous component has a squashed mean head and a
        sigmoid variance head, a discrete one a single classifier."""
        wc = self.wypt_cfg

        def squashed(activation):
            return nn.Sequential(nn.Linear(width, 1), activation)

        if wc.continuous_distance:
            self.distance_linear = squashed(nn.Sigmoid())
            self.distance_var_linear = squashed(nn.Sigmoid())
        else:
            self.distance_linear = nn.Linear(width, wc.discrete_distances)
        if wc.continuous_offset:
            self.offset_linear = squashed(TemperatureTanh(temperature=wc.offset_temperature))
            self.offset_scale = np.pi / self._num_panos
            self.offset_var_linear = squashed(nn.Sigmoid())
        else:
            self.offset_linear = nn.Linear(width, wc.discrete_offsets)

    def _encode_frames(self, encoder, key, frames, history, masks):
        """12 pano frames + the (done-masked) history frame as one batch of B*13 images
        (:330-375).  Returns rows [B, 12, P, C] and [B, P, C]."""
        # the ingest kernel reads the 12 frames and the history frame in place (times its
        # not-done mask) -- no concatenated fp32 copy of B*13 frames (ops.frames)
        b, n = frames.size(0), frames.size(1) + 1
        emb = rows_of(encoder({key: (frames, history, masks.reshape(-1))}))  # [B*13, P, C]
        emb = emb.reshape(b, n, *emb.shape[1:])
        return emb[:, : self._num_panos], emb[:, self._num_panos]

    def forward(self, observations, rnn_states, prev_actions, masks):
        for k in ("rgb", "depth", "instruction", "rgb_history", "depth_history", "angle_features"):
            assert k in observations
        P = self._num_panos
        assert observations["rgb"].shape[1] == P
        assert observations["depth"].shape[1] == P
        mc, wc = self.model_config, self.wypt_cfg
        hs = self._hidden_size
        half = hs // 2

        # the instruction encoder (200-token RxR-length instructions: a 200-step recurrence on a
        # handful of workgroups, ~0.7 ms, and one host sync for the lengths) on a side stream under
        # the two trunks' 2 x 13 x B frames; autograd runs its backward on that stream too
        fork = self._branches.fork(rnn_states.device)
        ins, join_ins = self._branches.run(fork, 0, rnn_states.device,
                                           lambda: self.instruction_encoder(observations))
        rgb, rgb_hist = self._encode_frames(self.rgb_encoder, "rgb", observations["rgb"],
                                            observations["rgb_history"], masks)
        dep, dep_hist = self._encode_frames(self.depth_encoder, "depth", observations["depth"],
                                            observations["depth_history"], masks)
        join_ins()
        ins = ins.permute(0, 2, 1)  # [B, L, C]
        B = rgb.shape[0]

        if len(prev_actions["pano"].shape) == 1:  # :380-382, mutates the caller's dict
            for k in prev_actions:
                prev_actions[k] = prev_actions[k].unsqueeze(1)
        heading = prev_actions["pano"] * ((np.pi * 2) / P)
        pa = torch.cat([torch.sin(heading), torch.cos(heading),
                        self.offset_to_continuous(prev_actions["offset"]),
                        self.distance_to_continuous(prev_actions["distance"])],
                       dim=1).float() * masks

        if mc.ablate_instruction:
            ins = ins * 0
        if mc.ablate_rgb:
            rgb, rgb_hist = rgb * 0, rgb_hist * 0
        if mc.ablate_depth:
            dep, dep_hist = dep * 0, dep_hist * 0
        ins = ins.contiguous()
        Pr, Cr = rgb.shape[2:]
        Pd, Cd = dep.shape[2:]

        logits, o1, o2, d1, d2, x, rnn_states_out = self._tail(
            ins, rgb.contiguous(), rgb_hist.contiguous(), dep.contiguous(), dep_hist.contiguous(),
            pa.contiguous(), observations["angle_features"].contiguous(), rnn_states.contiguous(),
            masks)
        pano_stop = CustomFixedCategorical(logits=logits)
        if not wc.continuous_distance:
            d2 = None
        if not wc.continuous_offset:
            o2 = None
        return pano_stop, o1, o2, d1, d2, x, rnn_states_out
