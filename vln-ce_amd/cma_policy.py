"""Cross-modal attention policy (reference: vlnce_baselines/models/
cma_policy.py:24-309; arXiv 2004.02857).  Same module / parameter names; the
arithmetic runs on the HIP kernels with visual and text features kept as
[B, positions, channels] rows (the reference's [B, C, P] tensors permuted)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .encoders.instruction_encoder import InstructionEncoder
from .net_parts import (apply_ablations, build_depth_encoder, build_rgb_encoder,
                        encode_three_branches, prev_action_index, register_progress_loss)
from .policy import ILPolicy, Net
from .registry import baseline_registry
from .rnn_state_encoder import build_rnn_state_encoder
from .streams import BranchStreams, GraphedTail, bucket_rows


@baseline_registry.register_policy
class CMAPolicy(ILPolicy):
    def __init__(self, observation_space, action_space, model_config):
        super().__init__(
            CMANet(observation_space=observation_space, model_config=model_config,
                   num_actions=action_space.n),
            action_space.n,
        )


def rows_of(feature_map):
    """logical [B, C, h, w] (NHWC memory) -> [B, h*w, C] rows without a copy."""
    b, c = feature_map.shape[:2]
    return feature_map.permute(0, 2, 3, 1).reshape(b, -1, c)


def nchw_flat_weight(linear, c, p):
    """nn.Linear over the reference's Flatten of [B, C, P] (column c*P + p) re-indexed
    for [B, P, C] rows (column p*C + c)."""
    w = linear.weight
    return w.view(w.size(0), c, p).permute(0, 2, 1).reshape(w.size(0), p * c)


class _CMATail(nn.Module):
    """The part of CMANet.forward downstream of the encoders, as a tensor-only callable
    (no host syncs, static shapes) so that it can be captured as a HIP graph.  It shares
    the parent's sub-modules; it is NOT registered as a child of the parent."""

    def __init__(self, net, rgb_frozen=0, dep_frozen=0):
        """`rgb_frozen` / `dep_frozen`: leading channels of the rgb / depth rows that carry no
        gradient (a frozen trunk's or cached features; the trailing 64 are the trainable spatial
        embeddings) -- the input gradients of rgb_kv / depth_kv are computed for the rest only."""
        super().__init__()
        self._rgb_frozen, self._dep_frozen = int(rgb_frozen), int(dep_frozen)
        for name in ("rgb_linear", "depth_linear", "state_encoder", "rgb_kv", "depth_kv", "state_q",
                     "text_k", "text_q", "second_state_compress", "second_state_encoder"):
            setattr(self, name, getattr(net, name))
        self._hidden_size = net._hidden_size
        self._scale_f = net._scale_f

    def forward(self, ins, dep, rgb, act, rnn_states, masks, ins_index=None):
        """`ins_index` (int64 [B]) given: `ins` holds the U DISTINCT instructions of a
        sequence-mode batch and row b attends over ins[ins_index[b]] -- text_k, the padding mask and
        the attention's K / V are then U blocks instead of B copies."""
        B = dep.size(0)
        L, Ci = ins.shape[1:]
        P_d, C_d = dep.shape[1:]
        C_r = rgb.shape[2]
        half = self._hidden_size // 2
        scale = self._scale_f
        rgb_in = ops.linear(ops.mean_rows(rgb), self.rgb_linear[2].weight,
                            self.rgb_linear[2].bias, ops.ACT_RELU)
        depth_in = ops.linear(dep.reshape(B, P_d * C_d),
                              nchw_flat_weight(self.depth_linear[1], C_d, P_d),
                              self.depth_linear[1].bias, ops.ACT_RELU)
        state_in = torch.cat([rgb_in, depth_in, act], dim=1)
        n1 = self.state_encoder.num_recurrent_layers
        state, h1 = self.state_encoder(state_in, rnn_states[:, 0:n1], masks)

        text_state_q = ops.linear(state, self.state_q.weight, self.state_q.bias)
        text_state_k = ops.linear(ins, self.text_k.weight.view(half, Ci), self.text_k.bias)
        text_mask = ops.rowzero_mask(ins.detach())  # (instruction_embedding == 0).all(dim=1)
        # softmax((q.k - 1e8 mask) * scale) . v over [B, P, C] rows (cma_policy.py:207-217)
        text_embedding = ops.attention(text_state_q, text_state_k, ins, text_mask, 1, scale,
                                       index=ins_index)

        rgb_kv = ops.linear(rgb, self.rgb_kv.weight.view(-1, C_r), self.rgb_kv.bias,
                            dx_from=self._rgb_frozen)
        depth_kv = ops.linear(dep, self.depth_kv.weight.view(-1, C_d), self.depth_kv.bias,
                              dx_from=self._dep_frozen)
        text_q = ops.linear(text_embedding, self.text_q.weight, self.text_q.bias)
        rgb_embedding = ops.attention_kv(text_q, rgb_kv, half, None, 1, scale)
        depth_embedding = ops.attention_kv(text_q, depth_kv, half, None, 1, scale)

        x = torch.cat([state, text_embedding, rgb_embedding, depth_embedding, act], dim=1)
        x = ops.linear(x, self.second_state_compress[0].weight,
                       self.second_state_compress[0].bias, ops.ACT_RELU)
        x, h2 = self.second_state_encoder(x, rnn_states[:, n1:], masks)
        return x, torch.cat([h1, h2], dim=1)


class CMANet(Net):
    def __init__(self, observation_space, model_config, num_actions):
        super().__init__()
        self.model_config = model_config
        model_config.defrost()
        model_config.INSTRUCTION_ENCODER.final_state_only = False
        model_config.freeze()
        self.instruction_encoder = InstructionEncoder(model_config.INSTRUCTION_ENCODER)
        self.depth_encoder = build_depth_encoder(observation_space, model_config,
                                                 spatial_output=True)
        self.rgb_encoder = build_rgb_encoder(model_config, spatial_output=True)
        self.prev_action_embedding = nn.Embedding(num_actions + 1, 32)
        hidden_size = model_config.STATE_ENCODER.hidden_size
        self._hidden_size = hidden_size
        r_out = model_config.RGB_ENCODER.output_size
        d_out = model_config.DEPTH_ENCODER.output_size
        self.rgb_linear = nn.Sequential(
            nn.AdaptiveAvgPool1d(1), nn.Flatten(),
            nn.Linear(self.rgb_encoder.output_shape[0], r_out), nn.ReLU(True))
        self.depth_linear = nn.Sequential(
            nn.Flatten(), nn.Linear(int(np.prod(self.depth_encoder.output_shape)), d_out),
            nn.ReLU(True))
        self.state_encoder = build_rnn_state_encoder(
            input_size=d_out + r_out + self.prev_action_embedding.embedding_dim,
            hidden_size=hidden_size, rnn_type=model_config.STATE_ENCODER.rnn_type, num_layers=1)
        self._output_size = (hidden_size + r_out + d_out + self.instruction_encoder.output_size)
        self.rgb_kv = nn.Conv1d(self.rgb_encoder.output_shape[0], hidden_size // 2 + r_out, 1)
        self.depth_kv = nn.Conv1d(self.depth_encoder.output_shape[0], hidden_size // 2 + d_out, 1)
        self.state_q = nn.Linear(hidden_size, hidden_size // 2)
        self.text_k = nn.Conv1d(self.instruction_encoder.output_size, hidden_size // 2, 1)
        self.text_q = nn.Linear(self.instruction_encoder.output_size, hidden_size // 2)
        self.register_buffer("_scale", torch.tensor(1.0 / ((hidden_size // 2) ** 0.5)))
        self._scale_f = 1.0 / ((hidden_size // 2) ** 0.5)
        self.second_state_compress = nn.Sequential(
            nn.Linear(self._output_size + self.prev_action_embedding.embedding_dim, hidden_size),
            nn.ReLU(True))
        self.second_state_encoder = build_rnn_state_encoder(
            input_size=hidden_size, hidden_size=hidden_size,
            rnn_type=model_config.STATE_ENCODER.rnn_type, num_layers=1)
        self._output_size = hidden_size
        self._branches = BranchStreams()
        # kept out of the module tree (shares our sub-modules): see _CMATail
        object.__setattr__(self, "_tail", GraphedTail(lambda *static: _CMATail(self, *static)))
        self.progress_monitor = nn.Linear(self.output_size, 1)
        if model_config.PROGRESS_MONITOR.use:
            nn.init.kaiming_normal_(self.progress_monitor.weight, nonlinearity="tanh")
            nn.init.constant_(self.progress_monitor.bias, 0)
        self.train()

    @property
    def output_size(self):
        return self._output_size

    @property
    def is_blind(self):
        return self.rgb_encoder.is_blind or self.depth_encoder.is_blind

    @property
    def num_recurrent_layers(self):
        return (self.state_encoder.num_recurrent_layers
                + self.second_state_encoder.num_recurrent_layers)

    def _attn(self, q, k, v, mask=None):
        """softmax((q.k - 1e8 mask) * scale) . v  over [B, P, C] rows (cma_policy.py:207-217)."""
        return ops.attention(q, k, v, mask, 1, self._scale_f)

    def _frozen_cols(self, enc, cached_key, observations):
        """leading channels of an encoder's [B, C, h, w] output that cannot need a gradient: the
        trunk's own channels when the trunk is frozen or its features came in precomputed
        (the channels behind them are the encoder's trainable spatial embeddings)."""
        if any(getattr(self.model_config, "ablate_" + k) for k in ("rgb", "depth")):
            return 0
        if not hasattr(enc, "spatial_embeddings") or not getattr(enc, "spatial_output", False):
            return 0
        if cached_key in observations:
            if getattr(observations[cached_key], "requires_grad", False):
                return 0
        elif any(p.requires_grad for p in enc.trunk_parameters()):
            return 0
        return enc.output_shape[0] - enc.spatial_embeddings.embedding_dim

    def forward(self, observations, rnn_states, prev_actions, masks):
        # three independent encoders: RGB trunk on this stream, instruction RNN + depth trunk on
        # a side stream (net_parts.encode_three_branches); then everything as [B, rows, C]
        ins, dep, rgb = encode_three_branches(self, observations, rnn_states.device,
                                              distinct_instructions=True)
        ins, ins_index = ins   # [U, 2H, L] distinct instructions + row map, or ([B, 2H, L], None)
        ins, dep, rgb = apply_ablations(self.model_config, ins.permute(0, 2, 1),  # [B, L, 2H]
                                        rows_of(dep),                             # [B, P, 192]
                                        rows_of(rgb))                             # [B, 16, 2112]
        act = F.embedding(prev_action_index(prev_actions, masks),
                          self.prev_action_embedding.weight)
        masks_u8 = masks.reshape(-1).to(torch.uint8)
        if ins.is_cuda:
            # Lmax is batch-dependent (App. B-8): pad it to a bucket so that the captured tail
            # graphs are shared across batches.  All-zero rows are exactly what rowzero_mask /
            # the additive -1e8 text-attention mask already treat as padding (weight exp(-inf)=0).
            pad = bucket_rows(ins.size(1)) - ins.size(1)
            if pad:
                ins = F.pad(ins, (0, 0, 0, pad))
        extra = () if ins_index is None else (ins_index,)
        x, rnn_states_out = self._tail(ins.contiguous(), dep.contiguous(), rgb.contiguous(), act,
                                       rnn_states.contiguous(), masks_u8, *extra,
                                       static=(self._frozen_cols(self.rgb_encoder, "rgb_features",
                                                                 observations),
                                               self._frozen_cols(self.depth_encoder, "depth_features",
                                                                 observations)))
        register_progress_loss(self, x, observations)
        return x, rnn_states_out
