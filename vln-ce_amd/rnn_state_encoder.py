"""Recurrent state encoder with done-mask reset (habitat-lab v0.1.7
`build_rnn_state_encoder`, call sites seq2seq_policy.py:109-114,164;
cma_policy.py:126-131,172-177; waypoint_predictors.py:69-74,157-162).

hidden_states are batch-first [N, L, H] (LSTM packs (h, c) as two layers).
x is [N, D] for one step or the time-major flattening [T*N, D]; the input
projection of all T steps is a single MFMA GEMM, each step then runs
mask -> h W_hh^T (GEMM) -> fused gate kernel.  T > 1 (cached-feature DAgger batches,
DD-PPO minibatches) goes through ops.MaskedRNNSeqFn: one autograd node for the rollout.
"""
import torch
import torch.nn as nn

from . import ops


class RNNStateEncoder(nn.Module):
    def __init__(self, input_size, hidden_size, rnn_type="GRU", num_layers=1):
        super().__init__()
        assert num_layers == 1, "the VLN-CE policies only build single-layer state encoders"
        self.is_lstm = rnn_type == "LSTM"
        self.rnn = (nn.LSTM if self.is_lstm else nn.GRU)(input_size, hidden_size, num_layers)
        self.num_recurrent_layers = num_layers * (2 if self.is_lstm else 1)
        for name, p in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)

    def forward(self, x, hidden_states, masks):
        n = hidden_states.size(0)
        t_steps = x.size(0) // n
        assert t_steps * n == x.size(0)
        m_u8 = masks.reshape(-1).to(torch.uint8).contiguous()
        r = self.rnn
        gi = ops.linear(x, r.weight_ih_l0, r.bias_ih_l0)
        h = hidden_states[:, 0]
        c = hidden_states[:, 1] if self.is_lstm else None
        if t_steps > 1 and r.bias_hh_l0 is not None:
            # a whole rollout: one autograd node, batched recurrent-weight gradients
            y, h, c = ops.MaskedRNNSeqFn.apply(self.is_lstm, gi, h, c, m_u8, r.weight_hh_l0,
                                               r.bias_hh_l0)
            new_states = torch.stack([h, c], dim=1) if self.is_lstm else h.unsqueeze(1)
            return y, new_states
        outs = []
        for t in range(t_steps):
            # (one step: no slicing.  The autograd of a slice is zeros + copy_, and a dense copy_
            # is a MEMCPY NODE in the tail's backward graph, not a kernel node.)
            m = m_u8 if t_steps == 1 else m_u8[t * n:(t + 1) * n]
            g = gi if t_steps == 1 else gi[t * n:(t + 1) * n]
            h = ops.mask_rows(h, m)
            if self.is_lstm:
                c = ops.mask_rows(c, m)
                h, c = ops.lstm_cell(g, h, c, r.weight_hh_l0, r.bias_hh_l0)
            else:
                h = ops.gru_cell(g, h, r.weight_hh_l0, r.bias_hh_l0)
            outs.append(h)
        y = outs[0] if t_steps == 1 else torch.cat(outs, dim=0)
        new_states = torch.stack([h, c], dim=1) if self.is_lstm else h.unsqueeze(1)
        return y, new_states


def build_rnn_state_encoder(input_size, hidden_size, rnn_type="GRU", num_layers=1):
    return RNNStateEncoder(input_size, hidden_size, rnn_type, num_layers)
