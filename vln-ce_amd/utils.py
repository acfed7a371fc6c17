"""Distributions and small heads of the policies (reference:
vlnce_baselines/models/utils.py:12-317).  The categorical / truncated-normal
math is O(N x 13) scalar work and stays in torch; the
attention classes run the fused HIP attention kernel."""
import math
from numbers import Number

import numpy as np
import torch
import torch.nn as nn

from . import ops


class TemperatureTanh(nn.Module):
    def __init__(self, temperature=1.0):
        super().__init__()
        assert temperature != 0.0, "temperature must be nonzero."
        self._T = temperature

    def forward(self, x):
        return torch.tanh(x / self._T)


def _phi(x):
    return (np.e ** (-0.5 * (x ** 2))) / math.sqrt(2 * math.pi)


def _Phi(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2.0)))


class TruncatedNormal(nn.Module):
    """Two-sided truncated normal (utils.py:24-152): mean/variance/entropy are
    precomputed at construction; mode() = loc; sampling is rejection."""

    def __init__(self, loc, scale, smin=-np.inf, smax=np.inf, validate_args=None):
        super().__init__()
        assert smin < smax, "smin must be less than smax"
        assert np.isfinite(smin) and np.isfinite(smax), "two-sided truncation is required"
        assert (loc >= smin).all() and (loc <= smax).all(), f"loc is out of range ({smin}, {smax})"
        if isinstance(scale, Number):
            assert scale >= 0.0, "scale is negative"
        else:
            assert (scale >= 0.0).all(), "scale is negative"
        self._normal = torch.distributions.Normal(loc, scale, validate_args=False)
        self._loc, self._scale, self._smin, self._smax = loc, scale, smin, smax
        self.A = 1 / (scale * math.sqrt(2 * math.pi))
        hi = torch.as_tensor(smax, dtype=loc.dtype, device=loc.device)
        lo = torch.as_tensor(smin, dtype=loc.dtype, device=loc.device)
        self.Z = self._normal.cdf(hi) - self._normal.cdf(lo)
        alpha = (smin - loc) / scale
        beta = (smax - loc) / scale
        a_pdf, b_pdf = _phi(alpha), _phi(beta)
        z = _Phi(beta) - _Phi(alpha)
        self._mean = loc - scale * ((b_pdf - a_pdf) / z)
        t1 = (beta * b_pdf - alpha * a_pdf) / z
        t2 = ((b_pdf - a_pdf) / z) ** 2
        self._variance = (scale ** 2) * (1 - t1 - t2)
        ent = 0.5 * np.log(2 * np.pi * np.e) + torch.log(scale * z)
        self._entropy = ent + (alpha * a_pdf - beta * b_pdf) / (2 * z)

    @property
    def mean(self):
        return self._mean

    @property
    def variance(self):
        return self._variance

    def sample(self, resample_limit=10000):
        s = self._normal.sample()
        bad = (s < self._smin).logical_or(s > self._smax)
        n = 0
        while bad.any():
            assert n < resample_limit, f"Hit resample limit of {resample_limit}"
            n += 1
            s[bad] = self._normal.sample()[bad]
            bad = (s < self._smin).logical_or(s > self._smax)
        return s

    def log_prob(self, value):
        msg = "value is out of truncation range and has an undefined log_prob."
        if isinstance(value, Number):
            assert self._smin <= value <= self._smax, msg
        else:
            assert (value >= self._smin).all() and (value <= self._smax).all(), msg
        dens = self.A * np.e ** (-0.5 * ((value - self._loc) / self._scale) ** 2)
        dens = dens / self.Z
        return np.log(dens) if isinstance(dens, Number) else dens.log()

    def mode(self):
        return self._loc

    def entropy(self):
        return self._entropy


class DotProductAttention(nn.Module):
    """utils.py:155-178.  Q [B,Dk], K [B,Dk,P], V [B,Dv,P] (logical layouts of
    the reference); the mask is MULTIPLICATIVE on the energies (App. B-3)."""

    def __init__(self, key_dimension):
        super().__init__()
        self.scale = torch.tensor(1.0 / (key_dimension ** 0.5))
        self._scale_f = 1.0 / (key_dimension ** 0.5)

    def forward(self, Q, K, V, mask=None):
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        return ops.attention(Q, K.permute(0, 2, 1), V.permute(0, 2, 1), m, 2, self._scale_f)


class MultiHeadDotProductAttention(nn.Module):
    """utils.py:181-266."""

    def __init__(self, d_q_in, d_k_in, d_v_in, d_qk, d_v, num_heads, d_out, normalize=True,
                 dropout_p=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.normalize = normalize
        self.q_linear = nn.Linear(d_q_in, d_qk * num_heads, bias=False)
        self.k_linear = nn.Linear(d_k_in, d_qk * num_heads, bias=False)
        self.v_linear = nn.Linear(d_v_in, d_v * num_heads, bias=False)
        self.attn = DotProductAttention(d_qk)
        self.final_linear = nn.Linear(d_v * num_heads, d_out, bias=False)
        assert dropout_p == 0.0, "the reference never enables attention dropout"
        self.dropout = None
        if self.normalize:
            self.layer_norm = nn.LayerNorm(d_out, eps=1e-6)

    def forward(self, Q, K, V, mask=None):
        """Q [B,d_q_in]; K [B,d_k_in,P]; V [B,d_v_in,P]."""
        assert K.shape[2] == V.shape[2], "keys must be the same size as values"
        nh = self.num_heads
        B, P = K.shape[0], K.shape[2]
        q = ops.linear(Q, self.q_linear.weight)                       # [B, nh*dqk]
        k = ops.linear(K.permute(0, 2, 1), self.k_linear.weight)      # [B, P, nh*dqk]
        v = ops.linear(V.permute(0, 2, 1), self.v_linear.weight)      # [B, P, nh*dv]
        dqk, dv = q.shape[1] // nh, v.shape[2] // nh
        if nh == 1:
            a = ops.attention(q, k, v, mask, 2, self.attn._scale_f)
        else:
            qh = q.view(B * nh, dqk)
            kh = k.view(B, P, nh, dqk).permute(0, 2, 1, 3).reshape(B * nh, P, dqk)
            vh = v.view(B, P, nh, dv).permute(0, 2, 1, 3).reshape(B * nh, P, dv)
            a = ops.attention(qh, kh, vh, mask, 2, self.attn._scale_f).view(B, nh * dv)
        out = ops.linear(a, self.final_linear.weight)
        if self.normalize:
            out = torch.nn.functional.layer_norm(out, self.layer_norm.normalized_shape,
                                                 self.layer_norm.weight, self.layer_norm.bias,
                                                 self.layer_norm.eps)
        return out


class CustomFixedCategorical(torch.distributions.Categorical):
    """utils.py:269-289."""

    def sample(self, sample_shape=torch.Size()):
        return super().sample(sample_shape).unsqueeze(-1)

    def log_prob(self, actions):
        return (super().log_prob(actions.squeeze(-1)).view(actions.size(0), -1).sum(-1)
                .unsqueeze(-1))

    def log_probs(self, actions):  # habitat-lab spelling
        return self.log_prob(actions)

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)


def batched_index_select(x, dim, index):
    """utils.py:292-317."""
    views = [x.shape[0]] + [1 if i != dim else -1 for i in range(1, len(x.shape))]
    expanse = list(x.shape)
    expanse[0] = -1
    expanse[dim] = -1
    return torch.gather(x, dim, index.view(views).expand(expanse)).squeeze(dim)
