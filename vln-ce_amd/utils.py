"""Distributions and small heads of the policies.

Plugin-surface mirror of vlnce_baselines/models/utils.py:12-317 (class / method / parameter
names are what the policies, checkpoints and habitat's PPO code touch).  The categorical and
truncated-normal math is O(N x 13) scalars per step and stays in torch;
the attention modules run the fused HIP attention kernel on [B, rows, C] tensors.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_SQRT_2PI = math.sqrt(2.0 * math.pi)
_HALF_LOG_2PIE = 0.5 * math.log(2.0 * math.pi * math.e)


class TemperatureTanh(nn.Module):
    """tanh(x / T): squashes a head's output into (-1, 1) with an adjustable slope."""

    def __init__(self, temperature=1.0):
        super().__init__()
        if temperature == 0.0:
            raise AssertionError("temperature must be nonzero.")
        self._T = temperature

    def forward(self, x):
        return torch.tanh(x / self._T)


def _pdf(z):
    """standard normal density"""
    return torch.exp(-0.5 * z * z) / _SQRT_2PI


def _cdf(z):
    """standard normal distribution function"""
    return 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0)))


class TruncatedNormal(nn.Module):
    """N(loc, scale) restricted to [smin, smax] (reference utils.py:24-152).

    Everything is expressed through the standardised window [a, b] = ([smin, smax] - loc) / scale
    and its probability mass under the parent normal:
        mean     = loc + scale (pdf(a) - pdf(b)) / mass
        variance = scale^2 [1 + (a pdf(a) - b pdf(b)) / mass - ((pdf(a) - pdf(b)) / mass)^2]
        entropy  = log(sqrt(2 pi e) scale mass) + (a pdf(a) - b pdf(b)) / (2 mass)
        log p(v) = -z^2 / 2 - log(sqrt(2 pi) scale mass),  z = (v - loc) / scale
    mode() is loc (the constructor requires loc inside the window); sampling is by rejection."""

    def __init__(self, loc, scale, smin=-math.inf, smax=math.inf, validate_args=None):
        super().__init__()
        if not (smin < smax):
            raise AssertionError("smin must be less than smax")
        if not (math.isfinite(smin) and math.isfinite(smax)):
            raise AssertionError("two-sided truncation is required")
        if not bool(((loc >= smin) & (loc <= smax)).all()):
            raise AssertionError(f"loc is out of range ({smin}, {smax})")
        if not bool(torch.as_tensor(scale >= 0.0).all()):
            raise AssertionError("scale is negative")
        self._loc, self._scale = loc, scale
        self._window = (smin, smax)
        self._a = (smin - loc) / scale
        self._b = (smax - loc) / scale
        self._mass = _cdf(self._b) - _cdf(self._a)
        self._parent = torch.distributions.Normal(loc, scale, validate_args=False)

    # -- moments ---------------------------------------------------------------------------
    def _edge_terms(self):
        pa, pb = _pdf(self._a), _pdf(self._b)
        return (pa - pb) / self._mass, (self._a * pa - self._b * pb) / self._mass

    @property
    def mean(self):
        first, _ = self._edge_terms()
        return self._loc + self._scale * first

    @property
    def variance(self):
        first, second = self._edge_terms()
        return self._scale ** 2 * (1.0 + second - first ** 2)

    def entropy(self):
        _, second = self._edge_terms()
        return _HALF_LOG_2PIE + torch.log(self._scale * self._mass) + 0.5 * second

    def mode(self):
        return self._loc

    # -- density / sampling ----------------------------------------------------------------
    def _inside(self, value):
        lo, hi = self._window
        return (value >= lo) & (value <= hi)

    def log_prob(self, value):
        value = torch.as_tensor(value, dtype=self._loc.dtype, device=self._loc.device)
        if not bool(self._inside(value).all()):
            raise AssertionError("value is out of truncation range and has an undefined log_prob.")
        z = (value - self._loc) / self._scale
        return -0.5 * z * z - torch.log(_SQRT_2PI * self._scale * self._mass)

    def sample(self, resample_limit=10000):
        draw = self._parent.sample()
        for _ in range(resample_limit):
            outside = ~self._inside(draw)
            if not bool(outside.any()):
                return draw
            draw = torch.where(outside, self._parent.sample(), draw)
        raise AssertionError(f"Hit resample limit of {resample_limit}")


def _attend(query, keys_cl, values_cl, mask, scale):
    """softmax over positions of (q.k) * mask * scale, applied to the values.  keys / values are
    channels-last [B, P, C]; the mask is MULTIPLICATIVE on the energies."""
    m = None if mask is None else mask.to(torch.uint8).contiguous()
    return ops.attention(query, keys_cl, values_cl, m, 2, scale)


class DotProductAttention(nn.Module):
    """Single-head attention of the waypoint net (reference utils.py:155-178): Q [B, Dk],
    K [B, Dk, P], V [B, Dv, P] in the reference's logical layouts."""

    def __init__(self, key_dimension):
        super().__init__()
        self._scale_f = float(key_dimension) ** -0.5
        self.scale = torch.tensor(self._scale_f)

    def forward(self, Q, K, V, mask=None):
        return _attend(Q, K.transpose(1, 2), V.transpose(1, 2), mask, self._scale_f)


class MultiHeadDotProductAttention(nn.Module):
    """Projected multi-head attention with an optional LayerNorm on the output (reference
    utils.py:181-266).  Heads are folded into the batch dimension of the HIP attention kernel."""

    def __init__(self, d_q_in, d_k_in, d_v_in, d_qk, d_v, num_heads, d_out, normalize=True,
                 dropout_p=0.0):
        super().__init__()
        if dropout_p != 0.0:
            raise AssertionError("the reference never enables attention dropout")
        self.num_heads = num_heads
        self.normalize = normalize
        self.dropout = None
        self.q_linear = nn.Linear(d_q_in, d_qk * num_heads, bias=False)
        self.k_linear = nn.Linear(d_k_in, d_qk * num_heads, bias=False)
        self.v_linear = nn.Linear(d_v_in, d_v * num_heads, bias=False)
        self.attn = DotProductAttention(d_qk)
        self.final_linear = nn.Linear(d_v * num_heads, d_out, bias=False)
        if normalize:
            self.layer_norm = nn.LayerNorm(d_out, eps=1e-6)

    def _heads_to_batch(self, t):
        """[B, P, heads*d] -> [B*heads, P, d]"""
        B, P, wide = t.shape
        d = wide // self.num_heads
        return t.view(B, P, self.num_heads, d).transpose(1, 2).reshape(B * self.num_heads, P, d)

    def forward(self, Q, K, V, mask=None):
        """Q [B, d_q_in]; K [B, d_k_in, P]; V [B, d_v_in, P]."""
        if K.shape[2] != V.shape[2]:
            raise AssertionError("keys must be the same size as values")
        B = Q.shape[0]
        query = ops.linear(Q, self.q_linear.weight)
        keys = ops.linear(K.transpose(1, 2), self.k_linear.weight)
        values = ops.linear(V.transpose(1, 2), self.v_linear.weight)
        scale = self.attn._scale_f
        if self.num_heads == 1:
            mixed = _attend(query, keys, values, mask, scale)
        else:
            per_head = _attend(query.view(B * self.num_heads, -1), self._heads_to_batch(keys),
                               self._heads_to_batch(values), mask, scale)
            mixed = per_head.view(B, -1)
        out = ops.linear(mixed, self.final_linear.weight)
        if self.normalize:
            ln = self.layer_norm
            out = F.layer_norm(out, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
        return out


class CustomFixedCategorical(torch.distributions.Categorical):
    """Categorical whose samples, modes and log-probabilities keep a trailing unit dimension
    (reference utils.py:269-289; `log_probs` is the habitat-lab spelling)."""

    @classmethod
    def from_normalized(cls, logits):
        """The distribution over ALREADY normalised logits (z - logsumexp(z), what
        torch.distributions.Categorical.__init__ computes first): no launches, no host sync.  The
        caller has validated the parameter (policy.CategoricalNet); sample / log_prob arguments are
        validated as torch's default says."""
        self = cls.__new__(cls)
        self.logits = logits
        self._param = logits
        self._num_events = logits.size(-1)
        batch_shape = logits.shape[:-1] if logits.dim() > 1 else torch.Size()
        torch.distributions.Distribution.__init__(self, batch_shape, validate_args=False)
        self._validate_args = torch.distributions.Distribution._validate_args
        return self

    def sample(self, sample_shape=torch.Size()):
        return super().sample(sample_shape)[..., None]

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)

    def log_prob(self, actions):
        flat = super().log_prob(actions.squeeze(-1))
        return flat.reshape(actions.size(0), -1).sum(dim=-1, keepdim=True)

    def log_probs(self, actions):
        return self.log_prob(actions)


def batched_index_select(x, dim, index):
    """x[b].index_select(dim - 1, index[b]) for every batch row b, with `dim` squeezed away
    (reference utils.py:292-317): x [B, ..., K(dim), ...], index [B] or [B, 1]."""
    shape = [x.shape[0]] + [1] * (x.dim() - 1)
    picker = index.reshape(shape).expand(*[-1 if i in (0, dim) else s for i, s in enumerate(x.shape)])
    return torch.take_along_dim(x, picker, dim).squeeze(dim)
