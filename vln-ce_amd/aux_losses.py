"""Auxiliary losses collected as a side effect of a policy forward.

Plugin-surface mirror of vlnce_baselines/common/aux_losses.py:4-44: the nets call
`AuxLosses.register_loss(name, per-row loss, alpha)` while the registry is active
(progress monitor, cma_policy.py:296-307 / seq2seq_policy.py:166-177) and the trainer
folds them into the objective with `AuxLosses.reduce(mask)` (base_il_trainer.py:163).
The reference keeps two parallel dicts; here one ordered table of (tensor, weight) entries,
a context manager for the activation window, and explicit errors instead of bare asserts.
"""
from collections import OrderedDict
from contextlib import contextmanager
from typing import NamedTuple

import torch


class _Entry(NamedTuple):
    values: torch.Tensor  # one loss value per batch row
    weight: float


class AuxLossRegistry:
    def __init__(self):
        self._table = OrderedDict()
        self._collecting = False
        self._dp = None  # (process group | None, world size) while data parallelism is declared

    # -- data parallelism ----------------------------------------------------------------
    def set_data_parallel(self, enabled=True, group=None):
        """Upstream reduces an auxiliary loss with a GLOBAL masked mean over the batch
        (aux_losses.py:24-32).  With the batch sharded over ranks and gradients averaged, the
        mean of per-rank masked means equals it only when every rank has the same number of
        unmasked rows.  Declaring data parallelism makes reduce() return
        world * local_sum / global_count (one tiny all-reduce of the counts), whose rank-average
        -- hence the averaged gradient -- is exactly the global masked mean.  Nets that build a
        loss from per-row targets of the WHOLE batch (the progress monitor's [B] x [B,1]
        broadcast, App. B-2) gather those targets through `gather_rows`."""
        if not enabled:
            self._dp = None
            return
        import torch.distributed as dist

        self._dp = (group, dist.get_world_size(group))

    def data_parallel(self):
        return self._dp is not None

    def gather_rows(self, t):
        """[b, ...] rows of this rank -> [B_global, ...] rows of all ranks (no gradient)."""
        if self._dp is None:
            return t
        import torch.distributed as dist

        parts = [torch.empty_like(t) for _ in range(self._dp[1])]
        dist.all_gather(parts, t.detach().contiguous(), group=self._dp[0])
        return torch.cat(parts, dim=0)

    # -- activation window -------------------------------------------------------------
    def activate(self):
        self._collecting = True

    def deactivate(self):
        self._collecting = False

    def is_active(self):
        return self._collecting

    @contextmanager
    def active(self):
        """`with AuxLosses.active(): ...` -- activate, and restore the previous state after."""
        was = self._collecting
        self._collecting = True
        try:
            yield self
        finally:
            self._collecting = was

    # -- table ---------------------------------------------------------------------------
    def clear(self):
        self._table = OrderedDict()

    def register_loss(self, name, loss, alpha=1.0):
        if not self._collecting:
            raise AssertionError("AuxLosses.register_loss() outside an activation window")
        if name in self._table:
            raise AssertionError(f"auxiliary loss '{name}' registered twice in one forward")
        self._table[name] = _Entry(loss, alpha)

    def get_loss(self, name):
        return self._table[name].values

    def names(self):
        return list(self._table)

    def reduce(self, mask):
        """sum_k alpha_k * mean(loss_k[mask]) -- a Python float 0.0 when nothing was registered,
        like upstream."""
        if not self._collecting:
            raise AssertionError("AuxLosses.reduce() outside an activation window")
        if self._dp is None:
            terms = [entry.weight * entry.values.masked_select(mask).mean()
                     for entry in self._table.values()]
            return sum(terms, 0.0)
        import torch.distributed as dist

        group, world = self._dp
        terms = []
        for entry in self._table.values():
            picked = entry.values.masked_select(mask)
            count = torch.tensor([float(picked.numel())], device=picked.device)
            dist.all_reduce(count, group=group)  # selected elements over all ranks
            terms.append(entry.weight * picked.sum() * (world / count[0]))
        return sum(terms, 0.0)


AuxLosses = AuxLossRegistry()
