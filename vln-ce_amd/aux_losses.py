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
        terms = [entry.weight * entry.values.masked_select(mask).mean()
                 for entry in self._table.values()]
        return sum(terms, 0.0)


AuxLosses = AuxLossRegistry()
