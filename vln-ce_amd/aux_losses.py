"""Global auxiliary-loss registry filled as a side effect of Net.forward
(reference: vlnce_baselines/common/aux_losses.py:4-44)."""
import torch


class _AuxLosses:
    def __init__(self):
        self._losses = {}
        self._alphas = {}
        self._active = False

    def clear(self):
        self._losses.clear()
        self._alphas.clear()

    def register_loss(self, name, loss, alpha=1.0):
        assert self.is_active()
        assert name not in self._losses
        self._losses[name] = loss
        self._alphas[name] = alpha

    def get_loss(self, name):
        return self._losses[name]

    def reduce(self, mask):
        assert self.is_active()
        total = 0.0
        for k, v in self._losses.items():
            total = total + self._alphas[k] * torch.masked_select(v, mask).mean()
        return total

    def is_active(self):
        return self._active

    def activate(self):
        self._active = True

    def deactivate(self):
        self._active = False


AuxLosses = _AuxLosses()
