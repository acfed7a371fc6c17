"""DAgger / recollect inner step: restatement of
BaseVLNCETrainer._update_agent (vlnce_baselines/common/base_il_trainer.py:134-180)
for callers that run without habitat (bench.py, tests).  The real trainers call
the policy's build_distribution() themselves and need nothing from here."""
import torch
import torch.nn.functional as F

from .aux_losses import AuxLosses


def update_agent(policy, optimizer, observations, prev_actions, not_done_masks,
                 corrected_actions, weights, hidden_size, step_grad=True,
                 loss_accumulation_scalar=1, grad_hook=None):
    T, N = corrected_actions.size()
    recurrent_hidden_states = torch.zeros(
        N, policy.net.num_recurrent_layers, hidden_size, device=corrected_actions.device)
    AuxLosses.clear()
    distribution = policy.build_distribution(
        observations, recurrent_hidden_states, prev_actions, not_done_masks)
    logits = distribution.logits.view(T, N, -1)
    action_loss = F.cross_entropy(logits.permute(0, 2, 1), corrected_actions, reduction="none")
    action_loss = ((weights * action_loss).sum(0) / weights.sum(0)).mean()
    aux_loss = AuxLosses.reduce((weights > 0).view(-1)) if AuxLosses.is_active() else 0.0
    loss = (action_loss + aux_loss) / loss_accumulation_scalar
    loss.backward()
    if grad_hook is not None:
        grad_hook()  # data-parallel gradient all-reduce (vlnce_amd.distributed)
    if step_grad:
        optimizer.step()
        optimizer.zero_grad()
    # the reference ends every step with three host read-backs (base_il_trainer.py:176-180):
    # `loss.item(), action_loss.item(), aux_loss.item()` -- Python floats, one sync each; the timed
    # loop of bench.py consumes them exactly as the trainers' loggers do
    if isinstance(aux_loss, torch.Tensor):
        aux_loss = aux_loss.item()
    return loss.item(), action_loss.item(), aux_loss
