"""Waypoint DD-PPO inner step: one minibatch of `WDDPPO.update`
(vlnce_baselines/common/ddppo_alg.py:53-141) for callers that run without habitat
(tests, benchmarks).  `ddppo_waypoint_trainer.py` drives the same math through
habitat's DDPPO class and only needs the policy's evaluate_actions() from this package.

`sample` is the 9-tuple `RolloutStorage.recurrent_generator` yields:
(observations, recurrent_hidden_states, actions{pano,offset,distance}, prev_actions,
 value_preds, returns, masks, old_action_log_probs, advantages), rows time-major (t*N + n).
"""
from dataclasses import dataclass

import torch


@dataclass
class PPOConfig:  # RL.PPO defaults, vlnce_baselines/config/default.py:180-201
    clip_param: float = 0.2
    value_loss_coef: float = 0.5
    entropy_coef: float = 0.01
    pano_entropy_coef: float = 1.0
    offset_entropy_coef: float = 0.0
    distance_entropy_coef: float = 0.0
    offset_regularize_coef: float = 0.1146
    use_clipped_value_loss: bool = True
    max_grad_norm: float = 0.2


def normalized_advantages(returns, value_preds, normalize=False, eps=1e-5):
    """WDDPPO.get_advantages (ddppo_alg.py:31-36); inputs carry the bootstrap row."""
    adv = returns[:-1] - value_preds[:-1]
    return (adv - adv.mean()) / (adv.std() + eps) if normalize else adv


def compute_returns(rewards, value_preds, masks, next_value, gamma, tau, use_gae=True):
    """RolloutStorage.compute_returns (rollout_storage.py:127-152) on the device, one launch.
    rewards [T,N,1]; value_preds, masks [T+1,N,1] (value_preds[T] is overwritten with next_value
    under GAE, as upstream); next_value [N,1].  Returns `returns` [T+1,N,1]."""
    from . import ops

    T, N = rewards.shape[0], rewards.shape[1]
    rewards, masks = rewards.contiguous().float(), masks.contiguous().float()
    assert value_preds.is_contiguous() and value_preds.dtype == torch.float32
    returns = torch.zeros_like(value_preds)
    ops.L().ppo_returns(rewards, value_preds, masks, next_value.contiguous().float(), returns, T, N,
                        gamma, tau, use_gae)
    return returns


class PPOLossFn(torch.autograd.Function):
    """The WDDPPO minibatch loss (ddppo_alg.py:78-121) as ONE launch that also produces every
    gradient (vlnce_ppo_loss), instead of ~40 elementwise / reduction launches forward and as many
    backward, all host-paced.  forward(values, logp, ent_pano, ent_offset, ent_distance,
    value_preds, returns, old_logp, adv, radians | None, cfg) -> (loss [], stats [8] = loss,
    value_loss, action_loss, entropy_loss, mean pano / offset / distance entropy, offset_loss)."""

    @staticmethod
    def forward(ctx, values, logp, ent_p, ent_o, ent_d, value_preds, returns, old_logp, adv, radians,
                cfg):
        from . import ops

        B = values.numel()
        flat = [ops._f32c(t.detach()).reshape(-1) if t is not None else None
                for t in (values, returns, value_preds, logp, old_logp, adv, ent_p, ent_o, ent_d,
                          radians)]
        assert all(t is None or t.numel() == B for t in flat), "one value per rollout row"
        stats = torch.empty(8, device=values.device, dtype=torch.float32)
        grads = torch.empty((5, B), device=values.device, dtype=torch.float32)
        ops.L().ppo_loss(*flat, B, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef,
                         cfg.pano_entropy_coef, cfg.offset_entropy_coef, cfg.distance_entropy_coef,
                         cfg.offset_regularize_coef, cfg.use_clipped_value_loss, stats, grads)
        ctx.save_for_backward(grads)
        ctx.shapes = tuple(t.shape for t in (values, logp, ent_p, ent_o, ent_d))
        loss = stats[0].clone()
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        (grads,) = ctx.saved_tensors
        g = grads * g_loss
        out = [g[k].view(shape) for k, shape in enumerate(ctx.shapes)]
        return (*out, None, None, None, None, None, None)


def wddppo_minibatch_update(policy, optimizer, sample, cfg=PPOConfig(), *, step_grad=True,
                            clip_grads=True, grad_hook=None):
    (obs, h0, actions, prev_actions, value_preds, returns, masks, old_logp, adv) = sample
    values, logp, entropy, _ = policy.evaluate_actions(obs, h0, prev_actions, masks, actions)
    # keep predicted headings near the pano centre (a constant: sampled offsets carry no gradient)
    radians = policy.net.offset_to_continuous(actions["offset"]) if "offset" in actions else None
    loss, stats = PPOLossFn.apply(values, logp, entropy["pano"], entropy["offset"],
                                  entropy["distance"], value_preds, returns, old_logp, adv, radians,
                                  cfg)

    if optimizer is not None:
        optimizer.zero_grad()
    loss.backward()
    if grad_hook is not None:
        grad_hook()  # data-parallel gradient all-reduce (vlnce_amd.distributed)
    if clip_grads and cfg.max_grad_norm is not None:
        torch.nn.utils.clip_grad_norm_(policy.parameters(), cfg.max_grad_norm)
    if step_grad and optimizer is not None:
        optimizer.step()
    # (value_loss, action_loss, entropy_loss, pano / offset / distance entropy) as WDDPPO.update sums
    return tuple(stats[1:7].unbind(0))
