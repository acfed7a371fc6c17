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


def wddppo_minibatch_update(policy, optimizer, sample, cfg=PPOConfig(), *, step_grad=True,
                            clip_grads=True, grad_hook=None):
    (obs, h0, actions, prev_actions, value_preds, returns, masks, old_logp, adv) = sample
    values, logp, entropy, _ = policy.evaluate_actions(obs, h0, prev_actions, masks, actions)

    weighted_entropy = (cfg.pano_entropy_coef * entropy["pano"]
                        + cfg.offset_entropy_coef * entropy["offset"]
                        + cfg.distance_entropy_coef * entropy["distance"])
    entropy_loss = weighted_entropy.mean() * cfg.entropy_coef

    ratio = (logp - old_logp).exp()
    lo, hi = 1.0 - cfg.clip_param, 1.0 + cfg.clip_param
    action_loss = -torch.minimum(ratio * adv, ratio.clamp(lo, hi) * adv).mean()

    err = (values - returns).square()
    if cfg.use_clipped_value_loss:
        near = value_preds + (values - value_preds).clamp(-cfg.clip_param, cfg.clip_param)
        err = torch.maximum(err, (near - returns).square())
    value_loss = 0.5 * err.mean() * cfg.value_loss_coef

    loss = value_loss + action_loss - entropy_loss
    if "offset" in actions:  # keep predicted headings near the pano centre
        radians = policy.net.offset_to_continuous(actions["offset"])
        loss = loss + cfg.offset_regularize_coef * radians.abs().mean()

    if optimizer is not None:
        optimizer.zero_grad()
    loss.backward()
    if grad_hook is not None:
        grad_hook()  # data-parallel gradient all-reduce (vlnce_amd.distributed)
    if clip_grads and cfg.max_grad_norm is not None:
        torch.nn.utils.clip_grad_norm_(policy.parameters(), cfg.max_grad_norm)
    if step_grad and optimizer is not None:
        optimizer.step()
    return tuple(v.detach() for v in (
        value_loss, action_loss, entropy_loss, entropy["pano"].mean(),
        entropy["offset"].mean(), entropy["distance"].mean()))
