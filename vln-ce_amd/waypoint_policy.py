"""Waypoint actor-critic policy: one categorical choice over the 12 panorama headings + STOP and,
for the chosen heading, a distance and a heading-offset component (continuous: truncated normal,
discrete: categorical), with a composite log-probability and per-component entropies.

Plugin-surface mirror of vlnce_baselines/models/waypoint_policy.py:19-347 (`from_config`, `act`,
`get_value`, `evaluate_actions` and their return tuples are what ddppo_waypoint_trainer.py and
WDDPPO call).  The two sub-actions are handled by ONE table-driven code path here instead of a
pair of parallel methods each; the distribution math is O(N x 13) scalars and stays in torch
(SURVEY.md 2.1), the net runs on the HIP kernels.
"""
import math
from typing import NamedTuple

import torch

from .policy import Policy
from .registry import baseline_registry
from .utils import CustomFixedCategorical, TruncatedNormal, batched_index_select
from .waypoint_predictors import WaypointPredictionNet

TWO_PI = 2.0 * math.pi


class _Draw(NamedTuple):
    """one sub-action of one act() call"""
    element: torch.Tensor   # what is stored in the rollout (network units / class index)
    metric: torch.Tensor    # metres or radians handed to the simulator
    log_prob: torch.Tensor
    variance: torch.Tensor
    mode: torch.Tensor


@baseline_registry.register_policy
class WaypointPolicy(Policy):
    # (config flag "continuous_*", config flag "predict_*", value the simulator gets when the
    # component is switched off, value stored in the rollout when switched off and continuous)
    _PARTS = {
        "distance": ("continuous_distance", "predict_distance", 0.25, 0.25),
        "offset": ("continuous_offset", "predict_offset", 0.0, 0.0),
    }

    def __init__(self, observation_space, action_space, model_config):
        net = WaypointPredictionNet(observation_space=observation_space, model_config=model_config)
        super().__init__(net, 1)  # the inherited 1-way action head is never used (App. B-6)
        self._config = model_config
        self.wypt_cfg = model_config.WAYPOINT
        self._offset_limit = math.pi / model_config.num_panos

    @classmethod
    def from_config(cls, config, observation_space, action_space):
        config.defrost()
        config.MODEL.num_panos = config.TASK_CONFIG.TASK.PANO_ROTATIONS
        config.freeze()
        return cls(observation_space=observation_space, action_space=action_space,
                   model_config=config.MODEL)

    # ------------------------------------------------------------------ sub-action plumbing
    def _bounds(self, part):
        if part == "distance":
            return self.wypt_cfg.min_distance_prediction, self.wypt_cfg.max_distance_prediction
        return -self._offset_limit, self._offset_limit

    def _to_metric(self, part, element):
        fn = self.net.distance_to_continuous if part == "distance" else self.net.offset_to_continuous
        return fn(element)

    def _distribution(self, part, first, second, heading):
        """distribution of `part` for the chosen heading: `first`/`second` are the per-heading
        (mean, variance) maps of a continuous head or the per-heading logits of a discrete one."""
        if getattr(self.wypt_cfg, self._PARTS[part][0]):
            lo, hi = self._bounds(part)
            return TruncatedNormal(loc=first.gather(1, heading),
                                   scale=second.gather(1, heading).sqrt(), smin=lo, smax=hi)
        return CustomFixedCategorical(logits=batched_index_select(first, dim=1, index=heading))

    def _draw(self, part, dist, deterministic):
        element = dist.mode() if deterministic else dist.sample()
        out = _Draw(element, self._to_metric(part, element), dist.log_prob(element), dist.variance,
                    dist.mode())
        _, enabled_flag, off_metric, off_element = self._PARTS[part]
        if getattr(self.wypt_cfg, enabled_flag):
            return out
        # component switched off: constants go to the simulator and into the rollout.  A class
        # index is stored as 0 (upstream scales a zero tensor by discrete_offsets // 2, which
        # leaves it zero: waypoint_policy.py:108-110), a continuous element as the constant.
        stored = torch.full_like(element, 0 if element.dtype == torch.int64 else off_element)
        return out._replace(element=stored, metric=torch.full_like(out.metric, off_metric),
                            variance=torch.zeros_like(out.variance))

    @staticmethod
    def _simulator_actions(stop, radius, theta):
        # one batched D2H copy instead of three .item() syncs per environment
        rows = torch.cat([stop.float(), radius.float(), theta.float()], dim=1).tolist()
        return [{"action": "STOP"} if s else
                {"action": {"action": "GO_TOWARD_POINT", "action_args": {"r": r, "theta": th}}}
                for s, r, th in rows]

    # ------------------------------------------------------------------ plugin surface
    def act(self, observations, rnn_states, prev_actions, masks, deterministic=False):
        n_pano = self._config.num_panos
        (heading_dist, off_a, off_b, dist_a, dist_b, features, rnn_states_out) = self.net(
            observations, rnn_states, prev_actions, masks)
        choice = heading_dist.mode() if deterministic else heading_dist.sample()
        heading = choice % n_pano
        going = choice != n_pano
        draws = {
            "distance": self._draw("distance", self._distribution("distance", dist_a, dist_b, heading),
                                   deterministic),
            "offset": self._draw("offset", self._distribution("offset", off_a, off_b, heading),
                                 deterministic),
        }
        theta = (heading * (TWO_PI / n_pano) + draws["offset"].metric) % TWO_PI
        actions = self._simulator_actions((~going).to(torch.uint8), draws["distance"].metric, theta)

        log_prob = heading_dist.log_prob(choice)
        gate = going.to(log_prob.dtype)
        for part, d in draws.items():
            enabled = getattr(self.wypt_cfg, self._PARTS[part][1])
            if enabled:
                log_prob = log_prob + gate * enabled * d.log_prob
        elements = {"pano": choice, "offset": draws["offset"].element,
                    "distance": draws["distance"].element}
        modes = {part: d.mode for part, d in draws.items()}
        variances = {part: d.variance for part, d in draws.items()}
        return (self.critic(features), actions, elements, modes, variances, log_prob,
                rnn_states_out, heading_dist)

    def get_value(self, observations, rnn_states, prev_actions, masks):
        features = self.net(observations, rnn_states, prev_actions, masks)[5]
        return self.critic(features)

    def evaluate_actions(self, observations, rnn_states, prev_actions, masks, action_components):
        n_pano = self._config.num_panos
        (heading_dist, off_a, off_b, dist_a, dist_b, features, rnn_states_out) = self.net(
            observations, rnn_states, prev_actions, masks)
        choice = action_components["pano"]
        heading = choice.to(torch.int64) % n_pano
        log_prob = heading_dist.log_prob(choice)
        gate = (choice != n_pano).to(log_prob.dtype)
        entropy = {"pano": heading_dist.entropy()}
        for part, (first, second) in (("distance", (dist_a, dist_b)), ("offset", (off_a, off_b))):
            dist = self._distribution(part, first, second, heading)
            weight = gate * getattr(self.wypt_cfg, self._PARTS[part][1])
            log_prob = log_prob + weight * dist.log_prob(action_components[part])
            entropy[part] = (weight * dist.entropy()).squeeze(1)
        return self.critic(features), log_prob, entropy, rnn_states_out
