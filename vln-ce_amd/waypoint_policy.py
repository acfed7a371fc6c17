"""Waypoint actor-critic policy (reference: vlnce_baselines/models/
waypoint_policy.py:19-347): pano / offset / distance action components with a
composite log-probability and entropy.  The distribution math is O(N x 13)
scalars and stays in torch; the net runs on the HIP kernels."""
import numpy as np
import torch

from .policy import Policy
from .registry import baseline_registry
from .utils import CustomFixedCategorical, TruncatedNormal, batched_index_select
from .waypoint_predictors import WaypointPredictionNet


@baseline_registry.register_policy
class WaypointPolicy(Policy):
    def __init__(self, observation_space, action_space, model_config):
        super().__init__(
            WaypointPredictionNet(observation_space=observation_space, model_config=model_config),
            1,  # the inherited 1-way action_distribution is never used (App. B-6)
        )
        self._config = model_config
        self.wypt_cfg = model_config.WAYPOINT
        self._offset_limit = np.pi / self._config.num_panos

    @classmethod
    def from_config(cls, config, observation_space, action_space):
        config.defrost()
        config.MODEL.num_panos = config.TASK_CONFIG.TASK.PANO_ROTATIONS
        config.freeze()
        return cls(observation_space=observation_space, action_space=action_space,
                   model_config=config.MODEL)

    def _create_distance_distribution(self, var1, var2, pano):
        if self.wypt_cfg.continuous_distance:
            return TruncatedNormal(
                loc=torch.gather(var1, dim=1, index=pano),
                scale=torch.sqrt(torch.gather(var2, dim=1, index=pano)),
                smin=self.wypt_cfg.min_distance_prediction,
                smax=self.wypt_cfg.max_distance_prediction)
        return CustomFixedCategorical(logits=batched_index_select(var1, dim=1, index=pano))

    def _create_offset_distribution(self, var1, var2, pano):
        if self.wypt_cfg.continuous_offset:
            return TruncatedNormal(
                loc=torch.gather(var1, dim=1, index=pano),
                scale=torch.sqrt(torch.gather(var2, dim=1, index=pano)),
                smin=-self._offset_limit, smax=self._offset_limit)
        return CustomFixedCategorical(logits=batched_index_select(var1, dim=1, index=pano))

    def get_offset_prediction(self, offset_distribution, deterministic=False):
        offset = offset_distribution.mode() if deterministic else offset_distribution.sample()
        offset_log_prob = offset_distribution.log_prob(offset)
        action_offset = self.net.offset_to_continuous(offset)
        variance = offset_distribution.variance
        mode = offset_distribution.mode()
        if not self.wypt_cfg.predict_offset:
            action_offset = torch.zeros_like(action_offset)
            offset = torch.zeros_like(offset)
            if offset.dtype == torch.int64:
                offset *= self.wypt_cfg.discrete_offsets // 2
            variance = torch.zeros_like(variance)
        return offset, action_offset, offset_log_prob, variance, mode

    def get_distance_prediction(self, distance_distribution, deterministic=False):
        distance = distance_distribution.mode() if deterministic else distance_distribution.sample()
        distance_log_prob = distance_distribution.log_prob(distance)
        action_distance = self.net.distance_to_continuous(distance)
        variance = distance_distribution.variance
        mode = distance_distribution.mode()
        if not self.wypt_cfg.predict_distance:
            action_distance = torch.zeros_like(action_distance) + 0.25
            distance = torch.zeros_like(distance)
            if distance.dtype != torch.int64:
                distance = torch.zeros_like(distance) + 0.25
            variance = torch.zeros_like(variance)
        return distance, action_distance, distance_log_prob, variance, mode

    def act(self, observations, rnn_states, prev_actions, masks, deterministic=False):
        P = self._config.num_panos
        (pano_stop_distribution, offset_variable1, offset_variable2, distance_variable1,
         distance_variable2, x, rnn_states_out) = self.net(observations, rnn_states,
                                                           prev_actions, masks)
        pano_stop = (pano_stop_distribution.mode() if deterministic
                     else pano_stop_distribution.sample())
        stop = (pano_stop == P).to(torch.uint8)
        pano = pano_stop % P
        distance_distribution = self._create_distance_distribution(
            distance_variable1, distance_variable2, pano)
        offset_distribution = self._create_offset_distribution(
            offset_variable1, offset_variable2, pano)
        (distance, action_distance, distance_log_probs, dist_var,
         dist_mode) = self.get_distance_prediction(distance_distribution, deterministic)
        (offset, action_offset, offset_log_probs, ofst_var,
         ofst_mode) = self.get_offset_prediction(offset_distribution, deterministic)

        radians_per_pano = 2 * np.pi / P
        theta = (pano * radians_per_pano + action_offset) % (2 * np.pi)
        # one batched D2H instead of 3 .item() syncs per env (waypoint_policy.py:191-208)
        host = torch.cat([stop.float(), action_distance.float(), theta.float()], dim=1).tolist()
        actions = []
        for s, r, th in host:
            if s:
                actions.append({"action": "STOP"})
            else:
                actions.append({"action": {"action": "GO_TOWARD_POINT",
                                           "action_args": {"r": r, "theta": th}}})

        action_log_probs = pano_stop_distribution.log_prob(pano_stop)
        pano_mask = (pano_stop != P).to(action_log_probs.dtype)
        if self.wypt_cfg.predict_distance:
            action_log_probs = action_log_probs + (
                pano_mask * self.wypt_cfg.predict_distance * distance_log_probs)
        if self.wypt_cfg.predict_offset:
            action_log_probs = action_log_probs + (
                pano_mask * self.wypt_cfg.predict_offset * offset_log_probs)
        value = self.critic(x)
        action_elements = {"pano": pano_stop, "offset": offset, "distance": distance}
        variances = {"distance": dist_var, "offset": ofst_var}
        modes = {"offset": ofst_mode, "distance": dist_mode}
        return (value, actions, action_elements, modes, variances, action_log_probs,
                rnn_states_out, pano_stop_distribution)

    def get_value(self, observations, rnn_states, prev_actions, masks):
        return self.critic(self.net(observations, rnn_states, prev_actions, masks)[5])

    def evaluate_actions(self, observations, rnn_states, prev_actions, masks, action_components):
        P = self._config.num_panos
        (pano_stop_distribution, offset_variable1, offset_variable2, distance_variable1,
         distance_variable2, x, rnn_states_out) = self.net(observations, rnn_states,
                                                           prev_actions, masks)
        value = self.critic(x)
        pano_log_probs = pano_stop_distribution.log_prob(action_components["pano"])
        idx = action_components["pano"].to(torch.int64) % P
        distance_distribution = self._create_distance_distribution(
            distance_variable1, distance_variable2, idx)
        offset_distribution = self._create_offset_distribution(
            offset_variable1, offset_variable2, idx)
        pano_mask = (action_components["pano"] != P).to(pano_log_probs.dtype)
        d_mask = pano_mask * self.wypt_cfg.predict_distance
        o_mask = pano_mask * self.wypt_cfg.predict_offset
        distance_log_probs = d_mask * distance_distribution.log_prob(action_components["distance"])
        offset_log_probs = o_mask * offset_distribution.log_prob(action_components["offset"])
        action_log_probs = pano_log_probs + distance_log_probs + offset_log_probs
        entropy = {
            "pano": pano_stop_distribution.entropy(),
            "offset": (o_mask * offset_distribution.entropy()).squeeze(1),
            "distance": (d_mask * distance_distribution.entropy()).squeeze(1),
        }
        return value, action_log_probs, entropy, rnn_states_out
