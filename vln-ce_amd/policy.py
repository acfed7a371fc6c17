"""Policy base classes of the plugin surface (reference: models/policy.py:10-58
and habitat-lab v0.1.7 rl/ppo/policy.py + utils/common.py)."""
import abc

import torch.nn as nn

from . import ops
from .utils import CustomFixedCategorical


class Net(nn.Module, metaclass=abc.ABCMeta):
    pass


class CategoricalNet(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.linear = nn.Linear(num_inputs, num_outputs)
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)

    def forward(self, x):
        return CustomFixedCategorical(logits=ops.linear(x, self.linear.weight, self.linear.bias))


class CriticHead(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.fc = nn.Linear(input_size, 1)
        nn.init.orthogonal_(self.fc.weight)
        nn.init.constant_(self.fc.bias, 0)

    def forward(self, x):
        return ops.linear(x, self.fc.weight, self.fc.bias)


class Policy(nn.Module):
    """habitat actor-critic base: net + action_distribution + critic."""

    def __init__(self, net, dim_actions):
        super().__init__()
        self.net = net
        self.dim_actions = dim_actions
        self.action_distribution = CategoricalNet(self.net.output_size, self.dim_actions)
        self.critic = CriticHead(self.net.output_size)

    def forward(self, *x):
        raise NotImplementedError


class ILPolicy(Policy):
    """Imitation-learning policy: act() + build_distribution(); deliberately
    skips Policy.__init__ so there is no critic (policy.py:15, App. B-6)."""

    def __init__(self, net, dim_actions):
        nn.Module.__init__(self)
        self.net = net
        self.dim_actions = dim_actions
        self.action_distribution = CategoricalNet(self.net.output_size, self.dim_actions)

    def act(self, observations, rnn_states, prev_actions, masks, deterministic=False):
        features, rnn_states = self.net(observations, rnn_states, prev_actions, masks)
        distribution = self.action_distribution(features)
        action = distribution.mode() if deterministic else distribution.sample()
        return action, rnn_states

    def get_value(self, *args, **kwargs):
        raise NotImplementedError

    def evaluate_actions(self, *args, **kwargs):
        raise NotImplementedError

    def build_distribution(self, observations, rnn_states, prev_actions, masks):
        features, rnn_states = self.net(observations, rnn_states, prev_actions, masks)
        return self.action_distribution(features)

    def encode_ahead(self, observations):
        """Optional (not in the reference): start the frozen visual trunks of a FUTURE batch on
        side HIP streams now and get back the observation dict to pass to act() /
        build_distribution() later (it carries `rgb_features` / `depth_features`).  Issued
        before the update on the current batch, the next batch's trunks overlap that update's
        tail; results are identical to calling without it.  See streams.BranchStreams."""
        branches = getattr(self.net, "_branches", None)
        if branches is None:
            return observations
        return branches.encode_visual_ahead(self.net, observations)
