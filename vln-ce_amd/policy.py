"""Policy base classes of the plugin surface (reference: models/policy.py:10-58
and habitat-lab v0.1.7 rl/ppo/policy.py + utils/common.py)."""
import abc
import os

import torch
import torch.nn as nn

from . import ops
from .streams import ActGraph, DropsGraphsOnApply
from .utils import CustomFixedCategorical


class Net(DropsGraphsOnApply, nn.Module, metaclass=abc.ABCMeta):
    pass


class CategoricalNet(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.linear = nn.Linear(num_inputs, num_outputs)
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)

    def forward(self, x):
        """Categorical(logits=Linear(x)) as the reference builds it (models/policy.py:19-21,
        utils.py:269-289) with the linear layer, the normalisation `z - logsumexp(z)` and the NaN
        test of the argument validation in ONE launch (ops.action_head; ~16 host-paced launches as
        torch ops), and the whole backward in one more."""
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
        # (the one-launch head holds a row's logits in registers: action spaces of up to 16 classes;
        # a larger discrete space takes the plain linear layer + Categorical)
        if (os.environ.get("VLNCE_ACTION_HEAD", "1") == "0"   # A/B: the head as separate torch ops
                or self.linear.weight.size(0) > 16):
            logits = ops.linear(x, self.linear.weight, self.linear.bias)
            return CustomFixedCategorical(logits=logits, validate_args=False if capturing else None)
        validate = torch.distributions.Distribution._validate_args and not capturing
        # (argument validation is a host sync: not inside a graph capture, streams.ActGraph)
        logits, nans = ops.action_head(x, self.linear.weight, self.linear.bias, count_nans=validate)
        # The validation's host read-back stays where the reference has it.  Measured
        # (profiles/archive/r04_i_sync_probe.txt, r04_j_*): WITHOUT it a training step is 1.3-1.7 ms SLOWER
        # -- the host then enqueues loss / backward / Adam while the trunks are still running, and
        # the forward phase takes 9.7 instead of 8.1 ms on the GPU's own clock (launches arriving
        # on the queue of a running graph slow its kernel-to-kernel dispatch).
        if validate and int(nans.item()) != 0:
            nans.zero_()
            raise ValueError(
                f"Expected parameter logits (Tensor of shape {tuple(logits.shape)}) of distribution "
                "CustomFixedCategorical to satisfy the constraint IndependentConstraint(Real(), 1), "
                f"but found invalid values:\n{logits}")
        return CustomFixedCategorical.from_normalized(logits)


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


def _not_part_of_imitation_learning(self, *args, **kwargs):
    raise NotImplementedError


class ILPolicy(DropsGraphsOnApply, Policy):
    """Imitation-learning policy: net + categorical action head, NO critic -- Policy.__init__ is
    deliberately bypassed as upstream (models/policy.py:10-23, App. B-6).  `act()` and
    `build_distribution()` are what the DAgger / recollect trainers call."""

    def __init__(self, net, dim_actions):
        nn.Module.__init__(self)
        self.net, self.dim_actions = net, dim_actions
        self.action_distribution = CategoricalNet(net.output_size, dim_actions)

    @classmethod
    def from_config(cls, config, observation_space, action_space):
        """baseline_registry entry point (base_il_trainer.py:61-66): MODEL gets the GPU id the
        reference's nets read from it."""
        config.defrost()
        config.MODEL.TORCH_GPU_ID = config.TORCH_GPU_ID
        config.freeze()
        return cls(observation_space=observation_space, action_space=action_space,
                   model_config=config.MODEL)

    def _step(self, observations, rnn_states, prev_actions, masks):
        features, new_states = self.net(observations, rnn_states, prev_actions, masks)
        return self.action_distribution(features), new_states

    def build_distribution(self, observations, rnn_states, prev_actions, masks):
        return self._step(observations, rnn_states, prev_actions, masks)[0]

    def act(self, observations, rnn_states, prev_actions, masks, deterministic=False):
        if os.environ.get("VLNCE_ACT_GRAPH", "0") == "1":   # opt-in: the whole call as one HIP graph
            graph = self.__dict__.get("_act_graph")
            if graph is None:
                graph = ActGraph(self)
                object.__setattr__(self, "_act_graph", graph)  # (not a sub-module, not in state_dict)
            if graph.usable(observations, rnn_states):
                return graph(observations, rnn_states, prev_actions, masks, deterministic)
        return self._act_eager(observations, rnn_states, prev_actions, masks, deterministic)

    def _act_eager(self, observations, rnn_states, prev_actions, masks, deterministic):
        dist, new_states = self._step(observations, rnn_states, prev_actions, masks)
        return (dist.mode() if deterministic else dist.sample()), new_states

    get_value = _not_part_of_imitation_learning
    evaluate_actions = _not_part_of_imitation_learning

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
