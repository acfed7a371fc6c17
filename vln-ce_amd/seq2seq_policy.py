"""Seq2Seq policy (reference: vlnce_baselines/models/seq2seq_policy.py:20-179)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoders.instruction_encoder import InstructionEncoder
from .net_parts import (apply_ablations, build_depth_encoder, build_rgb_encoder,
                        encode_three_branches, prev_action_index, register_progress_loss)
from .policy import ILPolicy, Net
from .registry import baseline_registry
from .rnn_state_encoder import build_rnn_state_encoder
from .streams import BranchStreams


@baseline_registry.register_policy
class Seq2SeqPolicy(ILPolicy):
    def __init__(self, observation_space, action_space, model_config):
        super().__init__(
            Seq2SeqNet(observation_space=observation_space, model_config=model_config,
                       num_actions=action_space.n),
            action_space.n,
        )


class Seq2SeqNet(Net):
    """instruction final state || depth fc || rgb fc (|| prev action) -> GRU/LSTM."""

    def __init__(self, observation_space, model_config, num_actions):
        super().__init__()
        self.model_config = model_config
        # (attribute order = state_dict / parameter order of the reference)
        self.instruction_encoder = InstructionEncoder(model_config.INSTRUCTION_ENCODER)
        self.depth_encoder = build_depth_encoder(observation_space, model_config)
        self.rgb_encoder = build_rgb_encoder(model_config, spatial_output=False)
        widths = [self.instruction_encoder.output_size, model_config.DEPTH_ENCODER.output_size,
                  model_config.RGB_ENCODER.output_size]
        if model_config.SEQ2SEQ.use_prev_action:
            self.prev_action_embedding = nn.Embedding(num_actions + 1, 32)
            widths.append(self.prev_action_embedding.embedding_dim)
        state_cfg = model_config.STATE_ENCODER
        self.state_encoder = build_rnn_state_encoder(
            input_size=sum(widths), hidden_size=state_cfg.hidden_size,
            rnn_type=state_cfg.rnn_type, num_layers=1)
        self._branches = BranchStreams()
        self.progress_monitor = nn.Linear(model_config.STATE_ENCODER.hidden_size, 1)
        nn.init.kaiming_normal_(self.progress_monitor.weight, nonlinearity="tanh")
        nn.init.constant_(self.progress_monitor.bias, 0)
        self.train()

    @property
    def output_size(self):
        return self.model_config.STATE_ENCODER.hidden_size

    @property
    def is_blind(self):
        return self.rgb_encoder.is_blind or self.depth_encoder.is_blind

    @property
    def num_recurrent_layers(self):
        return self.state_encoder.num_recurrent_layers

    def forward(self, observations, rnn_states, prev_actions, masks):
        mc = self.model_config
        feats = list(apply_ablations(
            mc, *encode_three_branches(self, observations, rnn_states.device)))
        if mc.SEQ2SEQ.use_prev_action:
            feats.append(F.embedding(prev_action_index(prev_actions, masks),
                                     self.prev_action_embedding.weight))
        state, rnn_states_out = self.state_encoder(torch.cat(feats, dim=1), rnn_states, masks)
        register_progress_loss(self, state, observations)
        return state, rnn_states_out
