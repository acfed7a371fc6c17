"""Seq2Seq policy (reference: vlnce_baselines/models/seq2seq_policy.py:20-179)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .aux_losses import AuxLosses
from .encoders import resnet_encoders
from .encoders.instruction_encoder import InstructionEncoder
from .policy import ILPolicy, Net
from .registry import baseline_registry
from .rnn_state_encoder import build_rnn_state_encoder
from .streams import BranchStreams


def prev_action_index(prev_actions, masks):
    """((a + 1) * mask).long(): index 0 = episode start (cma_policy.py:233-235)."""
    return ((prev_actions.float() + 1) * masks).long().view(-1)


def register_progress_loss(net, x, observations):
    """tanh(Linear(x)) vs observations["progress"], including F.mse_loss's
    [B] x [B,1] -> [B,B] broadcast of the reference."""
    cfg = net.model_config
    if cfg.PROGRESS_MONITOR.use and AuxLosses.is_active():
        hat = ops.linear(x, net.progress_monitor.weight, net.progress_monitor.bias, ops.ACT_TANH)
        hat_b, tgt_b = torch.broadcast_tensors(hat.squeeze(1), observations["progress"])
        AuxLosses.register_loss("progress_monitor", (hat_b - tgt_b) ** 2,
                                cfg.PROGRESS_MONITOR.alpha)


@baseline_registry.register_policy
class Seq2SeqPolicy(ILPolicy):
    def __init__(self, observation_space, action_space, model_config):
        super().__init__(
            Seq2SeqNet(observation_space=observation_space, model_config=model_config,
                       num_actions=action_space.n),
            action_space.n,
        )

    @classmethod
    def from_config(cls, config, observation_space, action_space):
        config.defrost()
        config.MODEL.TORCH_GPU_ID = config.TORCH_GPU_ID
        config.freeze()
        return cls(observation_space=observation_space, action_space=action_space,
                   model_config=config.MODEL)


class Seq2SeqNet(Net):
    """instruction final state || depth fc || rgb fc (|| prev action) -> GRU/LSTM."""

    def __init__(self, observation_space, model_config, num_actions):
        super().__init__()
        self.model_config = model_config
        self.instruction_encoder = InstructionEncoder(model_config.INSTRUCTION_ENCODER)
        assert model_config.DEPTH_ENCODER.cnn_type in ["VlnResnetDepthEncoder"]
        self.depth_encoder = getattr(resnet_encoders, model_config.DEPTH_ENCODER.cnn_type)(
            observation_space,
            output_size=model_config.DEPTH_ENCODER.output_size,
            checkpoint=model_config.DEPTH_ENCODER.ddppo_checkpoint,
            backbone=model_config.DEPTH_ENCODER.backbone,
            trainable=model_config.DEPTH_ENCODER.trainable,
        )
        assert model_config.RGB_ENCODER.cnn_type in ["TorchVisionResNet18", "TorchVisionResNet50"]
        self.rgb_encoder = getattr(resnet_encoders, model_config.RGB_ENCODER.cnn_type)(
            model_config.RGB_ENCODER.output_size,
            normalize_visual_inputs=model_config.normalize_rgb,
            trainable=model_config.RGB_ENCODER.trainable,
            spatial_output=False,
        )
        if model_config.SEQ2SEQ.use_prev_action:
            self.prev_action_embedding = nn.Embedding(num_actions + 1, 32)
        rnn_input_size = (self.instruction_encoder.output_size
                          + model_config.DEPTH_ENCODER.output_size
                          + model_config.RGB_ENCODER.output_size)
        if model_config.SEQ2SEQ.use_prev_action:
            rnn_input_size += self.prev_action_embedding.embedding_dim
        self.state_encoder = build_rnn_state_encoder(
            input_size=rnn_input_size, hidden_size=model_config.STATE_ENCODER.hidden_size,
            rnn_type=model_config.STATE_ENCODER.rnn_type, num_layers=1)
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
        dev = rnn_states.device
        fork = self._branches.fork(dev)
        rgb_embedding = self.rgb_encoder(observations)
        instruction_embedding, join_i = self._branches.run(
            fork, 0, dev, lambda: self.instruction_encoder(observations))
        depth_embedding, join_d = self._branches.run(
            fork, 0, dev, lambda: self.depth_encoder(observations))
        join_i()
        join_d()
        if mc.ablate_instruction:
            instruction_embedding = instruction_embedding * 0
        if mc.ablate_depth:
            depth_embedding = depth_embedding * 0
        if mc.ablate_rgb:
            rgb_embedding = rgb_embedding * 0
        parts = [instruction_embedding, depth_embedding, rgb_embedding]
        if mc.SEQ2SEQ.use_prev_action:
            parts.append(F.embedding(prev_action_index(prev_actions, masks),
                                     self.prev_action_embedding.weight))
        x, rnn_states_out = self.state_encoder(torch.cat(parts, dim=1), rnn_states, masks)
        register_progress_loss(self, x, observations)
        return x, rnn_states_out
