"""Pieces the Seq2Seq, CMA and waypoint nets share: construction of the visual encoders from
`config.MODEL`, the three-branch encoder pass on side HIP streams, ablation switches, the
previous-action index and the progress-monitor auxiliary loss."""
import torch

from . import ops
from .aux_losses import AuxLosses
from .encoders import resnet_encoders

_DEPTH_TYPES = ("VlnResnetDepthEncoder",)
_RGB_TYPES = ("TorchVisionResNet18", "TorchVisionResNet50")


def build_depth_encoder(observation_space, model_config, **extra):
    """MODEL.DEPTH_ENCODER -> encoder module (seq2seq_policy.py:69-81, cma_policy.py:69-82,
    waypoint_predictors.py:41-52)."""
    cfg = model_config.DEPTH_ENCODER
    if cfg.cnn_type not in _DEPTH_TYPES:
        raise AssertionError(f"DEPTH_ENCODER.cnn_type must be one of {_DEPTH_TYPES}")
    cls = getattr(resnet_encoders, cfg.cnn_type)
    trainable = extra.pop("trainable", cfg.trainable)  # the waypoint net never unfreezes it
    return cls(observation_space, output_size=cfg.output_size, checkpoint=cfg.ddppo_checkpoint,
               backbone=cfg.backbone, trainable=trainable, **extra)


def build_rgb_encoder(model_config, **extra):
    """MODEL.RGB_ENCODER -> encoder module (seq2seq_policy.py:83-93, cma_policy.py:84-96)."""
    cfg = model_config.RGB_ENCODER
    if cfg.cnn_type not in _RGB_TYPES:
        raise AssertionError(f"RGB_ENCODER.cnn_type must be one of {_RGB_TYPES}")
    cls = getattr(resnet_encoders, cfg.cnn_type)
    trainable = extra.pop("trainable", cfg.trainable)
    return cls(cfg.output_size, normalize_visual_inputs=model_config.normalize_rgb,
               trainable=trainable, pretrained_weights=cfg.get("pretrained_weights", None)
               if hasattr(cfg, "get") else getattr(cfg, "pretrained_weights", None), **extra)


def relu_fc(n_in, n_out, *front):
    """[*front, Linear(n_in, n_out), ReLU]: the small projection blocks of the nets (the module
    index of the Linear inside the Sequential is part of the checkpoint key)."""
    return torch.nn.Sequential(*front, torch.nn.Linear(n_in, n_out), torch.nn.ReLU(True))


def encode_three_branches(net, observations, device, distinct_instructions=False):
    """RGB encoder on the caller's stream; instruction encoder (one host sync for the lengths,
    as upstream) and depth encoder on a side stream, overlapping the RGB trunk's long MFMA
    kernels.  Returns (instruction, depth, rgb) with the side-stream results joined; with
    `distinct_instructions` the first is InstructionEncoder.forward(..., distinct=True)'s pair."""
    branches = net._branches

    def instruction():
        if distinct_instructions:
            return net.instruction_encoder(observations, distinct=True)
        return net.instruction_encoder(observations)

    fork = branches.fork(device)
    if not torch.is_grad_enabled():
        # act() at a handful of environments: every branch is a chain of latency-bound launches
        # and the longest is the depth trunk (~250 of them, GroupNorm = 3-4 launches per layer;
        # profiles/archive/r03_f_act_one_call.txt), which used to start only after the instruction
        # encoder's host sync because both shared side stream 0.  It is issued first and on its
        # own stream (the one a run-ahead depth trunk would use), then RGB, then the instruction.
        dep, join_dep = branches.run(fork, 2, device, lambda: net.depth_encoder(observations))
        rgb = net.rgb_encoder(observations)
        ins, join_ins = branches.run(fork, 0, device, instruction)
    else:
        # training step: RGB trunk on the caller's stream, the instruction encoder (~1 ms, one host
        # sync for the lengths) on side stream 0 and the depth trunk on side stream 2.  Every one
        # of the depth trunk's ~200 small launches waits for a boundary between the RGB trunk's
        # one-workgroup-per-CU kernels, so it takes ~6 ms beside the RGB trunk (1.4 ms alone) and
        # must not also wait behind the instruction encoder: on a shared side stream it ended
        # 0.2 ms AFTER the RGB trunk (profiles/r05_d_*; 9.71 -> 9.57 ms/step with its own stream,
        # r05_g_depth_trunk_stream_and_order.txt; issuing it before the RGB trunk delays that
        # one by as much as it gains).  GPU event stamps, not the tracer, are the evidence for
        # overlap on this runtime (profiles/archive/r04_d_overlap_probe2_event_stamps.txt).
        rgb = net.rgb_encoder(observations)
        ins, join_ins = branches.run(fork, 0, device, instruction)
        dep, join_dep = branches.run(fork, 2, device, lambda: net.depth_encoder(observations))
    join_ins()
    join_dep()
    return ins, dep, rgb


def apply_ablations(model_config, ins, dep, rgb):
    """MODEL.ablate_{instruction,depth,rgb}: the feature is multiplied by zero, not removed."""
    if model_config.ablate_instruction:
        ins = ins * 0
    if model_config.ablate_depth:
        dep = dep * 0
    if model_config.ablate_rgb:
        rgb = rgb * 0
    return ins, dep, rgb


def prev_action_index(prev_actions, masks):
    """((a + 1) * mask).long(): embedding row 0 = first step of an episode (cma_policy.py:233-235)."""
    return ((prev_actions.float() + 1) * masks).long().view(-1)


def register_progress_loss(net, features, observations):
    """Progress monitor: tanh(Linear(features)) against observations["progress"], including the
    [B] x [B,1] -> [B,B] broadcast of the reference's F.mse_loss call."""
    cfg = net.model_config.PROGRESS_MONITOR
    if not (cfg.use and AuxLosses.is_active()):
        return
    head = net.progress_monitor
    estimate = ops.linear(features, head.weight, head.bias, ops.ACT_TANH).squeeze(1)
    # rows i of the [B, B] loss matrix range over the progress targets of the WHOLE batch: under
    # data parallelism they are gathered from all ranks (AuxLosses.set_data_parallel)
    estimate, target = torch.broadcast_tensors(estimate,
                                               AuxLosses.gather_rows(observations["progress"]))
    AuxLosses.register_loss("progress_monitor", (estimate - target) ** 2, cfg.alpha)
