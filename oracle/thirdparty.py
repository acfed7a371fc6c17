"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (plain torch.nn primitives, fp32) of the *third-party* pieces
the VLN-CE policy hot path calls but that are NOT vendored under
/root/reference: habitat-lab v0.1.7 (`habitat_baselines`) and torchvision.

Named dependency + pin (reference README.md:31-41, requirements.txt:11-12):
  * habitat-lab  v0.1.7  -> resnet.py, resnet_policy.ResNetEncoder,
    rnn_state_encoder.build_rnn_state_encoder, ppo.policy.{Policy,Net,
    CriticHead}, utils.common.{CategoricalNet,CustomFixedCategorical}
  * torchvision  0.2.2.post3 -> models.resnet18 / resnet50 graph

PARITY UNPINNED for this file: neither package is installed here and the
reference carries no test that pins these modules, so they are restated from
their published structure.  Every block reduces to
torch.nn primitives whose CPU kernels are the per-op oracle.  The reference's
call sites are cited next to each class.  The torchvision graphs are held to
the facts torchvision publishes about them (parameter counts, multiply-
accumulates per 224x224 frame, checkpoint key layout, stride placement):
tests/test_thirdparty_published_facts.py; habitat-lab publishes no such table.

This module is shared by (a) tools/oracle/shims.py, which injects it under the
third-party module names so the reference's own model files import unchanged,
and (b) oracle/policy_cpu.py, the first-party restatement.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# minimal gym-like spaces (reference: `from gym import Space, spaces`)
# --------------------------------------------------------------------------
# habitat_baselines/utils/common.py::image_resize_shortest_edge and
# habitat_baselines/common/obs_transformers.py::ResizeShortestEdge [3P-mem, habitat-lab v0.1.7;
# not under /root/reference, PARITY UNPINNED].  Enabled by every RxR config of the reference
# (vlnce_baselines/config/rxr_baselines/rxr_cma_en.yaml:27-30: ENABLED_TRANSFORMS
# [ResizeShortestEdge, CenterCropperPerSensor]) and applied per step at
# vlnce_baselines/common/base_il_trainer.py:284-285.  Published algorithm: scale =
# size / min(h, w); (h, w) <- (int(h*scale), int(w*scale)); torch.nn.functional.interpolate(
# img.float(), size=(h, w), mode="area").to(dtype=img.dtype) on the NCHW view of the frames.
def image_resize_shortest_edge(img, size, channels_last=False):
    img = torch.as_tensor(img)
    no_batch_dim = len(img.shape) == 3
    if len(img.shape) < 3 or len(img.shape) > 5:
        raise NotImplementedError()
    if no_batch_dim:
        img = img.unsqueeze(0)
    if channels_last:
        h, w = img.shape[-3:-1]
        img = img.permute(0, 3, 1, 2) if len(img.shape) == 4 else img.permute(0, 1, 4, 2, 3)
    else:
        h, w = img.shape[-2:]
    scale = size / min(h, w)
    h, w = int(h * scale), int(w * scale)
    lead = img.shape[:-3]
    flat = img.reshape((-1,) + tuple(img.shape[-3:]))
    flat = F.interpolate(flat.float(), size=(h, w), mode="area").to(dtype=img.dtype)
    img = flat.reshape(tuple(lead) + tuple(flat.shape[-3:]))
    if channels_last:
        img = img.permute(0, 2, 3, 1) if len(img.shape) == 4 else img.permute(0, 1, 3, 4, 2)
    if no_batch_dim:
        img = img.squeeze(dim=0)
    return img


def resize_shortest_edge(observations, size, trans_keys=("rgb", "depth", "semantic")):
    """ResizeShortestEdge.forward [3P-mem]"""
    for sensor in trans_keys:
        if sensor in observations:
            observations[sensor] = image_resize_shortest_edge(observations[sensor], size,
                                                              channels_last=True)
    return observations


# --------------------------------------------------------------------------
class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape, dtype="float32"):
        import numpy as np

        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)


# --------------------------------------------------------------------------
# habitat_baselines.rl.ddppo.policy.resnet  (GroupNorm ResNet)
# call site: vlnce_baselines/models/encoders/resnet_encoders.py:10,31-43
# --------------------------------------------------------------------------
def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


def _c1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


class GNBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None):
        super().__init__()
        self.convs = nn.Sequential(
            _c3(inplanes, planes, stride),
            nn.GroupNorm(ngroups, planes),
            nn.ReLU(True),
            _c3(planes, planes),
            nn.GroupNorm(ngroups, planes),
        )
        self.downsample = downsample
        self.relu = nn.ReLU(True)

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(self.convs(x) + skip)


class GNBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None):
        super().__init__()
        self.convs = nn.Sequential(
            _c1(inplanes, planes),
            nn.GroupNorm(ngroups, planes),
            nn.ReLU(True),
            _c3(planes, planes, stride),
            nn.GroupNorm(ngroups, planes),
            nn.ReLU(True),
            _c1(planes, planes * self.expansion),
            nn.GroupNorm(ngroups, planes * self.expansion),
        )
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(self.convs(x) + skip)


class GNResNet(nn.Module):
    def __init__(self, in_channels, base_planes, ngroups, block, layers):
        super().__init__()
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_channels, base_planes, 7, stride=2, padding=3, bias=False),
            nn.GroupNorm(ngroups, base_planes),
            nn.ReLU(True),
        )
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = base_planes
        self.layer1 = self._stage(block, ngroups, base_planes, layers[0], 1)
        self.layer2 = self._stage(block, ngroups, base_planes * 2, layers[1], 2)
        self.layer3 = self._stage(block, ngroups, base_planes * 4, layers[2], 2)
        self.layer4 = self._stage(block, ngroups, base_planes * 8, layers[3], 2)
        self.final_channels = self.inplanes
        self.final_spatial_compress = 1.0 / (2 ** 5)

    def _stage(self, block, ngroups, planes, nblocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(
                _c1(self.inplanes, planes * block.expansion, stride),
                nn.GroupNorm(ngroups, planes * block.expansion),
            )
        blocks = [block(self.inplanes, planes, ngroups, stride, down)]
        self.inplanes = planes * block.expansion
        for _ in range(1, nblocks):
            blocks.append(block(self.inplanes, planes, ngroups))
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def gn_resnet18(in_channels, base_planes, ngroups):
    return GNResNet(in_channels, base_planes, ngroups, GNBasicBlock, [2, 2, 2, 2])


def gn_resnet50(in_channels, base_planes, ngroups):
    return GNResNet(in_channels, base_planes, ngroups, GNBottleneck, [3, 4, 6, 3])


# --------------------------------------------------------------------------
# habitat_baselines.rl.ddppo.policy.resnet_policy.ResNetEncoder
# call site: resnet_encoders.py:11,31-43 (depth only; rgb branch kept for
# completeness of the constructor contract)
# --------------------------------------------------------------------------
class ResNetEncoder(nn.Module):
    def __init__(
        self,
        observation_space,
        baseplanes=32,
        ngroups=32,
        spatial_size=128,
        make_backbone=None,
        normalize_visual_inputs=False,
    ):
        super().__init__()
        sp = observation_space.spaces
        self._n_input_rgb = sp["rgb"].shape[2] if "rgb" in sp else 0
        if "rgb" in sp:
            spatial_size = sp["rgb"].shape[0] // 2
        self._n_input_depth = sp["depth"].shape[2] if "depth" in sp else 0
        if "depth" in sp:
            spatial_size = sp["depth"].shape[0] // 2
        assert not normalize_visual_inputs, "RunningMeanAndVar not on the VLN-CE path"
        self.running_mean_and_var = nn.Sequential()
        cin = self._n_input_depth + self._n_input_rgb
        self.backbone = make_backbone(cin, baseplanes, ngroups)
        final_spatial = int(spatial_size * self.backbone.final_spatial_compress)
        ncomp = int(round(2048 / (final_spatial ** 2)))
        self.compression = nn.Sequential(
            nn.Conv2d(self.backbone.final_channels, ncomp, 3, padding=1, bias=False),
            nn.GroupNorm(1, ncomp),
            nn.ReLU(True),
        )
        self.output_shape = (ncomp, final_spatial, final_spatial)

    @property
    def is_blind(self):
        return self._n_input_rgb + self._n_input_depth == 0

    def layer_init(self):
        for layer in self.modules():
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, val=0)

    def forward(self, observations):
        parts = []
        if self._n_input_rgb > 0:
            parts.append(observations["rgb"].permute(0, 3, 1, 2) / 255.0)
        if self._n_input_depth > 0:
            parts.append(observations["depth"].permute(0, 3, 1, 2))
        x = F.avg_pool2d(torch.cat(parts, dim=1), 2)
        x = self.running_mean_and_var(x)
        return self.compression(self.backbone(x))


# --------------------------------------------------------------------------
# habitat_baselines.rl.models.rnn_state_encoder.build_rnn_state_encoder
# call sites: seq2seq_policy.py:109-114, cma_policy.py:126-131,172-177,
#             waypoint_predictors.py:69-74,157-162
# --------------------------------------------------------------------------
class RNNStateEncoder(nn.Module):
    """Single-layer GRU/LSTM with done-mask zeroing.  hidden_states are
    batch-first [N, L, H]; LSTM packs (h, c) as two "layers".  x is [N, D]
    (one step) or the time-major flattening [T*N, D] (sequence)."""

    def __init__(self, input_size, hidden_size, rnn_type="GRU", num_layers=1):
        super().__init__()
        assert num_layers == 1
        self.is_lstm = rnn_type == "LSTM"
        self.rnn = (nn.LSTM if self.is_lstm else nn.GRU)(input_size, hidden_size, num_layers)
        self.num_recurrent_layers = num_layers * (2 if self.is_lstm else 1)
        for name, p in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)

    def _unpack(self, hs):
        if self.is_lstm:
            h, c = torch.chunk(hs, 2, 0)
            return h.contiguous(), c.contiguous()
        return hs.contiguous()

    def _pack(self, hs):
        return torch.cat(hs, 0) if self.is_lstm else hs

    def forward(self, x, hidden_states, masks):
        hs = hidden_states.permute(1, 0, 2)  # [L, N, H]
        n = hs.size(1)
        t = x.size(0) // n
        xs = x.view(t, n, x.size(1))
        ms = masks.view(t, n, 1).to(x.dtype)
        outs = []
        for i in range(t):
            hs = hs * ms[i].view(1, n, 1)
            y, st = self.rnn(xs[i : i + 1], self._unpack(hs))
            hs = self._pack(st)
            outs.append(y[0])
        return torch.cat(outs, 0), hs.permute(1, 0, 2)


def build_rnn_state_encoder(input_size, hidden_size, rnn_type="GRU", num_layers=1):
    return RNNStateEncoder(input_size, hidden_size, rnn_type, num_layers)


# --------------------------------------------------------------------------
# habitat_baselines.utils.common.{CustomFixedCategorical,CategoricalNet}
# habitat_baselines.rl.ppo.policy.{Net,Policy,CriticHead}
# call sites: models/policy.py:4-5,15-21; waypoint_policy.py:8,27-33,229
# --------------------------------------------------------------------------
class HabitatFixedCategorical(torch.distributions.Categorical):
    def sample(self, sample_shape=torch.Size()):
        return super().sample(sample_shape).unsqueeze(-1)

    def log_probs(self, actions):
        return (
            super().log_prob(actions.squeeze(-1)).view(actions.size(0), -1).sum(-1).unsqueeze(-1)
        )

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)


class CategoricalNet(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.linear = nn.Linear(num_inputs, num_outputs)
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)

    def forward(self, x):
        return HabitatFixedCategorical(logits=self.linear(x))


class CriticHead(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.fc = nn.Linear(input_size, 1)
        nn.init.orthogonal_(self.fc.weight)
        nn.init.constant_(self.fc.bias, 0)

    def forward(self, x):
        return self.fc(x)


class Net(nn.Module):
    pass


class Policy(nn.Module):
    def __init__(self, net, dim_actions):
        super().__init__()
        self.net = net
        self.dim_actions = dim_actions
        self.action_distribution = CategoricalNet(self.net.output_size, self.dim_actions)
        self.critic = CriticHead(self.net.output_size)


# --------------------------------------------------------------------------
# torchvision.models.resnet18 / resnet50 (BatchNorm; stride on the 3x3)
# call site: resnet_encoders.py:7,136-139
# --------------------------------------------------------------------------
class TVBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _c3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + skip)


class TVBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _c3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _c1(planes, planes * 4)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + skip)


class TVResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(block, 64, layers[0], 1)
        self.layer2 = self._stage(block, 128, layers[1], 2)
        self.layer3 = self._stage(block, 256, layers[2], 2)
        self.layer4 = self._stage(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, block, planes, nblocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(
                _c1(self.inplanes, planes * block.expansion, stride),
                nn.BatchNorm2d(planes * block.expansion),
            )
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        for _ in range(1, nblocks):
            blocks.append(block(self.inplanes, planes))
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def tv_resnet18(pretrained=False, **kw):
    # ImageNet weights are a download; synthetic work uses seeded random init.
    return TVResNet(TVBasicBlock, [2, 2, 2, 2])


def tv_resnet50(pretrained=False, **kw):
    return TVResNet(TVBottleneck, [3, 4, 6, 3])


# --------------------------------------------------------------------------
# habitat.Config stand-in (yacs CfgNode): attribute dict with no-op freeze.
# --------------------------------------------------------------------------
class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def defrost(self):
        pass

    def freeze(self):
        pass

    def clone(self):
        out = Config()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, Config) else v
        return out


def default_model_config():
    """MODEL.* defaults of vlnce_baselines/config/default.py:214-285, with the
    two file-backed options switched off for synthetic work
    (use_pretrained_embeddings=False, ddppo_checkpoint="NONE")."""
    m = Config()
    m.policy_name = "CMAPolicy"
    m.normalize_rgb = False
    m.ablate_depth = False
    m.ablate_rgb = False
    m.ablate_instruction = False
    ie = Config()
    ie.sensor_uuid = "instruction"
    ie.vocab_size = 2504
    ie.use_pretrained_embeddings = False
    ie.embedding_file = "NONE"
    ie.fine_tune_embeddings = False
    ie.embedding_size = 50
    ie.hidden_size = 128
    ie.rnn_type = "LSTM"
    ie.final_state_only = True
    ie.bidirectional = False
    m.INSTRUCTION_ENCODER = ie
    rgb = Config()
    rgb.cnn_type = "TorchVisionResNet50"
    rgb.output_size = 256
    rgb.trainable = False
    m.RGB_ENCODER = rgb
    d = Config()
    d.cnn_type = "VlnResnetDepthEncoder"
    d.output_size = 128
    d.backbone = "resnet50"
    d.ddppo_checkpoint = "NONE"
    d.trainable = False
    m.DEPTH_ENCODER = d
    se = Config()
    se.hidden_size = 512
    se.rnn_type = "GRU"
    m.STATE_ENCODER = se
    pm = Config()
    pm.use = False
    pm.alpha = 1.0
    m.PROGRESS_MONITOR = pm
    s2s = Config()
    s2s.use_prev_action = False
    m.SEQ2SEQ = s2s
    w = Config()
    w.predict_distance = True
    w.continuous_distance = True
    w.min_distance_var = 0.0625
    w.max_distance_var = 3.52
    w.max_distance_prediction = 2.75
    w.min_distance_prediction = 0.25
    w.discrete_distances = 6
    w.predict_offset = True
    w.continuous_offset = True
    w.min_offset_var = 0.0110
    w.max_offset_var = 0.0685
    w.discrete_offsets = 7
    w.offset_temperature = 1.0
    m.WAYPOINT = w
    return m


def make_config(policy_name="CMAPolicy", **overrides):
    """Top-level experiment config as the trainers hand it to from_config():
    config.MODEL, config.TORCH_GPU_ID, config.TASK_CONFIG.TASK.PANO_ROTATIONS.
    `overrides` use dotted keys relative to MODEL, e.g.
    {"INSTRUCTION_ENCODER.bidirectional": True}."""
    cfg = Config()
    cfg.TORCH_GPU_ID = 0
    cfg.MODEL = default_model_config()
    cfg.MODEL.policy_name = policy_name
    task = Config()
    task.PANO_ROTATIONS = 12
    tc = Config()
    tc.TASK = task
    cfg.TASK_CONFIG = tc
    if policy_name == "CMAPolicy":
        # r2r_baselines/cma.yaml:34-35
        cfg.MODEL.INSTRUCTION_ENCODER.bidirectional = True
    if policy_name == "WaypointPolicy":
        # r2r_waypoint/1-wpn-cc.yaml:17-37
        w = cfg.MODEL.WAYPOINT
        w.min_offset_var = 0.00030625
        w.max_offset_var = 0.06853892
        w.offset_temperature = 4.0
        w.min_distance_var = 0.01
        w.max_distance_var = 3.516
        w.max_distance_prediction = 4.0
        w.min_distance_prediction = 0.25
        cfg.MODEL.INSTRUCTION_ENCODER.bidirectional = True
        cfg.MODEL.INSTRUCTION_ENCODER.final_state_only = False
        cfg.MODEL.RGB_ENCODER.cnn_type = "TorchVisionResNet18"
        cfg.MODEL.RGB_ENCODER.output_size = 128
        cfg.MODEL.STATE_ENCODER.hidden_size = 256
    for k, v in overrides.items():
        node = cfg.MODEL
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def make_spaces(h, w, pano=False, num_actions=4):
    shp_d = (12, h, w, 1) if pano else (h, w, 1)
    shp_r = (12, h, w, 3) if pano else (h, w, 3)
    obs = Dict({"rgb": Box(0, 255, shp_r), "depth": Box(0.0, 1.0, shp_d)})
    return obs, Discrete(num_actions)


# --------------------------------------------------------------------------
# deterministic, construction-order-independent weights keyed by state_dict
# names (so the reference import, the oracle and the HIP policy all get the
# same values through the shared key contract).
# --------------------------------------------------------------------------
def _key_seed(key):
    import zlib

    return zlib.crc32(key.encode()) & 0x7FFFFFFF


def synth_state_dict(module, salt=""):
    """Return a state_dict with every tensor replaced by seeded values of a
    sensible scale (signal keeps O(1) magnitude through ~50 layers)."""
    out = {}
    for key, ref in module.state_dict().items():
        g = torch.Generator().manual_seed(_key_seed(salt + key))
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros_like(ref)
            continue
        if not ref.dtype.is_floating_point:
            out[key] = ref.clone()
            continue
        if leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "running_var":
            t = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif leaf == "_scale":
            t = ref.clone()
        elif len(shape) == 1 and leaf == "weight":  # norm gains (kept < 1 so
            # ~50 residual layers stay O(1) in eval-mode BatchNorm too)
            t = 0.25 + 0.5 * torch.rand(shape, generator=g)
        elif len(shape) == 1:  # biases
            t = 0.1 * torch.randn(shape, generator=g)
        elif "embedding" in key and len(shape) == 2:
            t = torch.randn(shape, generator=g)
            if "embedding_layer" in key:
                t[0] = 0.0  # padding_idx=0 row (instruction_encoder.py:41-45)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 1.0
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        out[key] = t.to(ref.dtype)
    return out
