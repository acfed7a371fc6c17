"""Config node + observation/action space stand-ins for the policy plugin
surface.  The trainers hand `from_config` a yacs CfgNode and gym spaces; any
object with attribute access / `.defrost()/.freeze()` and `.spaces[...]
.shape` / `.n` works, so these light classes are only needed when habitat /
gym are absent (benchmarks, tests).  Defaults follow
vlnce_baselines/config/default.py:214-285."""
import numpy as np


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


class Box:
    def __init__(self, low, high, shape, dtype="float32"):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)


class Discrete:
    def __init__(self, n):
        self.n = int(n)


def _node(**kw):
    c = Config()
    c.update(kw)
    return c


def default_model_config():
    return _node(
        policy_name="CMAPolicy", normalize_rgb=False, ablate_depth=False, ablate_rgb=False,
        ablate_instruction=False,
        INSTRUCTION_ENCODER=_node(
            sensor_uuid="instruction", vocab_size=2504, use_pretrained_embeddings=False,
            embedding_file="NONE", fine_tune_embeddings=False, embedding_size=50,
            hidden_size=128, rnn_type="LSTM", final_state_only=True, bidirectional=False),
        RGB_ENCODER=_node(cnn_type="TorchVisionResNet50", output_size=256, trainable=False),
        DEPTH_ENCODER=_node(cnn_type="VlnResnetDepthEncoder", output_size=128, backbone="resnet50",
                            ddppo_checkpoint="NONE", trainable=False),
        STATE_ENCODER=_node(hidden_size=512, rnn_type="GRU"),
        PROGRESS_MONITOR=_node(use=False, alpha=1.0),
        SEQ2SEQ=_node(use_prev_action=False),
        WAYPOINT=_node(
            predict_distance=True, continuous_distance=True, min_distance_var=0.0625,
            max_distance_var=3.52, max_distance_prediction=2.75, min_distance_prediction=0.25,
            discrete_distances=6, predict_offset=True, continuous_offset=True,
            min_offset_var=0.0110, max_offset_var=0.0685, discrete_offsets=7,
            offset_temperature=1.0),
    )


def make_config(policy_name="CMAPolicy", **overrides):
    """Experiment-level config with the yaml overrides of the reference's
    cma.yaml / 1-wpn-cc.yaml applied; `overrides` are dotted MODEL.* keys."""
    cfg = _node(TORCH_GPU_ID=0, MODEL=default_model_config(),
                TASK_CONFIG=_node(TASK=_node(PANO_ROTATIONS=12)))
    m = cfg.MODEL
    m.policy_name = policy_name
    if policy_name == "CMAPolicy":
        m.INSTRUCTION_ENCODER.bidirectional = True
    if policy_name == "WaypointPolicy":
        m.WAYPOINT.update(min_offset_var=0.00030625, max_offset_var=0.06853892,
                          offset_temperature=4.0, min_distance_var=0.01, max_distance_var=3.516,
                          max_distance_prediction=4.0, min_distance_prediction=0.25)
        m.INSTRUCTION_ENCODER.update(bidirectional=True, final_state_only=False)
        m.RGB_ENCODER.update(cnn_type="TorchVisionResNet18", output_size=128)
        m.STATE_ENCODER.hidden_size = 256
    for k, v in overrides.items():
        node = m
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def make_spaces(h, w, pano=False, num_actions=4):
    shp_d = (12, h, w, 1) if pano else (h, w, 1)
    shp_r = (12, h, w, 3) if pano else (h, w, 3)
    return Dict({"rgb": Box(0, 255, shp_r), "depth": Box(0.0, 1.0, shp_d)}), Discrete(num_actions)
