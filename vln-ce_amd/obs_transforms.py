"""Observation transforms on the device (the step immediately before the policy).

Mirrors `habitat_extensions/obs_transformers.py:21-145` -- `CenterCropperPerSensor` and
`ObsStack`, same constructor arguments, same `from_config`, same observation-space rewrite,
`forward(observations)` mutating and returning the dict -- plus `batch_obs`, the list-of-dicts
-> device-batch step the trainers call first (`dagger_trainer.py:276`,
`ddppo_waypoint_trainer.py:232`).  Differences that matter for speed, none for values:

* `batch_obs` ships every sensor in its STORAGE dtype: uint8 RGB stays uint8 (habitat v0.1.7 casts
  to fp32 before the copy: 4x the PCIe bytes for the RGB frames); each sensor is stacked into one
  pinned staging buffer and copied once.  The encoders take uint8 or fp32 frames alike (the
  stem's ingest kernel does the cast and the /255).
* `CenterCropperPerSensor` returns centre-crop VIEWS exactly like habitat's `center_crop`; the
  encoders' ingest kernel reads through the view (ops._frame_view), so the crop is never copied.
* `ObsStack` stacks the 12 camera sensors with ONE kernel (`vlnce_frames_gather`), in the
  sensors' own dtype.

When habitat_baselines is importable the classes register themselves under the reference's
names, replacing the torch-op versions."""
import copy
import numbers

import numpy as np
import torch

from . import ops
from .config import Box


def center_crop_window(h, w, size):
    """habitat_baselines.utils.common.center_crop's window [3P, v0.1.7]: (y0, x0, H, W)."""
    ch, cw = (size, size) if isinstance(size, numbers.Number) else size
    ch, cw = int(ch), int(cw)
    return h // 2 - ch // 2, w // 2 - cw // 2, ch, cw


_PINNED = {}


def batch_obs(observations, device=None):
    """List of per-env observation dicts (numpy arrays / tensors) -> dict of [N, ...] device
    tensors in the sensors' own dtypes (habitat_baselines.utils.common.batch_obs [3P] minus the
    fp32 cast).  One pinned staging buffer per (sensor, shape, dtype), one H2D copy per sensor."""
    out = {}
    if not observations:
        return out
    dev = torch.device(device) if device is not None else torch.device("cpu")
    for sensor in observations[0]:
        first = torch.as_tensor(observations[0][sensor])
        shape = (len(observations),) + tuple(first.shape)
        if dev.type != "cuda":
            out[sensor] = torch.stack([torch.as_tensor(o[sensor]) for o in observations], dim=0)
            continue
        key = (sensor, shape, first.dtype)
        stage = _PINNED.get(key)
        if stage is None:
            stage = torch.empty(shape, dtype=first.dtype).pin_memory()
            _PINNED[key] = stage
        for i, o in enumerate(observations):
            stage[i].copy_(torch.as_tensor(o[sensor]))
        out[sensor] = stage.to(dev, non_blocking=True)
    if dev.type == "cuda":
        # the staging buffers are reused by the next call: the copies must have left them
        torch.cuda.current_stream(dev).synchronize()
    return out


def _space_with_shape(space, hw):
    """habitat's overwrite_gym_box_shape [3P]: same bounds, new (H, W) in the trailing dims"""
    shape = tuple(space.shape[:-3]) + tuple(hw) + (space.shape[-1],)
    return Box(float(np.min(space.low)), float(np.max(space.high)), shape, space.dtype)


class CenterCropperPerSensor(torch.nn.Module):
    """obs_transformers.py:21-92."""

    def __init__(self, sensor_crops, channels_last=True):
        super().__init__()
        self.sensor_crops = dict(sensor_crops)
        for k, size in self.sensor_crops.items():
            if isinstance(size, numbers.Number):
                self.sensor_crops[k] = (int(size), int(size))
            assert len(self.sensor_crops[k]) == 2, "forced input size must be len of 2 (h, w)"
        assert channels_last, "the policies take channels-last sensors"
        self.channels_last = channels_last

    def transform_observation_space(self, observation_space):
        observation_space = copy.deepcopy(observation_space)
        for key, space in observation_space.spaces.items():
            if key in self.sensor_crops and tuple(space.shape[-3:-1]) != tuple(self.sensor_crops[key]):
                observation_space.spaces[key] = _space_with_shape(space, self.sensor_crops[key])
        return observation_space

    @torch.no_grad()
    def forward(self, observations):
        for sensor, size in self.sensor_crops.items():
            if sensor in observations:
                t = observations[sensor]
                y0, x0, h, w = center_crop_window(t.size(-3), t.size(-2), size)
                observations[sensor] = t[..., y0:y0 + h, x0:x0 + w, :]
        return observations

    @classmethod
    def from_config(cls, config):
        return cls(config.RL.POLICY.OBS_TRANSFORMS.CENTER_CROPPER_PER_SENSOR.SENSOR_CROPS)


class ObsStack(torch.nn.Module):
    """obs_transformers.py:95-145: several same-shaped sensors -> one [N, len, ...] sensor."""

    def __init__(self, sensor_rewrites):
        super().__init__()
        self.rewrite_dict = dict(sensor_rewrites)

    def transform_observation_space(self, observation_space):
        observation_space = copy.deepcopy(observation_space)
        for target_uuid, sensors in self.rewrite_dict.items():
            orig = observation_space.spaces[sensors[0]]
            for k in sensors:
                del observation_space.spaces[k]
            observation_space.spaces[target_uuid] = Box(
                float(np.min(orig.low)), float(np.max(orig.high)),
                (len(sensors),) + tuple(orig.shape), orig.dtype)
        return observation_space

    @torch.no_grad()
    def forward(self, observations):
        for new_key, old_keys in self.rewrite_dict.items():
            srcs = [observations[k] for k in old_keys]
            s0 = srcs[0]
            if s0.is_cuda and s0.dim() == 4 and len(srcs) <= 16:
                new_obs = ops.frames_gather(srcs)
            else:
                new_obs = torch.stack(srcs, dim=1)
            for k in old_keys:
                del observations[k]
            observations[new_key] = new_obs
        return observations

    @classmethod
    def from_config(cls, config):
        return cls(config.RL.POLICY.OBS_TRANSFORMS.OBS_STACK.SENSOR_REWRITES)


def apply_obs_transforms_batch(batch, obs_transforms):
    """habitat_baselines.common.obs_transformers.apply_obs_transforms_batch [3P]"""
    for t in obs_transforms:
        batch = t(batch)
    return batch


def _register_with_habitat():
    try:
        from habitat_baselines.common.baseline_registry import baseline_registry as hb
    except Exception:
        return
    for cls in (CenterCropperPerSensor, ObsStack):
        try:
            hb.register_obs_transformer(name=cls.__name__)(cls)
        except Exception:
            pass


_register_with_habitat()
