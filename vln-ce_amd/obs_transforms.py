"""Observation transforms on the device (the step immediately before the policy).

Mirrors `habitat_extensions/obs_transformers.py:21-145` -- `CenterCropperPerSensor` and
`ObsStack`, same constructor arguments, same `from_config`, same observation-space rewrite,
`forward(observations)` mutating and returning the dict -- plus `batch_obs`, the list-of-dicts
-> device-batch step the trainers call first (`dagger_trainer.py:276`,
`ddppo_waypoint_trainer.py:232`).  Differences that matter for speed, none for values:

* `batch_obs` ships uint8 image sensors as uint8 (habitat v0.1.7 casts them to fp32 before the
  copy: 4x the PCIe bytes for the RGB frames) and every other sensor as fp32 exactly like
  habitat (`torch.tensor(..., dtype=torch.float)`: progress / angle_features arrive as float64
  numpy arrays, shortest-path actions as int64); each sensor is stacked into one pinned staging
  buffer and copied once.  The encoders take uint8 or fp32 frames alike (the stem's ingest
  kernel does the cast and the /255).
* `ResizeShortestEdge` (habitat's own transformer, enabled by every RxR config in front of the
  centre crop) runs as ONE kernel (`vlnce_frames_resize_area`) in the sensor's dtype; when a
  `CenterCropperPerSensor` follows it, `apply_obs_transforms_batch` evaluates only the crop
  window of the resized image (same values, a fraction of the work).
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
        # storage dtype only for uint8 image sensors; everything else fp32 as habitat's batch_obs
        dtype = torch.uint8 if first.dtype == torch.uint8 else torch.float32
        if dev.type != "cuda":
            out[sensor] = torch.stack([torch.as_tensor(o[sensor]).to(dtype) for o in observations],
                                      dim=0)
            continue
        key = (sensor, shape, dtype)
        stage = _PINNED.get(key)
        if stage is None:
            stage = torch.empty(shape, dtype=dtype).pin_memory()
            _PINNED[key] = stage
        for i, o in enumerate(observations):
            stage[i].copy_(torch.as_tensor(o[sensor]))  # copy_ converts float64 / int64 -> fp32
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


def resize_shortest_edge_hw(h, w, size):
    """habitat_baselines.utils.common.image_resize_shortest_edge's output size [3P, v0.1.7]"""
    scale = size / min(h, w)
    return int(h * scale), int(w * scale)


class ResizeShortestEdge(torch.nn.Module):
    """habitat_baselines.common.obs_transformers.ResizeShortestEdge [3P, habitat-lab v0.1.7; not
    under /root/reference]: resize rgb / depth / semantic so that the shortest edge is `size`
    (area interpolation, result cast back to the sensor dtype).  Enabled in front of the centre
    crop by every RxR config (rxr_baselines/rxr_cma_en.yaml:27-30), applied per step at
    base_il_trainer.py:284-285."""

    def __init__(self, size, channels_last=True, trans_keys=("rgb", "depth", "semantic")):
        super().__init__()
        self._size = int(size)
        assert channels_last, "the policies take channels-last sensors"
        self.channels_last = channels_last
        self.trans_keys = tuple(trans_keys)

    def transform_observation_space(self, observation_space):
        observation_space = copy.deepcopy(observation_space)
        for key, space in observation_space.spaces.items():
            if key in self.trans_keys:
                h, w = space.shape[-3:-1]
                if self._size == min(h, w):
                    continue
                observation_space.spaces[key] = _space_with_shape(
                    space, resize_shortest_edge_hw(h, w, self._size))
        return observation_space

    def _resize(self, t, crop=None):
        """t [..., H, W, C] -> image_resize_shortest_edge(t) (optionally only the window `crop`)"""
        if t.dim() < 3 or t.dim() > 5:
            raise NotImplementedError()
        h, w = t.shape[-3:-1]
        oh, ow = resize_shortest_edge_hw(h, w, self._size)
        if t.is_cuda and t.dtype in (torch.uint8, torch.float32):
            return ops.frames_resize_area(t, (oh, ow), crop)
        lead = t.shape[:-3]
        v = t.reshape((-1,) + tuple(t.shape[-3:])).permute(0, 3, 1, 2)
        v = torch.nn.functional.interpolate(v.float(), size=(oh, ow), mode="area").to(dtype=t.dtype)
        v = v.permute(0, 2, 3, 1).reshape(tuple(lead) + (oh, ow, t.shape[-1]))
        if crop is not None:
            y0, x0, ch, cw = crop
            v = v[..., y0:y0 + ch, x0:x0 + cw, :]
        return v

    @torch.no_grad()
    def forward(self, observations):
        if self._size is not None:
            for sensor in self.trans_keys:
                if sensor in observations:
                    observations[sensor] = self._resize(observations[sensor])
        return observations

    @torch.no_grad()
    def forward_cropped(self, observations, cropper):
        """ResizeShortestEdge followed by `cropper` (a CenterCropperPerSensor), evaluating only the
        crop windows; sensors the cropper does not name are resized whole.  Same values as the two
        transforms in sequence."""
        for sensor in self.trans_keys:
            if sensor not in observations:
                continue
            t = observations[sensor]
            size = cropper.sensor_crops.get(sensor)
            if size is None:
                observations[sensor] = self._resize(t)
                continue
            oh, ow = resize_shortest_edge_hw(t.size(-3), t.size(-2), self._size)
            y0, x0, ch, cw = center_crop_window(oh, ow, size)
            if y0 < 0 or x0 < 0 or y0 + ch > oh or x0 + cw > ow:   # crop larger than the image:
                observations[sensor] = self._resize(t)             # let the cropper slice what exists
                continue
            observations[sensor] = self._resize(t, (y0, x0, ch, cw))
        for sensor, size in cropper.sensor_crops.items():   # sensors only the cropper touches
            if sensor in observations and sensor not in self.trans_keys:
                observations = CenterCropperPerSensor({sensor: size})(observations)
        return observations

    @classmethod
    def from_config(cls, config):
        return cls(config.RL.POLICY.OBS_TRANSFORMS.RESIZE_SHORTEST_EDGE.SIZE)


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
    i = 0
    while i < len(obs_transforms):
        t = obs_transforms[i]
        nxt = obs_transforms[i + 1] if i + 1 < len(obs_transforms) else None
        if isinstance(t, ResizeShortestEdge) and isinstance(nxt, CenterCropperPerSensor):
            batch = t.forward_cropped(batch, nxt)   # only the crop window of the resized frames
            i += 2
            continue
        batch = t(batch)
        i += 1
    return batch


def _register_with_habitat():
    try:
        from habitat_baselines.common.baseline_registry import baseline_registry as hb
    except Exception:
        return
    for cls in (CenterCropperPerSensor, ObsStack, ResizeShortestEdge):
        try:
            hb.register_obs_transformer(name=cls.__name__)(cls)
        except Exception:
            pass


_register_with_habitat()
