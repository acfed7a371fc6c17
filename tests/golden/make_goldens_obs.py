"""Golden vectors of the observation transforms, produced by running the REAL reference
classes (habitat_extensions/obs_transformers.py: CenterCropperPerSensor, ObsStack) in the CPU
container.  Third-party helpers they import are shimmed: habitat's `center_crop`,
`get_image_height_width`, `overwrite_gym_box_shape`, `ObservationTransformer` [3P-mem v0.1.7].

    python tests/golden/make_goldens_obs.py        # writes tests/golden/obs_*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import policy_cpu as oc  # noqa: E402
from oracle import thirdparty as tp  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference_transforms():
    spaces = _mod("gym.spaces", Box=tp.Box, Dict=tp.Dict, Discrete=tp.Discrete, Space=tp.Space)
    _mod("gym", Space=tp.Space, spaces=spaces)
    _mod("habitat")
    _mod("habitat.config", Config=tp.Config)
    _mod("habitat.core")
    _mod("habitat.core.logging", logger=types.SimpleNamespace(info=lambda *a, **k: None))
    _mod("habitat.core.simulator", Observations=dict)

    class _Reg:
        @staticmethod
        def register_obs_transformer(to_register=None, *, name=None):
            return (lambda c: c) if to_register is None else to_register

    _mod("habitat_baselines")
    _mod("habitat_baselines.common")
    _mod("habitat_baselines.common.baseline_registry", baseline_registry=_Reg)
    _mod("habitat_baselines.common.obs_transformers", ObservationTransformer=torch.nn.Module)

    def get_image_height_width(img, channels_last=False):
        return (img.shape[-3:-1] if channels_last else img.shape[-2:])

    def overwrite_gym_box_shape(box, shape):
        return tp.Box(float(np.min(box.low)), float(np.max(box.high)),
                      tuple(box.shape[:-3]) + tuple(shape) + (box.shape[-1],), box.dtype)

    _mod("habitat_baselines.utils")
    _mod("habitat_baselines.utils.common", center_crop=oc.center_crop,
         get_image_height_width=get_image_height_width,
         overwrite_gym_box_shape=overwrite_gym_box_shape)
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(
        "ref_obs_transformers", "/root/reference/habitat_extensions/obs_transformers.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def inputs(seed=7, n=2, hs=20, ws=26):
    g = torch.Generator().manual_seed(seed)
    obs = {}
    for i in range(12):
        sfx = "" if i == 0 else f"_{i}"
        obs["rgb" + sfx] = torch.randint(0, 256, (n, hs, ws, 3), generator=g, dtype=torch.uint8)
        obs["depth" + sfx] = torch.rand(n, hs, ws, 1, generator=g)
    obs["instruction"] = torch.randint(0, 50, (n, 9), generator=g)
    return obs


CROPS = [("rgb", (14, 16)), ("depth", (16, 18))]
STACK = [("rgb", ["rgb"] + [f"rgb_{i}" for i in range(1, 12)]),
         ("depth", ["depth"] + [f"depth_{i}" for i in range(1, 12)])]


# name -> (Hs, Ws, RESIZE_SHORTEST_EDGE.SIZE, SENSOR_CROPS); RxR itself is 480 x 640 -> 256 ->
# rgb 224 x 224 / depth 256 x 256 (checked at full size against the oracle in the GPU tier)
RESIZE_CASES = {
    "down": (30, 40, 16, [("rgb", (14, 14)), ("depth", (16, 16))]),
    "portrait": (45, 33, 20, [("rgb", (18, 16)), ("depth", (20, 20))]),
    "up": (12, 17, 19, [("rgb", (16, 20)), ("depth", (19, 19))]),
}


def main():
    ref = load_reference_transforms()
    # 1. centre crop of single-camera sensors (rxr_cma_en.yaml:27-30 order: crop after resize)
    o = {k: v for k, v in inputs().items() if k in ("rgb", "depth", "instruction")}
    out = ref.CenterCropperPerSensor(CROPS)(dict(o))
    np.savez_compressed(os.path.join(HERE, "obs_center_crop.npz"),
                        **{"out_" + k: v.numpy() for k, v in out.items()})
    # 2. ObsStack of the 12 cameras (r2r_waypoint/*.yaml: ENABLED_TRANSFORMS [ObsStack])
    out = ref.ObsStack(STACK)(dict(inputs()))
    np.savez_compressed(os.path.join(HERE, "obs_stack.npz"),
                        **{"out_" + k: v.numpy() for k, v in out.items()})
    # 3. both: stack, then crop the stacked sensors
    out = ref.CenterCropperPerSensor(CROPS)(ref.ObsStack(STACK)(dict(inputs())))
    np.savez_compressed(os.path.join(HERE, "obs_stack_crop.npz"),
                        **{"out_" + k: v.numpy() for k, v in out.items()})
    # 4. RxR order (rxr_cma_en.yaml:27-30): habitat's ResizeShortestEdge [3P, restated in
    # oracle/thirdparty.py from the published algorithm: F.interpolate(mode="area"), cast back to
    # the sensor dtype], then the reference's own CenterCropperPerSensor.  Landscape and portrait
    # frames, uint8 RGB and fp32 depth, non-integer scale factors down and up.
    for tag, (hs, ws, size, crops) in RESIZE_CASES.items():
        o = {k: v for k, v in inputs(seed=11, n=3, hs=hs, ws=ws).items()
             if k in ("rgb", "depth", "instruction")}
        rs = tp.resize_shortest_edge(dict(o), size)
        out = {"resized_" + k: v.numpy() for k, v in rs.items()}
        cr = ref.CenterCropperPerSensor(crops)(dict(rs))
        out.update({"out_" + k: v.numpy() for k, v in cr.items()})
        np.savez_compressed(os.path.join(HERE, f"obs_resize_{tag}.npz"), **out)
    print("wrote obs_center_crop.npz obs_stack.npz obs_stack_crop.npz obs_resize_*.npz")


if __name__ == "__main__":
    main()
