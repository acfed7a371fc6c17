"""How far apart are two fp32-class implementations of the visual trunks at the bench geometry?
Runs tests/cross_kernel_worker.py under several convolution configurations (child processes) and
prints max |a - b| / max |b| against the fp32-MFMA reference run, including the fp32-MFMA kernel
against ITSELF with a different tile shape (pure summation-order noise: the floor).

    python scripts/cross_kernel_floor.py
"""
import os
import subprocess
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(os.path.dirname(HERE), "tests", "cross_kernel_worker.py")
CONFIGS = {
    "f32 (reference)": {"VLNCE_CONV_MATH": "f32"},
    "f32, 64x64 tiles": {"VLNCE_CONV_MATH": "f32", "VLNCE_IGEMM_TILE": "3"},
    "f32, inputs moved by 1 ulp": {"VLNCE_CONV_MATH": "f32", "VLNCE_TEST_PERTURB": "1"},
    "planes: x3 only": {"VLNCE_P3": "0"},
    "planes: p3 KxK + x3": {"VLNCE_U3": "0"},
    "planes: default (p3 + u3 + x3)": {},
}


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def rms(a, b):
    return ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()


outs = {}
with tempfile.TemporaryDirectory() as d:
    for i, (name, env) in enumerate(CONFIGS.items()):
        out = os.path.join(d, f"o{i}.pt")
        e = {k: v for k, v in os.environ.items() if not k.startswith("VLNCE_")}
        # (the fp32-MFMA kernel with another tile shape sums in the same k order: bit-identical;
        # a one-rounding change of the INPUT is the honest yardstick for "two fp32 computations")
        e.update(env)
        r = subprocess.run([sys.executable, WORKER, "cma", out], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = torch.load(out)
ref = outs["f32 (reference)"]
print(f"{'configuration':34s} {'rgb max':>9s} {'rgb rms':>9s} {'depth max':>9s} {'bn max':>9s} {'loss':>9s} {'grad_q':>9s}")
for name, o in outs.items():
    bn = max(rel(o[k], ref[k]) for k in o if k.startswith("bn/"))
    print(f"{name:34s} {rel(o['rgb_trunk'], ref['rgb_trunk']):9.2e} {rms(o['rgb_trunk'], ref['rgb_trunk']):9.2e} "
          f"{rel(o['depth_trunk'], ref['depth_trunk']):9.2e} {bn:9.2e} {rel(o['loss'], ref['loss']):9.2e} "
          f"{rel(o['grad_state_q'], ref['grad_state_q']):9.2e}")
