"""How the plain-loop step time depends on where the host synchronises: the Categorical's argument
validation after the tail forward (as the reference: a host sync), none at all, and the
reference trainer's loss.item() at the end of every update.  Also counts device allocations
(cudaMalloc segments) inside the timed loop.  python scripts/sync_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd import ops, policy as _pol  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402
from vlnce_amd.utils import CustomFixedCategorical  # noqa: E402

NOVALIDATE = [False]
_orig = _pol.CategoricalNet.forward


def _forward(self, x):   # the product's forward with the argument validation (a host sync) switchable
    if NOVALIDATE[0]:
        return CustomFixedCategorical(logits=ops.linear(x, self.linear.weight, self.linear.bias),
                                      validate_args=False)
    return _orig(self, x)


_pol.CategoricalNet.forward = _forward

dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
vlnce_amd.AuxLosses.activate()
batches = [bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i) for i in range(4)]


def loop(n, item):
    for i in range(n):
        obs, prev, masks, tgt, w = batches[i % 4]
        loss, al, aux = update_agent(policy, opt, obs, prev, masks, tgt, w, 512)
        if item:
            pass


for novalidate in (False, True):
    for item in (False, True):
        NOVALIDATE[0] = novalidate
        loop(6, item)
        torch.cuda.synchronize()
        s0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        loop(20, item)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        s1 = torch.cuda.memory_stats()
        print(f"validate sync {'off' if novalidate else 'on '}, loss.item() per step {item!s:5}: {dt:.3f} ms/step; "
              f"device mallocs in the loop {s1['num_device_alloc'] - s0['num_device_alloc']}, "
              f"frees {s1['num_device_free'] - s0['num_device_free']}, reserved {s1['reserved_bytes.all.current'] / 2**30:.2f} GiB",
              flush=True)
