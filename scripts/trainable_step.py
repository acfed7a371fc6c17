"""A few CMA update steps with TRAINABLE visual encoders (MODEL.*_ENCODER.trainable = True) and nothing
else, for `rocprofv3 --kernel-trace` (profiles/jobs/r6_25.sh): prints the dispatch count of the
warm-up so that scripts/rocpd_stats.py can skip it.

    python scripts/trainable_step.py [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = vlnce_amd.make_config("CMAPolicy", **{"RGB_ENCODER.trainable": True, "DEPTH_ENCODER.trainable": True})
pol = vlnce_amd.build_model(cfg, *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(pol.parameters(), lr=2.5e-4)
vlnce_amd.AuxLosses.activate()
b = [bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i) for i in range(2)]
for i in range(3):
    update_agent(pol, opt, *b[i % 2], 512)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    update_agent(pol, opt, *b[i % 2], 512)
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step over {steps} steps")
