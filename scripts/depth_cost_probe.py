"""What the depth trunk costs a CMA update step at num_envs 64: the plain trainer loop (bench.py's
`value`) (a) as is and (b) with `depth_features` handed in (the cached-feature bypass of
resnet_encoders.py:70-72: no depth trunk at all).  (Round 6 also ran it with the GroupNorm finalize
launches, and with finalize + apply, skipped through a timing-only switch in ops.conv_group_norm_act
that is not in the tree: profiles/r06_k_depth_trunk_cost_probe.txt.)

    python scripts/depth_cost_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402


def run(mode, steps=40):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
    opt = torch.optim.Adam(pol.parameters(), lr=2.5e-4)
    vlnce_amd.AuxLosses.activate()
    batches = [bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i) for i in range(4)]
    if mode == "cached_depth":
        with torch.no_grad():
            for b in batches:
                b[0]["depth_features"] = pol.net.depth_encoder.trunk_features(b[0]).clone()
    for i in range(6):
        o, p, m, t, w = batches[i % 4]
        update_agent(pol, opt, o, p, m, t, w, 512)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        o, p, m, t, w = batches[i % 4]
        update_agent(pol, opt, o, p, m, t, w, 512)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


if __name__ == "__main__":
    from vlnce_amd.distributed import bind_host_threads_to_gpu_socket
    bind_host_threads_to_gpu_socket(0, scope="l3")
    for rep in range(2):
        for mode in ("as is", "cached_depth"):
            print(f"{mode:14s} {run(mode):.3f} ms/step", flush=True)
