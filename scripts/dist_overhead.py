"""Where the data-parallel wrapper's per-step cost goes, on one GPU with a 1-rank RCCL group.
    python scripts/dist_overhead.py   (prints ms/step for 4 variants)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd import distributed as D  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
batch = bench.synth_batch(64, 256, 80, dev, seed=1)


class _Done:
    def wait(self):
        pass


def run(variant):
    torch.manual_seed(0)
    policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"),
                                   *vlnce_amd.make_spaces(256, 256)).to(dev)
    opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
    vlnce_amd.AuxLosses.activate()
    hook = None
    red = None
    if variant != "none":
        red = D.GradientAllReducer(policy)
        hook = red.finish
        if variant == "hooks_only":
            red._launch = lambda b: setattr(b, "work", _Done())
            red.finish = lambda: [setattr(b, "pending", len(b.params)) or setattr(b, "work", None)
                                  for b in red.buckets]
            red.avg = True
            hook = red.finish
    real_ar = dist.all_reduce_coalesced
    if variant == "no_collective":
        dist.all_reduce_coalesced = lambda *a, **k: _Done()

    def step():
        obs, prev, masks, tgt, w = batch
        update_agent(policy, opt, obs, prev, masks, tgt, w, 512, grad_hook=hook)

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 15 * 1e3
    dist.all_reduce_coalesced = real_ar
    if red is not None:
        red.remove()
    print(f"{variant:14s} {ms:7.2f} ms/step", flush=True)


for v in ("none", "hooks_only", "no_collective", "full", "none"):
    run(v)
dist.destroy_process_group()
