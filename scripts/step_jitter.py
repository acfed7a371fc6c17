"""Per-step wall time of the plain trainer loop (every step ends with the read-backs, so the
host clock per step is the step): percentiles and 20-step means of one process.  Run several
processes back to back to separate slow STEPS (host hiccups) from slow PROCESSES (placement,
clocks).  Diagnostic only."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

dev = torch.device("cuda:0")
from vlnce_amd.distributed import bind_host_threads_to_gpu_socket, gpu_numa_node  # noqa: E402

mode = os.environ.get("BIND", "none")  # none | local | remote
local = gpu_numa_node(0)
bound = None
if mode in ("local", "remote") and local is not None:
    bound = bind_host_threads_to_gpu_socket(0, node=local if mode == "local" else 1 - local)
if mode == "l3" and local is not None:  # the socket first, then one L3 domain (CCD) of it
    from vlnce_amd.distributed import _parse_cpulist

    bound = bind_host_threads_to_gpu_socket(0, node=local)
    first = min(os.sched_getaffinity(0))
    with open(f"/sys/devices/system/cpu/cpu{first}/cache/index3/shared_cpu_list") as f:
        ccd = _parse_cpulist(f.read()) & os.sched_getaffinity(0)
    for tid in os.listdir("/proc/self/task"):
        os.sched_setaffinity(int(tid), ccd)
    print(f"one L3 domain: {sorted(ccd)}")
print(f"GPU on NUMA node {local}; BIND={mode}: threads bound to node {bound}")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
vlnce_amd.AuxLosses.activate()
batches = [bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i) for i in range(4)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ts = []
cpus = []
_libc = ctypes.CDLL(None)
for k in range(10 + n):
    obs, prev, masks, tgt, w = batches[k % 4]
    t0 = time.perf_counter()
    update_agent(policy, opt, obs, prev, masks, tgt, w, 512)
    ts.append(1e3 * (time.perf_counter() - t0))
    cpus.append(_libc.sched_getcpu())
ts = torch.tensor(ts[10:])
q = torch.quantile(ts, torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0]))
print("mean %.3f  min/p10/p50/p90/p99/max " % ts.mean().item() + " ".join("%.3f" % v for v in q.tolist()))
print("20-step means: " + " ".join("%.2f" % ts[i:i + 20].mean().item() for i in range(0, n, 20)))
cpus = cpus[10:]
print("cpu of the issuing thread per 20 steps: " + " ".join(",".join(str(c) for c in sorted(set(cpus[i:i + 20]))) for i in range(0, n, 20)))
