"""Per phase of a plain-loop step: when has the HOST finished issuing it, when has the GPU finished
executing it (steady state, no syncs other than the step's own)?  Tells which of the two paces the
stretch behind the action head's host read-back.

    python scripts/host_vs_gpu_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd.aux_losses import AuxLosses  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
AuxLosses.activate()
batches = [bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i) for i in range(4)]
NAMES = ["build_distribution (ends with the host read-back)", "loss", "backward", "Adam", "zero_grad"]


def step(i, stamps=None):
    obs, prev, masks, tgt, w = batches[i % 4]
    T, N = tgt.size()
    marks = []

    def mark():
        if stamps is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((time.perf_counter(), e))

    mark()
    h = torch.zeros(N, policy.net.num_recurrent_layers, 512, device=dev)
    AuxLosses.clear()
    dist = policy.build_distribution(obs, h, prev, masks)
    mark()
    logits = dist.logits.view(T, N, -1)
    al = F.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    al = ((w * al).sum(0) / w.sum(0)).mean()
    loss = al + AuxLosses.reduce((w > 0).view(-1))
    mark()
    loss.backward()
    mark()
    opt.step()
    mark()
    opt.zero_grad()
    mark()
    if stamps is not None:
        stamps.append(marks)


for i in range(8):
    step(i)
torch.cuda.synchronize()
all_marks = []
t0 = time.perf_counter()
for i in range(12):
    step(i, all_marks)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 12 * 1e3:.3f} ms/step with the stamps in")
host = [0.0] * 5
gpu = [0.0] * 5
lag = [0.0] * 6
for marks in all_marks[2:]:
    for k in range(5):
        host[k] += (marks[k + 1][0] - marks[k][0]) * 1e3
        gpu[k] += marks[k][1].elapsed_time(marks[k + 1][1])
n = len(all_marks) - 2
print(f"{'phase':52s} {'host issue ms':>14s} {'GPU stamps ms':>14s}")
for k in range(5):
    print(f"{NAMES[k]:52s} {host[k] / n:14.3f} {gpu[k] / n:14.3f}")
print(f"{'sum':52s} {sum(host) / n:14.3f} {sum(gpu) / n:14.3f}")
