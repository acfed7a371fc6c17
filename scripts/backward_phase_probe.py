"""Inside the backward of a plain-loop step: when does the HOST reach each point, when has the GPU
executed up to it?  Points (tensor hooks, each records a host time stamp and an event on the stream
autograd is running that node on):
   logits grad ready (the trainer's loss backward is done)      -> action head backward, then
   tail output grad ready (start of the tail's backward graph)  -> tail backward graph replay
   instruction grad ready (tail's backward graph issued)        -> instruction encoder backward
   token-embedding grad ready (BPTT + parameter gradients issued)
   end of loss.backward()
    python scripts/backward_phase_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd import ops, streams  # noqa: E402
from vlnce_amd.aux_losses import AuxLosses  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
AuxLosses.activate()
batches = [bench.synth_batch(64, 256, 80, dev, seed=1 + 101 * i) for i in range(4)]
marks = None


def mark(name):
    if marks is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append((name, time.perf_counter(), e))


def hooked(name):
    def h(g):
        mark(name)
        return None
    return h


orig_tail_call = streams.GraphedTail.__call__


def tail_call(self, *tensors, static=()):
    mark("tail forward graph: replay call starts")
    if marks is not None and tensors[0].requires_grad:
        tensors[0].register_hook(hooked("instruction grad ready (tail backward graph issued)"))
    out = orig_tail_call(self, *tensors, static=static)
    if marks is not None and out[0].requires_grad:
        out[0].register_hook(hooked("tail output grad ready (action head backward done)"))
    mark("tail forward graph: replay call returned")
    return out


import vlnce_amd.cma_policy as cma_mod  # noqa: E402
import vlnce_amd.net_parts as net_parts  # noqa: E402

orig_enc = cma_mod.encode_three_branches


def enc(net, observations, device, **k):
    rgb_fwd = net.rgb_encoder.forward

    def rgb_marked(obs):
        y = rgb_fwd(obs)
        mark("RGB encoder issued (GPU: RGB trunk done)")
        return y

    net.rgb_encoder.forward = rgb_marked
    try:
        out = orig_enc(net, observations, device, **k)
    finally:
        del net.rgb_encoder.forward
    mark("three encoders joined on the main stream")
    return out


cma_mod.encode_three_branches = enc
import vlnce_amd.encoders.resnet_encoders as renc  # noqa: E402

orig_runner_call = renc._GraphRunner.__call__
runner_seen = []


def runner_call(self, x, key):
    first = marks is not None and not any(m[0].startswith("first trunk graph") for m in marks)
    if first:
        mark("first trunk graph: runner entered (host: Python of the step so far)")
    out = orig_runner_call(self, x, key)
    if first:
        mark("first trunk graph: replay call returned")
    return out


renc._GraphRunner.__call__ = runner_call


streams.GraphedTail.__call__ = tail_call
orig_embedding = ops.embedding


def embedding(tokens, weight, padding_idx=None):
    y = orig_embedding(tokens, weight, padding_idx)
    if marks is not None and y.requires_grad and y.dim() == 3:
        y.register_hook(hooked("token-embedding grad ready (instruction encoder backward issued)"))
    return y


ops.embedding = embedding


def step(i):
    obs, prev, masks, tgt, w = batches[i % 4]
    T, N = tgt.size()
    h = torch.zeros(N, policy.net.num_recurrent_layers, 512, device=dev)
    AuxLosses.clear()
    mark("step start")
    dist = policy.build_distribution(obs, h, prev, masks)
    mark("build_distribution returned")
    logits = dist.logits.view(T, N, -1)
    if marks is not None:
        logits.register_hook(hooked("logits grad ready (trainer's loss backward done)"))
    al = F.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    al = ((w * al).sum(0) / w.sum(0)).mean()
    loss = al + AuxLosses.reduce((w > 0).view(-1))
    mark("loss built, backward starts")
    loss.backward()
    mark("backward returned")
    opt.step()
    mark("Adam issued")
    opt.zero_grad()
    a, b = loss.item(), al.item()
    mark("step end (after the two .item() read-backs)")


for i in range(8):
    step(i)
torch.cuda.synchronize()
runs = []
for i in range(12):
    marks = []
    step(i)
    torch.cuda.synchronize()
    runs.append(marks)
marks = None
names = [m[0] for m in runs[0]]
assert all([m[0] for m in r] == names for r in runs), "hook order differs between steps"
n = len(runs)
print(f"plain-loop step with hooks and events in: {sum(r[-1][1] - r[0][1] for r in runs) / n * 1e3:.3f} ms/step (host clock)")
print(f"{'reached point':70s} {'host ms':>9s} {'GPU ms':>9s}   (since step start, mean of {n} steps)")
for k, name in enumerate(names):
    host = sum(r[k][1] - r[0][1] for r in runs) / n * 1e3
    gpu = sum(r[0][2].elapsed_time(r[k][2]) for r in runs) / n
    print(f"{name:70s} {host:9.3f} {gpu:9.3f}")
