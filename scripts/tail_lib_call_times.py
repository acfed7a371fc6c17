"""GPU-paced time of every library call (and of the torch glue between them) of the CMA tail's forward
and backward at num_envs rows: the tail runs eagerly behind a spin kernel, an event is recorded before and
after every HipLib method; the stretch between two calls is booked as `torch glue`.
    python scripts/tail_lib_call_times.py [num_envs]"""
import os
import sys

os.environ["VLNCE_HIP_GRAPHS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vlnce_amd  # noqa: E402
from vlnce_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
net = policy.net
g = torch.Generator(device="cpu").manual_seed(0)


def mk(*shape, grad=False):
    return torch.randn(*shape, generator=g).to(dev).requires_grad_(grad)


ins, dep, rgb = mk(N, 80, 256, grad=True), mk(N, 16, 192, grad=True), mk(N, 16, 2112, grad=True)
act, h0 = mk(N, 32, grad=True), mk(N, 2, 512)
masks = torch.ones(N, dtype=torch.uint8, device=dev)
lib = _lib.get_lib()
marks = []  # (label, event)


def ev(label):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((label, e))


class Traced:
    def __getattr__(self, name):
        f = getattr(lib, name)
        if not callable(f) or name in ("gn_chunks", "conv2d_tiles", "conv2d_pack_bytes", "rnn_seq_supported",
                                       "conv2d_bn_workspace_bytes", "get_option", "conv2d_last_path"):
            return f

        def wrapped(*a, **k):
            shapes = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)][:3]
            ints = [v for v in a if isinstance(v, int)][:6]
            ev("<" + name)
            r = f(*a, **k)
            ev(">" + name + " " + str(shapes) + " " + str(ints))
            return r
        return wrapped


def once(record):
    marks.clear()
    _lib._LIB = Traced() if record else lib
    try:
        torch.cuda._sleep(60_000_000)
        ev("start")
        x, h = net._tail(ins, dep, rgb, act, h0, masks, static=(2048, 128))
        ev("fwd_end")
        gx, gh = torch.ones_like(x), torch.ones_like(h)
        ev("bwd_start")
        torch.autograd.backward([x, h], [gx, gh])
        ev("bwd_end")
        torch.cuda.synchronize()
    finally:
        _lib._LIB = lib


once(False)
once(True)
once(True)
rows = []
glue = {"fwd": [0.0, 0], "bwd": [0.0, 0]}
phase = "fwd"
tot = {"fwd": 0.0, "bwd": 0.0}
per = {}
for (l0, e0), (l1, e1) in zip(marks, marks[1:]):
    dt = e0.elapsed_time(e1) * 1e3
    if l1 == "bwd_start":
        phase = "bwd"
        continue
    tot[phase] += dt
    if l0.startswith("<") and l1.startswith(">"):
        rows.append((phase, l1[1:], dt))
        k = (phase, l1[1:].split(" ")[0])
        per.setdefault(k, [0.0, 0])
        per[k][0] += dt
        per[k][1] += 1
    else:
        glue[phase][0] += dt
        glue[phase][1] += 1
for ph in ("fwd", "bwd"):
    print(f"== {ph}: {tot[ph]:.0f} us GPU-paced; torch glue between library calls {glue[ph][0]:.0f} us in {glue[ph][1]} stretches")
    for (p_, name), (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        if p_ == ph:
            print(f"   {name:28s} x{n:3d} {t:8.1f} us")
print("== every call in order")
for ph, label, dt in rows:
    print(f"{ph} {dt:7.1f} us  {label}")
