"""Localises a gradient discrepancy of the trainable tail at a given batch: the same CMA update (trunk
outputs fed through the reference's bypass keys) once on the GPU through the HIP library and once on the
CPU through tests/hostsim.py (plain torch), with every ops.* call of the tail recorded; prints the
gradient of each recorded output, last forward op first.
    python scripts/tail_grad_bisect.py [num_envs]"""
import os
import sys

os.environ["VLNCE_HIP_GRAPHS"] = "0"
os.environ["VLNCE_SIDE_STREAMS"] = "0"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402

import hostsim  # noqa: E402
import vlnce_amd  # noqa: E402
from oracle import thirdparty as tp  # noqa: E402
from vlnce_amd import _lib, ops  # noqa: E402
from vlnce_amd.il_harness import update_agent  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(3)
obs = {"rgb_features": torch.rand(N, 2048, 4, 4, generator=g) * 2.0,
       "depth_features": torch.rand(N, 128, 4, 4, generator=g),
       "instruction": torch.zeros(N, 200, dtype=torch.long)}
for i in range(N):
    L = 80 - (i % 6)
    obs["instruction"][i, :L] = torch.randint(1, 2504, (L,), generator=g)
prev = torch.randint(0, 4, (N, 1), generator=g)
masks = (torch.rand(N, 1, generator=g) > 0.1).to(torch.uint8)
tgt = torch.randint(0, 4, (1, N), generator=g)
w = torch.rand(1, N, generator=g) + 0.5

NAMES = ["linear", "attention", "gru_cell", "mean_rows", "mask_rows", "rnn_layer", "embedding", "action_head"]
orig = {n: getattr(ops, n) for n in NAMES}


def run(dev):
    rec = []

    def wrap(name):
        def f(*a, **k):
            out = orig[name](*a, **k)
            outs = out if isinstance(out, (tuple, list)) else (out,)
            flat = []
            for o in outs:
                flat += list(o) if isinstance(o, (tuple, list)) else [o]
            for j, o in enumerate(flat):
                if isinstance(o, torch.Tensor) and o.requires_grad and o.dtype == torch.float32:
                    o.retain_grad()
                    rec.append((f"{len(rec):02d} {name}[{j}] {tuple(o.shape)}", o))
            return out
        return f

    for n in NAMES:
        setattr(ops, n, wrap(n))
    import vlnce_amd.policy as pol_mod
    try:
        torch.manual_seed(0)
        pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256))
        pol.load_state_dict(tp.synth_state_dict(pol))
        pol.to(dev)
        vlnce_amd.AuxLosses.activate()
        mv = lambda t: t.to(dev)  # noqa: E731
        loss = update_agent(pol, None, {k: mv(v) for k, v in obs.items()}, mv(prev), mv(masks), mv(tgt), mv(w),
                            512, step_grad=False)
    finally:
        for n in NAMES:
            setattr(ops, n, orig[n])
    grads = [(n, (o.grad.detach().cpu().double() if o.grad is not None else None), o.detach().cpu().double())
             for n, o in rec]
    pg = {n: p.grad.detach().cpu().double() for n, p in pol.named_parameters() if p.grad is not None}
    return loss, grads, pg


l_gpu, g_gpu, p_gpu = run("cuda:0")
_lib._LIB = hostsim.HostSim()
l_cpu, g_cpu, p_cpu = run("cpu")
print("loss", l_gpu[0], l_cpu[0])
assert len(g_gpu) == len(g_cpu), (len(g_gpu), len(g_cpu))
for (n, a, va), (n2, b, vb) in reversed(list(zip(g_gpu, g_cpu))):
    fe = (va - vb).norm().item() / (vb.norm().item() + 1e-30)
    if a is None or b is None:
        print(f"{n:44s} fwd rel {fe:.2e}  grad None")
        continue
    ge = (a - b).norm().item() / (b.norm().item() + 1e-30)
    rows = ""
    if ge > 1e-4 and a.dim() >= 2:
        per = (a - b).reshape(a.size(0), -1).norm(dim=1) / (b.reshape(b.size(0), -1).norm(dim=1) + 1e-30)
        bad = (per > 1e-3).nonzero().flatten().tolist()
        rows = f" bad leading-index rows {bad[:12]}{'...' if len(bad) > 12 else ''} ({len(bad)})"
    print(f"{n:44s} fwd rel {fe:.2e}  grad rel {ge:.2e}{rows}")
for n in p_gpu:
    e = (p_gpu[n] - p_cpu[n]).norm().item() / (p_cpu[n].norm().item() + 1e-30)
    if e > 1e-4:
        print(f"param {n:56s} grad rel {e:.2e}")
