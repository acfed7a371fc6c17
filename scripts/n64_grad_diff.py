"""HIP policy vs the CPU oracle on the N=64 golden's inputs: per-parameter gradient differences and the
encoder outputs (diagnostic for tests/golden/cma_update_n64_256.npz)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import torch  # noqa: E402

import cases  # noqa: E402
import vlnce_amd  # noqa: E402
from oracle import policy_cpu as oc  # noqa: E402
from oracle import thirdparty as tp  # noqa: E402
from test_policy_gpu import hip_update, to_dev  # noqa: E402

torch.set_num_threads(32)
name = sys.argv[1] if len(sys.argv) > 1 else "cma_update_n64_256"
case = dict(cases.CASES[name])
if len(sys.argv) > 2:
    n = int(sys.argv[2])
    case["N"], case["lengths"] = n, case["lengths"][:n]
obs, prev, masks, extra = cases.build_inputs(case)
hip, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces, tp.synth_state_dict)
ref, _ = cases.build_policy(oc, case, tp.make_config, tp.make_spaces, tp.synth_state_dict)
hip.to("cuda:0")
with torch.no_grad():
    for k, (he, re_) in {"depth": (hip.net.depth_encoder, ref.net.depth_encoder),
                         "rgb": (hip.net.rgb_encoder, ref.net.rgb_encoder)}.items():
        a = he(to_dev(obs)).cpu()
        b = re_(obs)
        print(k, "encoder out", tuple(a.shape), "max|d|", (a - b).abs().max().item(), "max|ref|", b.abs().max().item())
vlnce_amd.AuxLosses.activate()
oc.AuxLosses.activate()
hip_update(hip, to_dev(obs), to_dev(prev), to_dev(masks), to_dev(extra["targets"]), to_dev(extra["weights"]))
oc.il_update(ref, None, obs, prev, masks, extra["targets"], extra["weights"], 512, step_grad=False)
rows = []
rp = dict(ref.named_parameters())
for n_, p in hip.named_parameters():
    if p.grad is None:
        continue
    g, r = p.grad.cpu().double(), rp[n_].grad.double()
    rows.append(((g - r).norm().item() / (r.norm().item() + 1e-30), (g - r).abs().max().item(), r.abs().max().item(), n_, tuple(g.shape)))
for rel, mx, rm, n_, sh in sorted(rows, reverse=True)[:int(os.environ.get('TOP', '14'))]:
    print(f"{n_:60s} {str(sh):14s} rel-norm-err {rel:.3e} max|d| {mx:.3e} max|ref| {rm:.3e}")
g = hip.net.depth_linear[1].weight.grad.cpu()
r = ref.net.depth_linear[1].weight.grad
d = (g - r).abs()
print("depth_linear dW: rows with err > 1e-6:", (d.max(1).values > 1e-6).sum().item(), "of", d.size(0),
      "cols:", (d.max(0).values > 1e-6).sum().item(), "of", d.size(1))
bad = (d > 1e-6).nonzero()
print("first bad (row, col):", bad[:8].tolist(), "last:", bad[-4:].tolist())
c = (d.max(0).values > 1e-6).nonzero().flatten()
if c.numel():
    print("bad col range", c.min().item(), c.max().item(), "c % 16 histogram (NCHW col = ch*16 + pos):",
          torch.bincount(c % 16, minlength=16).tolist())
