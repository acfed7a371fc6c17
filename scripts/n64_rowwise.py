"""Row-wise comparison of intermediate gradients, HIP policy vs CPU oracle, on the N=64 golden inputs."""
import os
import sys

os.environ.setdefault("VLNCE_HIP_GRAPHS", "0")
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
from vlnce_amd import ops  # noqa: E402

torch.set_num_threads(32)
case = dict(cases.CASES["cma_update_n64_256"])
obs, prev, masks, extra = cases.build_inputs(case)
if os.environ.get("ALL_MASKS_ONE"):
    masks = torch.ones_like(masks)
hip, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces, tp.synth_state_dict)
ref, _ = cases.build_policy(oc, case, tp.make_config, tp.make_spaces, tp.synth_state_dict)
hip.to("cuda:0")
rec = []
orig_linear, orig_gru = ops.linear, ops.gru_cell


def lin(*a, **k):
    y = orig_linear(*a, **k)
    if y.requires_grad:
        y.retain_grad()
        rec.append(("linear", y))
    return y


def gru(*a, **k):
    y = orig_gru(*a, **k)
    y.retain_grad()
    rec.append(("gru", y))
    return y


ops.linear, ops.gru_cell = lin, gru
keep = {}


def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        o.retain_grad()
        keep[name] = o
    return f


n = ref.net
n.second_state_compress.register_forward_hook(hook("compress"))
n.state_encoder.register_forward_hook(hook("state"))
n.second_state_encoder.register_forward_hook(hook("x_out"))
n.text_q.register_forward_hook(hook("text_q"))
n.state_q.register_forward_hook(hook("state_q"))
n.rgb_linear.register_forward_hook(hook("rgb_linear"))
vlnce_amd.AuxLosses.activate()
oc.AuxLosses.activate()
hip_update(hip, to_dev(obs), to_dev(prev), to_dev(masks), to_dev(extra["targets"]), to_dev(extra["weights"]))
oc.il_update(ref, None, obs, prev, masks, extra["targets"], extra["weights"], 512, step_grad=False)
print("recorded HIP ops:", [(k, tuple(v.shape)) for k, v in rec])
lins = [v for k, v in rec if k == "linear" and v.dim() == 2 and v.size(0) == 64]
grus = [v for k, v in rec if k == "gru"]
# order of the 64-row linears in _CMATail.forward: rgb_linear, depth_linear, gi1, gh1, state_q, text_q, compress, gi2, gh2
pairs = {"rgb_linear": lins[0], "state_q": lins[4], "text_q": lins[5], "compress": lins[6],
         "state": grus[0], "x_out": grus[1]}
for name, h in pairs.items():
    r = keep[name]
    fv = (h.detach().cpu() - r.detach()).abs().max().item()
    gh, gr = h.grad.cpu().double(), r.grad.double()
    per = (gh - gr).norm(dim=1) / (gr.norm(dim=1) + 1e-30)
    bad = (per > 1e-3).nonzero().flatten().tolist()
    print(f"{name:12s} fwd max|d| {fv:.2e}  grad rel {((gh - gr).norm() / gr.norm()).item():.2e}  bad rows {bad[:20]} ({len(bad)})"
          f"  worst row {int(per.argmax())} rel {per.max().item():.2e}")
    if name in ("compress", "x_out") and bad:
        r0 = bad[0]
        print("   row", r0, "hip", gh[r0, :6].tolist(), "\n   ref", gr[r0, :6].tolist(), "\n   ratio", (gh[r0, :6] / gr[r0, :6]).tolist())
print("masks[:4]", masks[:4].flatten().tolist(), "weights[:4]", extra["weights"][0, :4].tolist())
