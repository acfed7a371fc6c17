"""Per-output error of the HIP policy against a committed reference golden (tests/golden/<case>.npz):
    python scripts/golden_errors.py cma_update_n64_256
prints, for every expected output, max |d|, max |expected| and the worst relative error."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import numpy as np  # noqa: E402

import cases  # noqa: E402
import vlnce_amd  # noqa: E402
from oracle import thirdparty as tp  # noqa: E402
from test_policy_gpu import hip_ppo, hip_update, to_dev  # noqa: E402

name = sys.argv[1]
case = cases.CASES[name]
obs, prev, masks, extra, gold = cases.load_case(os.path.join(REPO, "tests", "golden", name + ".npz"))
policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                               tp.synth_state_dict)
policy.to("cuda:0")
outs = cases.run_case(policy, case, to_dev(obs), to_dev(prev), to_dev(masks), to_dev(extra),
                      hip_update, vlnce_amd.AuxLosses, ppo_fn=hip_ppo)
for k, g in gold.items():
    if isinstance(g, np.ndarray) or not g.dtype.is_floating_point:
        continue
    o = outs[k].double().cpu()
    g = g.double()
    err = (o - g).abs()
    i = int(err.argmax())
    rel = (err / (1e-4 + 1e-4 * g.abs())).max().item()
    name_i = ""
    if k == "grad_norms":
        name_i = str(gold["grad_names"][i])
    print(f"{k:44s} max|d| {err.max().item():.3e}  at value {g.flatten()[i].item():+.4e}  "
          f"max|g| {g.abs().max().item():.3e}  err/tol {rel:.2f} {name_i}")
