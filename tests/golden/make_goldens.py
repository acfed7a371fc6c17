"""Generates tests/golden/*.npz by running the REAL reference policy classes
(imported from /root/reference through tools/oracle/shims.py) on the seeded
inputs of cases.py.  CPU container only; the reference never travels.

    python tests/golden/make_goldens.py [case ...]

H1 (BaseVLNCETrainer._update_agent, base_il_trainer.py:134-180) cannot be
imported as a class (habitat / tensorflow imports), so the function's source
is extracted from the reference file with `ast` at run time and executed
against the shimmed policy -- the reference's own statements, not a copy.
"""
import ast
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import cases  # noqa: E402
from oracle import thirdparty as tp  # noqa: E402
from tools.oracle import shims  # noqa: E402


def reference_update_agent(ref):
    path = os.path.join(shims.REFERENCE_ROOT, "vlnce_baselines/common/base_il_trainer.py")
    tree = ast.parse(open(path).read())
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "_update_agent":
            fn = node
    mod = ast.Module(body=[fn], type_ignores=[])
    scope = {"torch": torch, "F": F, "AuxLosses": ref.AuxLosses}
    exec(compile(mod, path, "exec"), scope)
    return scope["_update_agent"]


def reference_wddppo_update():
    """WDDPPO.update (ddppo_alg.py:38-149) extracted with ast and executed unbound."""
    path = os.path.join(shims.REFERENCE_ROOT, "vlnce_baselines/common/ddppo_alg.py")
    tree = ast.parse(open(path).read())
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "update":
            fn = node
    mod = ast.Module(body=[fn], type_ignores=[])
    from typing import Tuple

    scope = {"torch": torch, "Tuple": Tuple, "Tensor": torch.Tensor, "l1_loss": F.l1_loss}
    exec(compile(mod, path, "exec"), scope)
    return scope["update"]


class _NoStepOptimizer:
    def zero_grad(self):
        pass

    def step(self):
        pass


def reference_ppo_fn(policy, sample):
    upd = reference_wddppo_update()
    noop = lambda *a, **k: None
    rollouts = types.SimpleNamespace(recurrent_generator=lambda adv, nmb: iter([sample]))
    fake_self = types.SimpleNamespace(
        actor_critic=policy, ppo_epoch=1, num_mini_batch=1, optimizer=_NoStepOptimizer(),
        get_advantages=lambda r: sample[-1], before_backward=noop, after_backward=noop,
        before_step=noop, after_step=noop, **cases.PPO)
    return upd(fake_self, rollouts)


def main(names):
    torch.set_num_threads(8)
    ref = shims.load_reference()
    upd = reference_update_agent(ref)
    for name in names:
        case = cases.CASES[name]
        torch.manual_seed(0)
        policy, cfg = cases.build_policy(
            ref, case, tp.make_config, tp.make_spaces, tp.synth_state_dict
        )
        obs, prev, masks, extra = cases.build_inputs(case)

        def update_fn(policy, obs, prev, masks, targets, weights):
            fake_self = types.SimpleNamespace(
                policy=policy, config=cfg, device=torch.device("cpu"), optimizer=None
            )
            return upd(fake_self, obs, prev, masks, targets, weights, step_grad=False)

        if case["call"] == "ppo_update":
            # action components: the reference's own deterministic act() on these inputs,
            # with one STOP row so the pano mask is exercised
            with torch.no_grad():
                L = policy.net.num_recurrent_layers
                pa = {k: v.clone() for k, v in prev.items()}
                elems = policy.act(obs, extra["h0"][:, :L].contiguous(), pa, masks,
                                   deterministic=True)[2]
            for k, v in elems.items():
                extra["act_" + k] = v.clone()
            # spread the pano choices (the synthetic weights pick STOP everywhere); keep one
            # STOP row (= num_panos) so the distance/offset mask is exercised
            B = extra["act_pano"].size(0)
            extra["act_pano"] = (torch.tensor([[3], [12], [0], [7], [11], [5]]) if B == 6
                                 else (torch.arange(B) * 5 % 13).view(B, 1))
        outs = cases.run_case(policy, case, obs, prev, masks, extra, update_fn, ref.AuxLosses,
                              ppo_fn=reference_ppo_fn)
        path = os.path.join(HERE, name + ".npz")
        cases.save_case(path, name, obs, prev, masks, extra, outs)
        print(f"{name}: wrote {os.path.getsize(path)/1e3:.0f} kB;",
              {k: (tuple(v.shape) if hasattr(v, 'shape') else v) for k, v in outs.items()
               if k in ('logits', 'loss', 'value', 'actions')})


if __name__ == "__main__":
    main(sys.argv[1:] or list(cases.CASES))
