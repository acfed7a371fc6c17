"""CPU tier: the oracle (oracle/policy_cpu.py) against the committed golden
vectors, which were produced by the REAL reference classes
(tests/golden/make_goldens.py).  Tolerance 1e-5 abs (fp32, same torch CPU
kernels underneath; differences come only from op re-association)."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import policy_cpu as oc
from oracle import thirdparty as tp

GOLD = os.path.join(os.path.dirname(__file__), "golden")
torch.distributions.Distribution.set_default_validate_args(False)


def _oracle_update(policy, obs, prev, masks, targets, weights):
    hs = policy.net.model_config.STATE_ENCODER.hidden_size
    return oc.il_update(policy, None, obs, prev, masks, targets, weights, hs, step_grad=False)


def _oracle_ppo(policy, sample):
    return oc.ppo_update(policy, None, sample, step_grad=False, **cases.PPO)


def compare(outs, gold, atol, rtol=1e-5):
    assert set(gold) <= set(outs), set(gold) - set(outs)
    for k, g in gold.items():
        o = outs[k]
        if isinstance(g, np.ndarray):  # names
            assert list(g) == list(o), k
            continue
        o = o if isinstance(o, torch.Tensor) else torch.as_tensor(o)
        assert tuple(o.shape) == tuple(g.shape), (k, o.shape, g.shape)
        if not g.dtype.is_floating_point:
            assert torch.equal(o.to(g.dtype), g), k
            continue
        assert torch.equal(torch.isnan(o), torch.isnan(g)), k  # Categorical.variance is NaN
        o, g = torch.nan_to_num(o), torch.nan_to_num(g)
        err = (o.double() - g.double()).abs()
        tol = atol + rtol * g.double().abs()
        assert bool((err <= tol).all()), (k, float(err.max()), float(g.abs().max()))


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference_golden(name):
    case = cases.CASES[name]
    obs, prev, masks, extra, gold = cases.load_case(os.path.join(GOLD, name + ".npz"))
    # stored inputs == regenerated inputs (guards the fixture recipe)
    obs2, prev2, masks2, extra2 = cases.build_inputs(case)
    for k in obs:
        assert torch.equal(obs[k].float(), obs2[k].float()), k
    policy, _ = cases.build_policy(oc, case, tp.make_config, tp.make_spaces, tp.synth_state_dict)
    outs = cases.run_case(policy, case, obs, prev, masks, extra, _oracle_update, oc.AuxLosses,
                          ppo_fn=_oracle_ppo)
    compare(outs, gold, atol=2e-5)


def test_state_dict_keys_match_reference_counts():
    # 497 keys (Seq2Seq) / 521 (CMA) observed on the reference
    for pol, n in (("Seq2SeqPolicy", 497), ("CMAPolicy", 521)):
        cfg = tp.make_config(pol)
        sp = tp.make_spaces(256, 256)
        p = getattr(oc, pol).from_config(cfg, *sp)
        assert len(p.state_dict()) == n
