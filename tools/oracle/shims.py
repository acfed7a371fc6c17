"""CPU-container-only helper: import the REAL reference model files from
/root/reference by injecting stand-ins for the third-party packages they
import (gym, habitat, habitat_baselines, torchvision -- all absent here and
not vendored by the reference).

Nothing in here ships to the GPU box at run time: tests marked `gpu`,
`__graft_entry__.smoke()` and `bench.py` never import this module.  It is
used only by tests/golden/make_goldens.py and by the CPU-only test that
checks oracle/policy_cpu.py against the imported reference.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vlnce_baselines", "models"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Inject shim modules and make `vlnce_baselines.models.*` importable
    without executing vlnce_baselines/__init__.py (which pulls trainers ->
    lmdb / tensorflow / jsonlines)."""
    if "vlnce_baselines.models.cma_policy" in sys.modules:
        return
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    import torch
    from oracle import thirdparty as tp

    # torch>=2 validates distribution args by default; the reference targets
    # torch 1.6 and calls Normal.cdf(float) (models/utils.py:61).
    torch.distributions.Distribution.set_default_validate_args(False)

    spaces = _mod("gym.spaces", Box=tp.Box, Dict=tp.Dict, Discrete=tp.Discrete, Space=tp.Space)
    _mod("gym", Space=tp.Space, spaces=spaces)

    _mod("habitat", Config=tp.Config)
    _mod("habitat.config", Config=tp.Config)
    _mod("habitat.config.default", Config=tp.Config)
    _mod("habitat.core")
    _mod("habitat.core.simulator", Observations=dict)

    class _Registry:
        _policies = {}

        @classmethod
        def register_policy(cls, to_register=None, *, name=None):
            def wrap(c):
                cls._policies[name or c.__name__] = c
                return c

            return wrap if to_register is None else wrap(to_register)

        @classmethod
        def get_policy(cls, name):
            return cls._policies.get(name)

    _mod("habitat_baselines")
    _mod("habitat_baselines.common")
    _mod(
        "habitat_baselines.common.baseline_registry",
        BaselineRegistry=_Registry,
        baseline_registry=_Registry,
    )
    _mod("habitat_baselines.utils")
    _mod("habitat_baselines.utils.common", CategoricalNet=tp.CategoricalNet)
    _mod("habitat_baselines.rl")
    _mod("habitat_baselines.rl.ppo")
    _mod(
        "habitat_baselines.rl.ppo.policy",
        Net=tp.Net,
        Policy=tp.Policy,
        CriticHead=tp.CriticHead,
    )
    _mod("habitat_baselines.rl.models")
    _mod(
        "habitat_baselines.rl.models.rnn_state_encoder",
        build_rnn_state_encoder=tp.build_rnn_state_encoder,
    )
    _mod("habitat_baselines.rl.ddppo")
    resnet_mod = _mod(
        "habitat_baselines.rl.ddppo.policy.resnet",
        resnet18=tp.gn_resnet18,
        resnet50=tp.gn_resnet50,
    )
    _mod("habitat_baselines.rl.ddppo.policy", resnet=resnet_mod)
    _mod(
        "habitat_baselines.rl.ddppo.policy.resnet_policy",
        ResNetEncoder=tp.ResNetEncoder,
    )
    tvm = _mod("torchvision.models", resnet18=tp.tv_resnet18, resnet50=tp.tv_resnet50)
    _mod("torchvision", models=tvm)

    # package objects with __path__ only -> sub-modules load from the reference
    # tree, package __init__ files are never executed.
    for pkg, rel in [
        ("vlnce_baselines", "vlnce_baselines"),
        ("vlnce_baselines.common", "vlnce_baselines/common"),
        ("vlnce_baselines.models", "vlnce_baselines/models"),
        ("vlnce_baselines.models.encoders", "vlnce_baselines/models/encoders"),
    ]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
        sys.modules[pkg] = m
    sys.dont_write_bytecode = True  # never write into /root/reference


def load_reference():
    """Returns a namespace with the reference policy classes + AuxLosses."""
    install()
    import importlib

    ns = types.SimpleNamespace()
    ns.Seq2SeqPolicy = importlib.import_module(
        "vlnce_baselines.models.seq2seq_policy"
    ).Seq2SeqPolicy
    ns.CMAPolicy = importlib.import_module("vlnce_baselines.models.cma_policy").CMAPolicy
    ns.WaypointPolicy = importlib.import_module(
        "vlnce_baselines.models.waypoint_policy"
    ).WaypointPolicy
    ns.AuxLosses = importlib.import_module("vlnce_baselines.common.aux_losses").AuxLosses
    ns.utils = importlib.import_module("vlnce_baselines.models.utils")
    return ns
