"""Host-side fixes from the round-1 review: ImageNet-weights path of the frozen RGB trunk,
DD-PPO checkpoint files with a pickled config next to the state_dict, kernel launches on the
tensors' device rather than the process's current device, LRU graph cache, length buckets."""
import types
import warnings

import pytest
import torch

import vlnce_amd
from vlnce_amd import _lib, streams
from vlnce_amd.encoders import resnet_encoders as enc


def _torchvision_named(trunk):
    """the trunk's parameters / buffers under torchvision's own resnet key names"""
    inv = {v: k for k, v in enc._TV_CHILD_INDEX.items()}
    out = {}
    for k, v in trunk.state_dict().items():
        head, _, rest = k.partition(".")
        out[inv[head] + "." + rest] = torch.randn_like(v) if v.is_floating_point() else v.clone()
    out["fc.weight"] = torch.randn(1000, trunk.final_channels)
    out["fc.bias"] = torch.randn(1000)
    return out


@pytest.mark.parametrize("cls", [enc.TorchVisionResNet18, enc.TorchVisionResNet50])
def test_torchvision_state_dict_loads_strictly(cls, tmp_path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        probe = cls(128, spatial_output=True)
    sd = _torchvision_named(probe.cnn)
    path = tmp_path / "resnet.pth"
    torch.save(sd, path)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # a trunk with weights must not warn
        m = cls(128, spatial_output=True, pretrained_weights=str(path))
    assert torch.equal(m.cnn[0].weight, sd["conv1.weight"])
    assert torch.equal(m.cnn[7][0].conv1.weight, sd["layer4.0.conv1.weight"])
    assert torch.equal(m.cnn[1].running_var, sd["bn1.running_var"])
    assert not any(p.requires_grad for p in m.cnn.parameters())
    bad = dict(sd)
    bad.pop("layer2.0.conv1.weight")
    with pytest.raises(RuntimeError):
        m.load_torchvision_weights(bad)


def test_frozen_random_trunk_warns_loudly():
    with pytest.warns(UserWarning, match="RANDOMLY initialised"):
        enc.TorchVisionResNet18(128)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        enc.TorchVisionResNet18(128, trainable=True)


def test_config_key_reaches_the_encoder(tmp_path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        probe = enc.TorchVisionResNet50(256, spatial_output=True)
    sd = _torchvision_named(probe.cnn)
    path = tmp_path / "r50.pth"
    torch.save(sd, path)
    cfg = vlnce_amd.make_config("CMAPolicy", **{"RGB_ENCODER.pretrained_weights": str(path)})
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pol = vlnce_amd.build_model(cfg, *vlnce_amd.make_spaces(64, 64))
    assert torch.equal(pol.net.rgb_encoder.cnn[0].weight, sd["conv1.weight"])


class _PickledConfig:  # stands in for the yacs node the published DD-PPO files carry
    def __init__(self):
        self.RL = types.SimpleNamespace(PPO=types.SimpleNamespace(hidden_size=512))


def test_ddppo_checkpoint_with_pickled_config_loads(tmp_path):
    spaces = vlnce_amd.make_spaces(64, 64)[0]
    src = enc.VlnResnetDepthEncoder(spaces, output_size=128)
    sd = {"actor_critic.net.visual_encoder." + k: torch.randn_like(v)
          for k, v in src.visual_encoder.state_dict().items()}
    sd["actor_critic.net.state_encoder.rnn.weight_ih_l0"] = torch.zeros(3)  # ignored prefix
    path = tmp_path / "gibson-2plus-resnet50.pth"
    torch.save({"state_dict": sd, "config": _PickledConfig()}, path)
    dst = enc.VlnResnetDepthEncoder(spaces, output_size=128, checkpoint=str(path))
    k0 = "backbone.conv1.0.weight"
    assert torch.equal(dst.visual_encoder.state_dict()[k0],
                       sd["actor_critic.net.visual_encoder." + k0])


class _FakeTensor:
    def __init__(self, idx):
        self.is_cuda = True
        self.device = types.SimpleNamespace(index=idx)

    def data_ptr(self):
        return 0x1000


def test_launch_device_follows_the_tensors(monkeypatch):
    """policy on cuda:1 while the process's current device is 0 (TORCH_GPU_ID != 0 and no
    torch.cuda.set_device, as the reference trainers run): the stream must be cuda:1's and the
    current device must be 1 for the launch and 0 again afterwards."""
    state = {"cur": 0, "log": []}
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["cur"])
    monkeypatch.setattr(torch.cuda, "set_device",
                        lambda d: (state["log"].append(d), state.__setitem__("cur", d)))
    monkeypatch.setattr(torch.cuda, "current_stream",
                        lambda dev=None: types.SimpleNamespace(
                            cuda_stream=1000 + (state["cur"] if dev is None else dev)))
    _lib._ptr(_FakeTensor(1))
    _lib._ptr(None)
    _lib._ptr(_FakeTensor(1))
    assert _lib._stream() == 1001 and state["cur"] == 1
    _lib._leave_device()
    assert state["cur"] == 0 and state["log"] == [1, 0]
    # same device as current: nothing switches
    _lib._ptr(_FakeTensor(0))
    assert _lib._stream() == 1000 and state["log"] == [1, 0]
    _lib._leave_device()
    # tensors of two devices in one call are refused
    _lib._ptr(_FakeTensor(0))
    with pytest.raises(RuntimeError, match="cuda:0 and cuda:1"):
        _lib._ptr(_FakeTensor(1))
    assert _lib._CALL.dev is None


def test_bucket_rows():
    assert [streams.bucket_rows(n) for n in (1, 8, 9, 80, 81, 200)] == [8, 8, 16, 80, 88, 200]


def test_graphed_tail_cache_is_lru():
    class Tail(streams.GraphedTail):
        MAX_GRAPHS = 2

    t = Tail(lambda: (lambda *a: a[0]))
    t.entries["a"] = 1
    t.entries["b"] = 1
    t.entries.move_to_end("a")          # "a" was used last -> "b" is the eviction victim
    while len(t.entries) >= t.MAX_GRAPHS:
        t.entries.popitem(last=False)
    assert list(t.entries) == ["a"]


def test_categorical_net_falls_back_past_16_actions(monkeypatch):
    """ the one-launch action head covers up to 16 classes; a larger discrete action
    space takes linear + Categorical instead of raising."""
    import hostsim
    from vlnce_amd import _lib
    from vlnce_amd.policy import CategoricalNet

    monkeypatch.setattr(_lib, "_LIB", hostsim.HostSim())
    net = CategoricalNet(24, 20)
    x = torch.randn(5, 24)
    d = net(x)
    ref = torch.distributions.Categorical(logits=torch.nn.functional.linear(x, net.linear.weight, net.linear.bias))
    assert torch.allclose(d.logits, ref.logits, atol=1e-5)


def test_malformed_tuning_variable_is_ignored_with_a_warning(monkeypatch):
    """ VLNCE_U3=off must not break loading the library."""
    import warnings

    from vlnce_amd import _lib

    calls = []
    fake = type("L", (), {"OPTION_NAMES": _lib.HipLib.OPTION_NAMES,
                          "set_option": lambda self, n, v: calls.append((n, v))})()
    monkeypatch.setenv("VLNCE_U3", "off")
    monkeypatch.setenv("VLNCE_P3", "3")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _lib.HipLib._options_from_env(fake)
    assert ("p3", 3) in calls and not any(n == "u3" for n, _ in calls)
    assert any("VLNCE_U3" in str(x.message) for x in w)


def test_batchnorm_sums_live_on_the_module_and_are_cleared_on_failure(monkeypatch):
    """ the persistent BatchNorm column sums are an attribute of the layer (not a global
    keyed by id()), and a failure between the convolution and the finalize leaves them zero."""
    import hostsim
    from vlnce_amd import _lib, ops

    monkeypatch.setattr(_lib, "_LIB", hostsim.HostSim())
    bn = torch.nn.BatchNorm2d(8)
    acc = ops._bn_state(bn)
    assert bn.__dict__["_vlnce_acc"] is acc and "_vlnce_acc" not in bn.state_dict()
    assert ops._bn_state(bn) is acc
    acc += 1.0

    def boom(*a, **k):
        raise RuntimeError("boom")

    monkeypatch.setattr(ops, "bn_finalize_sums", boom)
    monkeypatch.setattr(ops, "conv2d_bn_sums", lambda *a, **k: torch.zeros(1, 2, 2, 8))
    with pytest.raises(RuntimeError, match="boom"):
        ops.conv2d_bn_train(torch.zeros(1, 2, 2, 4), torch.zeros(8, 1, 1, 4), 1, 0, bn)
    assert float(acc.abs().sum()) == 0.0


def test_cached_parameter_list_notices_rebound_parameters():
    """ DropsGraphsOnApply._plist() is what the trunks' graph keys read the parameter
    versions from; a Parameter rebound behind the module's back (assignment, load_state_dict(
    assign=True)) must not leave it -- and the captured graphs -- on the old tensors."""
    import torch.nn as nn

    from vlnce_amd.streams import DropsGraphsOnApply

    class Trunk(DropsGraphsOnApply, nn.Sequential):
        pass

    class Holder:
        def __init__(self):
            self.entries = {"k": "graph"}

    t = Trunk(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    object.__setattr__(t, "_graphs", Holder())
    pl = t._plist()
    assert [id(p) for p in pl] == [id(p) for p in t.parameters()] and t._plist() is pl
    t[0].weight = nn.Parameter(torch.zeros(4, 3, 1, 1))
    pl2 = t._plist()
    assert pl2 is not pl and pl2[0] is t[0].weight and not t._graphs.entries
    t._graphs.entries["k"] = "graph"
    t.load_state_dict({k: v.clone() for k, v in t.state_dict().items()}, assign=True)
    pl3 = t._plist()
    assert [id(p) for p in pl3] == [id(p) for p in t.parameters()] and not t._graphs.entries
    t._graphs.entries["k"] = "graph"
    assert t._plist() is pl3 and t._graphs.entries      # nothing changed: nothing dropped


def test_option_scope_is_per_thread():
    """ `with lib.options(...)` must not change the dispatch of launches made from other
    threads (autograd workers, side-stream helpers): the scope lives in a threading.local."""
    import threading

    import hostsim
    from vlnce_amd import _lib

    lib = _lib.HipLib.__new__(_lib.HipLib)     # the scope logic only: no shared library needed
    lib._tls = threading.local()
    lib._conv_math = 2
    lib.get_option = lambda name: 0
    lib.set_option = lambda name, v: None
    seen = {}
    with _lib.HipLib.options(lib, conv_math=1, u3=2):
        assert lib._tls.scoped == {"conv_math": 1, "u3": 2} and lib.plane_format() == 1
        th = threading.Thread(target=lambda: seen.update(scoped=getattr(lib._tls, "scoped", None),
                                                         fmt=lib.plane_format()))
        th.start()
        th.join()
    assert seen == {"scoped": None, "fmt": 2}
    assert getattr(lib._tls, "scoped", None) is None and lib.plane_format() == 2
    del hostsim
