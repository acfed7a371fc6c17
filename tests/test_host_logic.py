"""CPU tier: the package's HOST logic (module wiring, channels-last layouts,
weight repacking, autograd pairing, sequence / packed-sequence handling,
BatchNorm running-stat bookkeeping) run through tests/hostsim.py -- a
tensor-level simulator of the C ABI -- against the golden vectors produced by
the real reference.  The kernels themselves are checked on the GPU tier."""
import os

import pytest
import torch

import cases
import hostsim
import vlnce_amd
from vlnce_amd import _lib
from oracle import thirdparty as tp
from test_oracle_golden import compare

GOLD = os.path.join(os.path.dirname(__file__), "golden")
torch.distributions.Distribution.set_default_validate_args(False)

IL_CASES = [n for n, c in cases.CASES.items() if c["policy"] in vlnce_amd.baseline_registry._policies]


@pytest.fixture()
def sim(monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", hostsim.HostSim())
    yield
    assert _lib._LIB.name == "hostsim"


def product_update(policy, obs, prev, masks, targets, weights):
    from vlnce_amd.il_harness import update_agent

    hs = policy.net.model_config.STATE_ENCODER.hidden_size
    loss, al, xl = update_agent(policy, None, obs, prev, masks, targets, weights, hs,
                                step_grad=False)
    return loss, al, xl


def product_ppo(policy, sample):
    from vlnce_amd.ppo_harness import PPOConfig, wddppo_minibatch_update

    stats = wddppo_minibatch_update(policy, None, sample, PPOConfig(**cases.PPO), step_grad=False,
                                    clip_grads=False)
    return [float(v) for v in stats]


@pytest.mark.parametrize("name", IL_CASES)
def test_host_logic_matches_golden(sim, name):
    case = cases.CASES[name]
    obs, prev, masks, extra, gold = cases.load_case(os.path.join(GOLD, name + ".npz"))
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    outs = cases.run_case(policy, case, obs, prev, masks, extra, product_update,
                          vlnce_amd.AuxLosses, ppo_fn=product_ppo)
    compare(outs, gold, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("fine_tune", [False, True])
def test_pretrained_embeddings_default_path(monkeypatch, fine_tune):
    """The upstream default instruction embedding (use_pretrained_embeddings=True,
    fine_tune_embeddings=False; config/default.py:225-232, instruction_encoder.py:36-61): the table
    comes from the embeddings file under the reference's state_dict key, is frozen unless
    fine-tuned, has no padding_idx -- and a frozen table costs NO backward work: the recurrent
    layer is not asked for the gradient of its input rows and no embedding-backward launch is
    made.  (Values against the reference: the two `*_embeddings_64` goldens.)"""
    calls = []

    class Recording(hostsim.HostSim):
        def rnn_seq_wgrad(self, *a):
            calls.append(("rnn_seq_wgrad", a[13] is not None))   # a[13] = dx_tm
            return super().rnn_seq_wgrad(*a)

        def embedding_bwd(self, *a, **k):
            calls.append(("embedding_bwd", True))
            return super().embedding_bwd(*a, **k)

    monkeypatch.setattr(_lib, "_LIB", Recording())
    cfg = vlnce_amd.make_config("CMAPolicy", **{
        "INSTRUCTION_ENCODER.use_pretrained_embeddings": True,
        "INSTRUCTION_ENCODER.embedding_file": cases.EMBEDDINGS_FILE,
        "INSTRUCTION_ENCODER.fine_tune_embeddings": fine_tune})
    policy = vlnce_amd.build_model(cfg, *vlnce_amd.make_spaces(64, 64))
    emb = policy.net.instruction_encoder.embedding_layer
    assert "net.instruction_encoder.embedding_layer.weight" in policy.state_dict()
    assert tuple(emb.weight.shape) == (cases.EMBEDDINGS_VOCAB, 50) and emb.padding_idx is None
    assert emb.weight.requires_grad == fine_tune
    assert float(emb.weight.detach()[0].abs().sum()) == 0.0 and float(emb.weight.detach()[1].abs().sum()) > 0.0
    case = dict(cases.CASES["cma_pretrained_embeddings_64"])
    obs, prev, masks, extra = cases.build_inputs(case)
    vlnce_amd.AuxLosses.activate()
    product_update(policy, obs, prev, masks, extra["targets"], extra["weights"])
    vlnce_amd.AuxLosses.deactivate()
    assert ("rnn_seq_wgrad", fine_tune) in calls and ("rnn_seq_wgrad", not fine_tune) not in calls
    assert fine_tune or ("embedding_bwd", True) not in calls   # (CPU tensors scatter through torch)
    assert (emb.weight.grad is not None) == fine_tune
    assert policy.net.instruction_encoder.encoder_rnn.weight_ih_l0.grad is not None


def test_no_cpu_fallback_without_library(monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvlnce_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.HipLib(_lib.LIB_PATH)


def test_cpu_tensors_are_rejected_by_the_binding():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    with pytest.raises(RuntimeError, match="not on a GPU"):
        _lib._ptr(torch.zeros(4))


def test_ppo_harness_clips_and_steps(sim):
    """max_grad_norm clipping (habitat PPO.before_step) and the optimizer step of the H2 harness."""
    from vlnce_amd.ppo_harness import PPOConfig, normalized_advantages, wddppo_minibatch_update

    name = "waypoint_ppo_update_64"
    case = cases.CASES[name]
    obs, prev, masks, extra, _ = cases.load_case(os.path.join(GOLD, name + ".npz"))
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    h0 = extra["h0"][:, :policy.net.num_recurrent_layers].contiguous()
    opt = torch.optim.Adam([p for p in policy.parameters() if p.requires_grad], lr=1e-3)
    before = policy.critic.fc.weight.detach().clone()
    wddppo_minibatch_update(policy, opt, cases.ppo_sample(obs, prev, masks, extra, h0),
                            PPOConfig(max_grad_norm=0.2))
    total = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in policy.parameters()
                           if p.grad is not None))
    assert total <= 0.2 * (1 + 1e-4)
    assert not torch.equal(policy.critic.fc.weight.detach(), before)
    r, v = torch.arange(8.0).view(4, 2, 1), torch.ones(4, 2, 1)
    a = normalized_advantages(r, v, normalize=True)
    assert a.shape == (3, 2, 1) and abs(a.mean().item()) < 1e-6


def test_s2d_stem_equals_direct_7x7(sim):
    """space-to-depth formulation of the 7x7/s2/p3 stems (RGB: 3 channels with /255 folded
    in; depth: 1 channel) against the direct convolution, on the ABI simulator."""
    import torch.nn.functional as F
    from vlnce_amd import ops

    for Cc, Cout, hw in ((3, 16, 20), (1, 8, 12)):
        g = torch.Generator().manual_seed(Cc)
        x = torch.rand(2, hw, hw + 4, Cc, generator=g) * 255
        w = torch.randn(Cout, 7, 7, Cc, generator=g) * 0.1
        sc, sh = torch.rand(Cc, generator=g) / 255 + 0.001, torch.randn(Cc, generator=g) * 0.1
        ref = F.conv2d((x * sc + sh).permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, 2, 3)
        y = ops.conv2d_nhwc(ops.space_to_depth2(x, 2, 1, sc, sh), ops.stem_weight_s2d(w), 1, 0)
        assert y.shape == (2, hw // 2, (hw + 4) // 2, Cout)
        assert torch.allclose(y.permute(0, 3, 1, 2), ref, atol=2e-5, rtol=1e-5)


def test_encode_ahead_is_transparent_without_side_streams(sim):
    """On a device without side streams (the ABI simulator runs on CPU) encode_ahead() hands the
    observations back unchanged and the policy output is the plain one."""
    name = "cma_act_64"
    case = cases.CASES[name]
    obs, prev, masks, extra, gold = cases.load_case(os.path.join(GOLD, name + ".npz"))
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    ahead = policy.encode_ahead(obs)
    assert set(ahead) == set(obs) and all(ahead[k] is obs[k] for k in obs)
    outs = cases.run_case(policy, case, ahead, prev, masks, extra, product_update,
                          vlnce_amd.AuxLosses)
    compare(outs, gold, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("H", [16, 64])
@pytest.mark.parametrize("lstm", [False, True])
def test_masked_rnn_rollout_matches_torch_cells(sim, lstm, H):
    """ops.MaskedRNNSeqFn (T-step rollout, one autograd node) against torch GRUCell / LSTMCell
    stepped with the not-done masks (habitat RNNStateEncoder.seq_forward semantics), forward and
    every gradient, with an episode boundary in the middle of the rollout.  H = 64 takes the
    one-launch GRU rollout entry points (vlnce_gru_rollout_fwd / _bwd), H = 16 the per-step ones."""
    from vlnce_amd import ops

    torch.manual_seed(3)
    T, N, D = 5, 3, 7
    cell = (torch.nn.LSTMCell if lstm else torch.nn.GRUCell)(D, H)
    x = torch.randn(T * N, D, requires_grad=True)
    h0 = torch.randn(N, H, requires_grad=True)
    c0 = torch.randn(N, H, requires_grad=True) if lstm else None
    masks = torch.ones(T, N, dtype=torch.uint8)
    masks[0] = 0
    masks[2, 1] = 0
    masks[3, 0] = 0
    wts = torch.randn(T * N, H)
    wh, wc = torch.randn(N, H), torch.randn(N, H)

    # torch reference
    h, c, outs = h0, c0, []
    for t in range(T):
        m = masks[t].float().unsqueeze(1)
        if lstm:
            h, c = cell(x[t * N:(t + 1) * N], (h * m, c * m))
        else:
            h = cell(x[t * N:(t + 1) * N], h * m)
        outs.append(h)
    loss = (torch.cat(outs) * wts).sum() + (h * wh).sum() + ((c * wc).sum() if lstm else 0)
    params = [cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh]
    inputs = [x, h0] + ([c0] if lstm else [])
    ref = torch.autograd.grad(loss, inputs + params)

    gi = ops.linear(x, cell.weight_ih, cell.bias_ih)
    y, hT, cT = ops.MaskedRNNSeqFn.apply(lstm, gi, h0, c0, masks.view(-1), cell.weight_hh,
                                         cell.bias_hh)
    assert torch.allclose(y, torch.cat(outs), atol=1e-5)
    loss2 = (y * wts).sum() + (hT * wh).sum() + ((cT * wc).sum() if lstm else 0)
    got = torch.autograd.grad(loss2, inputs + params)
    for g, r_ in zip(got, ref):
        assert torch.allclose(g, r_, atol=2e-5, rtol=1e-4)


def test_edge_cases_match_reference_behaviour(sim):
    """empty instruction -> the reference's pack_padded_sequence error (message kept);
    a 200-token (maximum length) instruction next to a 1-token one; batch of one."""
    from oracle import policy_cpu as oc

    case = cases.CASES["cma_act_64"]
    ref, _ = cases.build_policy(oc, case, tp.make_config, tp.make_spaces, tp.synth_state_dict)
    hip, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                tp.synth_state_dict)
    obs, prev, masks, extra = cases.build_inputs(case)
    h0 = extra["h0"][:, :hip.net.num_recurrent_layers].contiguous()

    bad = dict(obs)
    bad["instruction"] = obs["instruction"].clone()
    bad["instruction"][1] = 0  # an all-padding instruction
    for policy in (ref, hip):
        with pytest.raises(RuntimeError, match="greater than 0"):
            with torch.no_grad():
                policy.act(bad, h0, prev, masks, deterministic=True)

    g = torch.Generator().manual_seed(9)
    ragged = dict(obs)
    ragged["instruction"] = torch.zeros_like(obs["instruction"])
    ragged["instruction"][0, :200] = torch.randint(1, 2504, (200,), generator=g)  # maximum length
    ragged["instruction"][1, :1] = 7                                               # minimum length
    ragged["instruction"][2, :33] = torch.randint(1, 2504, (33,), generator=g)
    with torch.no_grad():
        a_ref = ref.build_distribution(ragged, h0, prev, masks).logits
        a_hip = hip.build_distribution(ragged, h0, prev, masks).logits
        one = {k: v[:1] for k, v in ragged.items()}
        b_ref = ref.build_distribution(one, h0[:1], prev[:1], masks[:1]).logits
        b_hip = hip.build_distribution(one, h0[:1], prev[:1], masks[:1]).logits
    assert torch.allclose(a_hip, a_ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(b_hip, b_ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("discrete", [False, True])
def test_waypoint_switched_off_components_match_oracle(sim, discrete):
    """WAYPOINT.predict_distance / predict_offset = False (waypoint_policy.py:93-134): the
    constants handed to the simulator and stored in the rollout, zeroed variances, and the
    log-probability / entropy masks, against the oracle restatement."""
    from oracle import policy_cpu as oc

    over = {"WAYPOINT.predict_distance": False, "WAYPOINT.predict_offset": False}
    if discrete:
        over.update({"WAYPOINT.continuous_distance": False, "WAYPOINT.continuous_offset": False})
    case = dict(cases.CASES["waypoint_64"], overrides=over)
    ref, _ = cases.build_policy(oc, case, tp.make_config, tp.make_spaces, tp.synth_state_dict)
    hip, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                tp.synth_state_dict)
    obs, prev, masks, extra = cases.build_inputs(case)
    a = cases.run_case(ref, case, obs, prev, masks, extra)
    b = cases.run_case(hip, case, obs, prev, masks, extra)
    compare(b, a, atol=1e-4, rtol=1e-4)
    assert float(b["elem_distance"].float().abs().max()) in (0.0, 0.25)


def test_truncated_normal_matches_oracle_including_sampling():
    """vlnce_amd.utils.TruncatedNormal (own formulation through the standardised window) against
    the oracle restatement of utils.py:24-152: moments, entropy, log-probability, and the
    rejection sampler draw for draw under the same seed."""
    from oracle import policy_cpu as oc
    from vlnce_amd.utils import TruncatedNormal

    g = torch.Generator().manual_seed(2)
    loc = (torch.rand(64, 1, generator=g) - 0.5) * 0.4
    scale = torch.rand(64, 1, generator=g) * 0.3 + 0.02
    lo, hi = -0.26, 0.26
    a, b = TruncatedNormal(loc, scale, lo, hi), oc.TruncatedNormal(loc, scale, lo, hi)
    for got, want in ((a.mean, b.mean), (a.variance, b.variance), (a.entropy(), b.entropy()),
                      (a.mode(), b.mode())):
        assert torch.allclose(got, want, atol=1e-6, rtol=1e-5)
    v = (torch.rand(64, 1, generator=g) - 0.5) * 0.5
    assert torch.allclose(a.log_prob(v), b.log_prob(v), atol=1e-5, rtol=1e-5)
    torch.manual_seed(11)
    sa = a.sample()
    torch.manual_seed(11)
    sb = b.sample()
    assert torch.equal(sa, sb) and bool(((sa >= lo) & (sa <= hi)).all())
    with pytest.raises(AssertionError):
        a.log_prob(torch.full((64, 1), 0.3))


@pytest.mark.parametrize("policy_name,final_only", [("CMAPolicy", False), ("Seq2SeqPolicy", True)])
def test_instruction_dedup_equals_row_by_row_encoding(sim, policy_name, final_only):
    """A sequence-mode batch repeats each episode's token row T times; encoding the distinct rows
    once and gathering must equal encoding every row (values and parameter gradients)."""
    from vlnce_amd.encoders.instruction_encoder import InstructionEncoder

    torch.manual_seed(3)
    cfg = vlnce_amd.make_config(policy_name).MODEL.INSTRUCTION_ENCODER
    cfg.final_state_only = final_only
    enc = InstructionEncoder(cfg)
    g = torch.Generator().manual_seed(4)
    eps = torch.zeros(3, 200, dtype=torch.long)
    for i, L in enumerate((5, 9, 7)):
        eps[i, :L] = torch.randint(1, 2504, (L,), generator=g)
    pad = torch.ones(1, 200, dtype=torch.long)            # collate_fn pads with 1.0 (App. B-7)
    tokens = torch.cat([eps, eps, eps[:2], pad, eps, pad, pad, eps, eps[1:]], dim=0)  # 20 rows
    assert tokens.size(0) >= 16  # (the default threshold, 128 rows, is lowered for this small batch)
    w = torch.randn(1)

    def run(min_rows):
        enc.DEDUP_MIN_ROWS = min_rows
        enc.zero_grad()
        out = enc({"instruction": tokens})
        (out * torch.linspace(0.5, 1.5, out.numel()).view_as(out) * w).sum().backward()
        return out.detach(), [p.grad.clone() for p in enc.parameters() if p.grad is not None]

    ref, gref = run(10 ** 9)
    got, ggot = run(16)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, atol=1e-6, rtol=1e-6)
    assert len(ggot) == len(gref) > 0
    for a, b in zip(ggot, gref):
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("rows", [4, 20])
def test_instruction_length_counts_embedded_vectors_not_token_ids(sim, rows):
    """instruction_encoder.py:79-80 overwrites the token-id count of :72 with "steps whose EMBEDDED
    vector is not all-zero": a non-pad token whose table row is zero (pretrained tables have
    such rows) does not count.  Both the plain and the de-duplicating path follow the table."""
    from oracle import policy_cpu as oc
    from vlnce_amd.encoders.instruction_encoder import InstructionEncoder

    torch.manual_seed(5)
    cfg = vlnce_amd.make_config("CMAPolicy").MODEL.INSTRUCTION_ENCODER
    cfg.final_state_only = True
    enc = InstructionEncoder(cfg)
    ref = oc.InstructionEncoder(cfg)
    ref.load_state_dict(enc.state_dict())
    with torch.no_grad():
        enc.embedding_layer.weight[7] = 0.0
        ref.embedding_layer.weight[7] = 0.0
    base = torch.zeros(2, 200, dtype=torch.long)
    base[0, :6] = torch.tensor([3, 9, 7, 7, 11, 7])   # zero-row token inside and at the end
    base[1, :4] = torch.tensor([5, 6, 8, 2])
    tokens = base.repeat(rows // 2, 1)
    enc.DEDUP_MIN_ROWS = 16
    got = enc({"instruction": tokens})
    want = ref({"instruction": tokens})
    assert got.shape == want.shape
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)


def test_split_weights_is_gpu_only_and_channel_gated():
    """The bf16-plane split belongs to the HIP convolution kernel: on the CPU (hostsim) path, and
    for weights whose input-channel count the kernel does not take, ops.split_weights hands back
    None and the call goes down the fp32 path."""
    from vlnce_amd import ops

    assert ops.split_weights(torch.randn(8, 3, 3, 32)) is None      # CPU tensor
    assert ops.split_weights(torch.randn(8, 7, 7, 3)) is None       # Cin % 32 != 0


def test_moving_a_policy_drops_its_captured_graphs():
    """Module._apply (.to / .float / .double) gives parameters new storage without touching the
    version counters the graph keys are made of: every graph holder of the policy (trunk graphs,
    tail graphs, the whole-act graph) must come out of it empty."""
    case = cases.CASES["cma_act_64"]
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    from vlnce_amd.streams import ActGraph

    object.__setattr__(policy, "_act_graph", ActGraph(policy))
    holders = [policy._act_graph, policy.net._tail, policy.net.rgb_encoder.cnn._graphs,
               policy.net.depth_encoder.visual_encoder._graphs]
    for h in holders:
        h.entries["stale"] = "seen"
    policy.net._tail.sightings["stale"] = 1
    policy._act_graph._tracked = ([], [])
    versions = [p._version for p in policy.parameters()]
    policy.double()
    assert [p._version for p in policy.parameters()] == versions  # (why the keys cannot see it)
    assert all(len(h.entries) == 0 for h in holders)
    assert len(policy.net._tail.sightings) == 0 and policy._act_graph._tracked is None


def test_categorical_net_matches_torch_categorical_and_raises_like_it(sim, monkeypatch):
    """policy.CategoricalNet builds the distribution from the fused action head (normalised logits,
    NaN count) -- same logits / log_prob / entropy / gradients as torch's Categorical(logits=...)
    and the reference's ValueError when the parameter holds a NaN (utils.py:269-289)."""
    from vlnce_amd.policy import CategoricalNet
    # (this module switches argument validation off for speed; torch's default is on)
    monkeypatch.setattr(torch.distributions.Distribution, "_validate_args", True)
    torch.manual_seed(3)
    net = CategoricalNet(32, 4)
    torch.nn.init.normal_(net.linear.weight, std=0.3)
    x = torch.randn(6, 32, requires_grad=True)
    d = net(x)
    ref = torch.distributions.Categorical(logits=torch.nn.functional.linear(x, net.linear.weight,
                                                                            net.linear.bias))
    assert torch.allclose(d.logits, ref.logits, atol=1e-6)
    assert torch.allclose(d.probs, ref.probs, atol=1e-6)
    a = torch.tensor([[0], [3], [1], [2], [2], [0]])
    assert d.log_probs(a).shape == (6, 1)
    assert torch.allclose(d.log_probs(a).squeeze(-1), ref.log_prob(a.squeeze(-1)), atol=1e-6)
    assert torch.allclose(d.entropy(), ref.entropy(), atol=1e-6)
    assert d.sample().shape == (6, 1) and d.mode().shape == (6, 1)
    gx, gw = torch.autograd.grad(d.logits[:, 1].sum(), (x, net.linear.weight))
    rx, rw = torch.autograd.grad(ref.logits[:, 1].sum(), (x, net.linear.weight))
    assert torch.allclose(gx, rx, atol=1e-6) and torch.allclose(gw, rw, atol=1e-6)
    with pytest.raises(ValueError):                      # sample outside the support: torch's check
        d.log_probs(torch.full((6, 1), 7))
    bad = x.detach().clone()
    bad[4, 0] = float("nan")
    with pytest.raises(ValueError, match="logits"):
        net(bad)
    with pytest.raises(ValueError):                      # the reference construction raises too
        torch.distributions.Categorical(logits=torch.nn.functional.linear(
            bad, net.linear.weight, net.linear.bias))
    assert float(net(x.detach()).logits.exp().sum(-1).mean()) == pytest.approx(1.0, abs=1e-6)
    monkeypatch.setattr(torch.distributions.Distribution, "_validate_args", False)
    assert bool(torch.isnan(net(bad).logits[4]).all())   # validation off: no read-back, no raise


def test_distinct_instruction_path_matches_reference_golden(sim):
    """The cached-feature CMA update with the instruction encoder run once per DISTINCT instruction
    and the text attention reading the distinct K / V blocks in place (ops.attention(index=...),
    vlnce_attn_*_shared + vlnce_segment_sum) against the REAL reference's golden outputs and
    gradients of the same batch -- the threshold is lowered so that this 8-row batch takes it."""
    name = "cma_cached_feats"
    case = cases.CASES[name]
    obs, prev, masks, extra, gold = cases.load_case(os.path.join(GOLD, name + ".npz"))
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    enc = policy.net.instruction_encoder
    enc.DEDUP_MIN_ROWS = 4
    seen = []
    orig = enc.forward

    def spy(observations, distinct=False):
        out = orig(observations, distinct=distinct)
        seen.append(out[1] if distinct else None)
        return out

    enc.forward = spy
    outs = cases.run_case(policy, case, obs, prev, masks, extra, product_update,
                          vlnce_amd.AuxLosses, ppo_fn=product_ppo)
    assert seen and seen[0] is not None and int(seen[0].max()) + 1 < obs["instruction"].size(0)
    compare(outs, gold, atol=1e-4, rtol=1e-4)


def test_linear_dx_from_and_frozen_feature_columns(sim):
    """ops.linear(dx_from=c0) returns the input gradient of the trailing columns only (leading ones
    zero) with the full weight / bias gradients; CMANet._frozen_cols says how many leading channels
    of an encoder's output cannot need a gradient: the trunk's channels when it is frozen or its
    features came in precomputed, none when the trunk trains or an ablation multiplies the rows."""
    from vlnce_amd import ops
    torch.manual_seed(0)
    x = torch.randn(6, 40, requires_grad=True)
    w = torch.randn(12, 40, requires_grad=True)
    b = torch.randn(12, requires_grad=True)
    g = torch.randn(6, 12)
    (ops.linear(x, w, b, dx_from=32) * g).sum().backward()
    got = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    (torch.nn.functional.linear(x, w, b) * g).sum().backward()
    assert torch.equal(got[0][:, :32], torch.zeros(6, 32))
    assert torch.allclose(got[0][:, 32:], x.grad[:, 32:], atol=1e-5)
    assert torch.allclose(got[1], w.grad, atol=1e-5) and torch.allclose(got[2], b.grad, atol=1e-5)
    # dx_from that is not a multiple of 4 (alignment of the column slice) or out of range: ignored
    x.grad = None
    (ops.linear(x, w, b, dx_from=30) * g).sum().backward()
    assert float(x.grad[:, :30].abs().min()) > 0.0

    case = cases.CASES["cma_update_64"]
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    net = policy.net
    rgb_c = net.rgb_encoder.output_shape[0] - 64
    dep_c = net.depth_encoder.output_shape[0] - 64
    assert net._frozen_cols(net.rgb_encoder, "rgb_features", {}) == rgb_c > 0
    assert net._frozen_cols(net.depth_encoder, "depth_features", {}) == dep_c > 0
    for p in net.rgb_encoder.trunk_parameters():
        p.requires_grad_(True)
    assert net._frozen_cols(net.rgb_encoder, "rgb_features", {}) == 0
    feats = torch.zeros(2, rgb_c, 4, 4)
    assert net._frozen_cols(net.rgb_encoder, "rgb_features", {"rgb_features": feats}) == rgb_c
    assert net._frozen_cols(net.rgb_encoder, "rgb_features",
                            {"rgb_features": feats.requires_grad_()}) == 0
    net.model_config.defrost()
    net.model_config.ablate_depth = True
    net.model_config.freeze()
    assert net._frozen_cols(net.depth_encoder, "depth_features", {}) == 0
