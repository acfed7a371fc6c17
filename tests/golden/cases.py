"""Golden-vector case table shared by make_goldens.py (runs the REAL reference,
CPU container only) and the parity tests (oracle on CPU, HIP path on GPU).

A case = policy + MODEL overrides + frame size + batch geometry + mode.
Weights are not stored: they are regenerated from state_dict key names by
oracle.thirdparty.synth_state_dict (seeded, construction-order independent).
Inputs ARE stored in the .npz next to the expected outputs.
"""
import os

import numpy as np
import torch

# a synthetic stand-in for data/datasets/R2R_VLNCE_v1-3_preprocessed/embeddings.json.gz (320 words x
# 50 dims, PAD row 0 = zeros, UNK row 1 = the mean row), written by make_goldens_embeddings.py
EMBEDDINGS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "embeddings_synth.json.gz")
EMBEDDINGS_VOCAB = 320

CASES = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "seq2seq_act_128": dict(
        policy="Seq2SeqPolicy", hw=128, N=2, T=1, lengths=[40, 33], mode="eval", call="act",
    ),
    "seq2seq_update_64": dict(
        policy="Seq2SeqPolicy", hw=64, N=2, T=3, lengths=[9, 4], mode="train", call="update",
        overrides={"SEQ2SEQ.use_prev_action": True, "PROGRESS_MONITOR.use": True},
    ),
    "seq2seq_lstm_state_64": dict(
        policy="Seq2SeqPolicy", hw=64, N=3, T=2, lengths=[5, 11, 1], mode="eval", call="dist",
        overrides={"STATE_ENCODER.rnn_type": "LSTM", "INSTRUCTION_ENCODER.rnn_type": "GRU"},
    ),
    "cma_act_64": dict(
        policy="CMAPolicy", hw=64, N=3, T=1, lengths=[7, 5, 12], mode="eval", call="act",
    ),
    "cma_act_train_bn_64": dict(
        policy="CMAPolicy", hw=64, N=4, T=1, lengths=[7, 5, 12, 3], mode="train", call="act",
    ),
    "cma_update_64": dict(
        policy="CMAPolicy", hw=64, N=2, T=3, lengths=[6, 10], mode="train", call="update",
        overrides={"PROGRESS_MONITOR.use": True, "PROGRESS_MONITOR.alpha": 0.5},
    ),
    "cma_cached_feats": dict(
        policy="CMAPolicy", hw=256, N=2, T=4, lengths=[8, 3], mode="train", call="update",
        cached=True,
    ),
    "cma_norm_ablate_64": dict(
        policy="CMAPolicy", hw=64, N=2, T=1, lengths=[4, 9], mode="eval", call="dist",
        overrides={"normalize_rgb": True, "ablate_depth": True},
    ),
    "waypoint_64": dict(
        policy="WaypointPolicy", hw=64, N=2, T=1, lengths=[9, 14], mode="eval", call="waypoint",
    ),
    "waypoint_discrete_64": dict(
        policy="WaypointPolicy", hw=64, N=2, T=1, lengths=[3, 6], mode="eval", call="waypoint",
        overrides={"WAYPOINT.continuous_distance": False, "WAYPOINT.continuous_offset": False},
    ),
    # the reference's real sensor geometry: RGB 224x224, depth 256x256
    # (habitat_extensions/config/vlnce_task.yaml:12-19): 7x7 trunk map -> adaptive pool to 4x4
    "cma_act_rgb224_depth256": dict(
        policy="CMAPolicy", hw=256, rgb_hw=224, N=2, T=1, lengths=[9, 6], mode="train", call="act",
    ),
    # RxR: precomputed 768-d multilingual-BERT token features instead of token ids
    # (rxr_baselines/rxr_cma_en.yaml:45-48), zero rows past each instruction's length
    "cma_rxr_features_64": dict(
        policy="CMAPolicy", hw=64, N=3, T=2, lengths=[11, 5, 8], mode="train", call="update",
        overrides={"INSTRUCTION_ENCODER.sensor_uuid": "rxr_instruction",
                   "INSTRUCTION_ENCODER.embedding_size": 768},
        rxr=True,
    ),
    # H2: one WDDPPO minibatch update (ddppo_alg.py:38-149) on a 3-step x 2-env rollout,
    # encoders in eval mode as ddppo_waypoint_trainer.py:526-530 sets them
    "waypoint_ppo_update_64": dict(
        policy="WaypointPolicy", hw=64, N=2, T=3, lengths=[9, 14], mode="ppo", call="ppo_update",
    ),
    # the heading-prediction-network configs (r2r_waypoint/5-hpn-_c.yaml: no distance head, offset
    # predicted; 6-hpn-__.yaml: neither -- waypoint_policy.py:93-134 then fixes the distance / offset)
    "waypoint_hpn_c_64": dict(
        policy="WaypointPolicy", hw=64, N=2, T=1, lengths=[8, 11], mode="eval", call="waypoint",
        overrides={"WAYPOINT.predict_distance": False},
    ),
    "waypoint_hpn_64": dict(
        policy="WaypointPolicy", hw=64, N=2, T=1, lengths=[5, 13], mode="eval", call="waypoint",
        overrides={"WAYPOINT.predict_distance": False, "WAYPOINT.predict_offset": False},
    ),
    # The upstream DEFAULT instruction embedding (config/default.py:225-232,
    # instruction_encoder.py:36-41,52-61): the table is read from embeddings.json.gz and FROZEN
    # (no embedding gradient, no state to optimise); the second case fine-tunes it.  The table keeps
    # the file's values (synth_state_dict's embedding is not loaded over it).
    "cma_pretrained_embeddings_64": dict(
        policy="CMAPolicy", hw=64, N=3, T=2, lengths=[7, 12, 4], mode="train", call="update",
        vocab=EMBEDDINGS_VOCAB, pretrained_embeddings=True,
        overrides={"INSTRUCTION_ENCODER.use_pretrained_embeddings": True,
                   "INSTRUCTION_ENCODER.embedding_file": EMBEDDINGS_FILE,
                   "INSTRUCTION_ENCODER.fine_tune_embeddings": False},
    ),
    "seq2seq_finetuned_embeddings_64": dict(
        policy="Seq2SeqPolicy", hw=64, N=2, T=2, lengths=[9, 5], mode="train", call="update",
        vocab=EMBEDDINGS_VOCAB, pretrained_embeddings=True,
        overrides={"INSTRUCTION_ENCODER.use_pretrained_embeddings": True,
                   "INSTRUCTION_ENCODER.embedding_file": EMBEDDINGS_FILE,
                   "INSTRUCTION_ENCODER.fine_tune_embeddings": True},
    ),
    # BASELINE.json configs[2] / the bench workload ITSELF: one `_update_agent`
    # (base_il_trainer.py:134-180) of the CMA policy at num_envs = 64, 256x256 RGB-D, instructions of
    # up to 80 tokens, batch-statistics BatchNorm as constructed -- the batch at which the library
    # picks the conv_p3 / conv_u3 / conv_s3 tile plans the bench times.  OUTPUTS ONLY (loss, the
    # [64, 4] logits, every gradient norm, three full gradients, BatchNorm running statistics and
    # their checksums over all 106 layers): the 67 MB of inputs regenerate from the seed
    # (build_inputs); the reference takes ~7 s of CPU for it (8 threads).
    "cma_update_n64_256": dict(
        policy="CMAPolicy", hw=256, N=64, T=1, lengths=[80 - (i % 6) for i in range(64)],
        mode="train", call="update", outputs_only=True, capture_logits=True,
    ),
    # BASELINE.json configs[1] at full size: Seq2Seq `_update_agent`, num_envs = 32, 256x256 RGB-D,
    # <= 80 tokens (outputs only, as above)
    "seq2seq_update_n32_256": dict(
        policy="Seq2SeqPolicy", hw=256, N=32, T=1, lengths=[80 - (i % 6) for i in range(32)],
        mode="train", call="update", outputs_only=True, capture_logits=True,
    ),
    # BASELINE.json configs[4] at full size: one WDDPPO minibatch update (ddppo_alg.py:38-149) of
    # the WaypointPolicy at num_envs = 32 -- 12 panorama frames + the history frame = 416 frames of
    # 256x256 RGB-D through ResNet-18 / the GroupNorm ResNet-50, RxR-length instructions of up to 200
    # tokens, encoders in eval mode (ddppo_waypoint_trainer.py:526-530).  Outputs only + the action
    # components the reference's own act() chose (a few hundred bytes); the 440 MB of frames
    # regenerate from the seed.  The batch at which the library picks the 416-frame tile plans.
    "waypoint_update_n32_256": dict(
        policy="WaypointPolicy", hw=256, N=32, T=1, lengths=[200 - 7 * (i % 9) for i in range(32)],
        mode="ppo", call="ppo_update", outputs_only=True,
    ),
}

VOCAB = 2504

# RL.PPO defaults of the reference (vlnce_baselines/config/default.py:180-201)
PPO = dict(clip_param=0.2, value_loss_coef=0.5, entropy_coef=0.01, pano_entropy_coef=1.0,
           offset_entropy_coef=0.0, distance_entropy_coef=0.0, offset_regularize_coef=0.1146,
           use_clipped_value_loss=True)


def build_inputs(case):
    """Seeded synthetic inputs (BASELINE.md section 3 recipe)."""
    c = case
    g = torch.Generator().manual_seed(1)
    N, T, hw = c["N"], c["T"], c["hw"]
    B = N * T
    pano = c["policy"] == "WaypointPolicy"
    obs = {}
    tok = torch.zeros(N, 200, dtype=torch.long)
    for i, L in enumerate(c["lengths"]):
        tok[i, :L] = torch.randint(1, c.get("vocab", VOCAB), (L,), generator=g)
    obs["instruction"] = tok.repeat(T, 1)  # time-major rows: row = t*N + n
    if c.get("rxr"):
        feats = torch.zeros(N, 32, 768)
        for i, L in enumerate(c["lengths"]):
            feats[i, :L] = torch.randn(L, 768, generator=g) * 0.5
        obs["rxr_instruction"] = feats.repeat(T, 1, 1)
        del obs["instruction"]
    if c.get("cached"):
        # hooks at dagger_trainer.py:300-314 cache the CNN trunk outputs
        obs["rgb_features"] = torch.rand(B, 2048, 4, 4, generator=g) * 2.0
        obs["depth_features"] = torch.rand(B, 128, 4, 4, generator=g)
    elif pano:
        obs["rgb"] = torch.randint(0, 256, (B, 12, hw, hw, 3), generator=g).float()
        obs["depth"] = torch.rand(B, 12, hw, hw, 1, generator=g)
        obs["rgb_history"] = torch.randint(0, 256, (B, hw, hw, 3), generator=g).float()
        obs["depth_history"] = torch.rand(B, hw, hw, 1, generator=g)
        obs["angle_features"] = torch.randn(B, 12, 4, generator=g)
    else:
        rhw = c.get("rgb_hw", hw)
        obs["rgb"] = torch.randint(0, 256, (B, rhw, rhw, 3), generator=g).float()
        obs["depth"] = torch.rand(B, hw, hw, 1, generator=g)
    obs["progress"] = torch.rand(B, 1, generator=g)
    masks = torch.ones(T, N, 1, dtype=torch.uint8)
    masks[0] = 0  # collate_fn: not_done_masks[0] = 0 (dagger_trainer.py:90-111)
    if T > 2:
        masks[2, 0] = 0  # an episode boundary mid-sequence
    if T == 1 and N > 1:
        masks[0, 1:] = 1  # act(): mix of fresh and running episodes
    masks = masks.view(B, 1)
    extra = {}
    if pano:
        prev = {
            "pano": torch.randint(0, 12, (B, 1), generator=g),
            "offset": (torch.rand(B, 1, generator=g) - 0.5) * 0.4,
            "distance": 0.25 + torch.rand(B, 1, generator=g) * 2.0,
        }
        if not c.get("overrides", {}).get("WAYPOINT.continuous_offset", True):
            prev["offset"] = torch.randint(0, 7, (B, 1), generator=g)
            prev["distance"] = torch.randint(0, 6, (B, 1), generator=g)
        extra["hidden"] = 256
    else:
        prev = torch.randint(0, 4, (B, 1), generator=g)
        extra["hidden"] = 512
    extra["targets"] = torch.randint(0, 4, (T, N), generator=g)
    w = torch.rand(T, N, generator=g) + 0.5
    if T > 1:
        w[-1, 0] = 0.0  # padded step of a shorter episode
    extra["weights"] = w
    extra["h0"] = 0.1 * torch.randn(N, 2, extra["hidden"], generator=g)
    if c["call"] == "ppo_update":
        # rollout statistics of the minibatch; the action components themselves are filled
        # in by make_goldens.py from the reference's own act() (they must lie inside the
        # truncated-normal supports) and stored with the inputs
        extra["value_preds"] = torch.randn(B, 1, generator=g) * 0.5
        extra["returns"] = extra["value_preds"] + torch.randn(B, 1, generator=g) * 0.3
        extra["old_logp"] = -2.0 + torch.randn(B, 1, generator=g) * 0.2
        extra["adv"] = torch.randn(B, 1, generator=g)
    return obs, prev, masks, extra


def to_numpy_tree(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(to_numpy_tree(v, prefix + k + "/"))
        elif isinstance(v, torch.Tensor):
            out[prefix + k] = v.detach().cpu().numpy()
        elif v is None:
            continue
        else:
            out[prefix + k] = np.asarray(v)
    return out


def build_policy(ns, case, make_config, make_spaces, synth_state_dict):
    """ns: namespace exposing Seq2SeqPolicy / CMAPolicy / WaypointPolicy."""
    cfg = make_config(case["policy"], **case.get("overrides", {}))
    pano = case["policy"] == "WaypointPolicy"
    obs_space, act_space = make_spaces(case["hw"], case["hw"], pano=pano)
    if "rgb_hw" in case:  # RGB sensor of a different size than depth
        r = case["rgb_hw"]
        rgb_space, _ = make_spaces(r, r, pano=pano)
        obs_space.spaces["rgb"] = rgb_space.spaces["rgb"]
    policy = getattr(ns, case["policy"]).from_config(cfg, obs_space, act_space)
    sd = synth_state_dict(policy)
    if case.get("pretrained_embeddings"):  # the table read from the embeddings file stays
        sd["net.instruction_encoder.embedding_layer.weight"] = \
            policy.net.instruction_encoder.embedding_layer.weight.detach().clone()
    policy.load_state_dict(sd)
    if case["mode"] == "eval":
        policy.eval()
    elif case["mode"] == "ppo":  # agent.train(); visual encoders .eval()
        policy.train()
        policy.net.rgb_encoder.eval()
        policy.net.depth_encoder.eval()
    # mode == "train": leave exactly as constructed (Net.__init__ ends with
    # self.train(): frozen CNN's BatchNorm runs on batch statistics, App. B-1)
    return policy, cfg


_BN_PROBES = [
    "net.rgb_encoder.cnn.1.running_mean",
    "net.rgb_encoder.cnn.1.running_var",
    "net.rgb_encoder.cnn.1.num_batches_tracked",
    "net.rgb_encoder.cnn.7.1.bn2.running_mean",
    "net.rgb_encoder.cnn.7.1.bn2.running_var",
]


def ppo_sample(obs, prev, masks, extra, h0):
    """the 9-tuple RolloutStorage.recurrent_generator yields (one minibatch = all envs)."""
    actions = {k[4:]: v for k, v in extra.items() if k.startswith("act_")}
    return (obs, h0, actions, prev, extra["value_preds"], extra["returns"], masks,
            extra["old_logp"], extra["adv"])


def run_case(policy, case, obs, prev, masks, extra, update_fn=None, aux=None, ppo_fn=None):
    """Runs one case through any implementation of the policy surface and
    returns {name: tensor}.  `update_fn(policy, obs, prev, masks, targets,
    weights) -> (loss, action_loss, aux_loss)` must leave .grad populated and
    NOT step (step_grad=False)."""
    out = {}
    L = policy.net.num_recurrent_layers
    h0 = extra["h0"][:, :L].contiguous()
    call = case["call"]
    dev = h0.device
    if call == "act":
        with torch.no_grad():
            a, h = policy.act(obs, h0, prev, masks, deterministic=True)
            out["actions"], out["rnn_states"] = a, h
            out["logits"] = policy.build_distribution(obs, h0, prev, masks).logits
    elif call == "dist":
        with torch.no_grad():
            out["logits"] = policy.build_distribution(obs, h0, prev, masks).logits
    elif call == "update":
        if aux is not None:
            aux.activate()
        seen = []
        if case.get("capture_logits"):
            # the logits of the very forward the update differentiates (a second forward would
            # move the BatchNorm running statistics again): an instance attribute shadows the
            # method for the duration of the call, whatever implementation `policy` is
            inner = policy.build_distribution

            def recording(*a, **k):
                dist = inner(*a, **k)
                seen.append(dist.logits.detach().clone())
                return dist

            policy.build_distribution = recording
        try:
            loss, al, xl = update_fn(policy, obs, prev, masks, extra["targets"], extra["weights"])
        finally:
            if case.get("capture_logits"):
                del policy.build_distribution
        if seen:
            out["logits"] = seen[0]
        if aux is not None:
            aux.deactivate()
        out["loss"] = torch.tensor([loss, al, xl], dtype=torch.float64)
        names, norms = [], []
        for n, p in policy.named_parameters():
            if p.grad is not None:
                names.append(n)
                norms.append(p.grad.double().norm().item())
        out["grad_names"] = np.array(names)
        out["grad_norms"] = torch.tensor(norms, dtype=torch.float64)
        out["grad_action_w"] = policy.action_distribution.linear.weight.grad
        if hasattr(policy.net, "state_q"):
            out["grad_state_q_w"] = policy.net.state_q.weight.grad
        out["grad_ins_w_hh"] = policy.net.instruction_encoder.encoder_rnn.weight_hh_l0.grad
        if case.get("pretrained_embeddings"):
            emb = policy.net.instruction_encoder.embedding_layer.weight
            out["embedding_requires_grad"] = torch.tensor(int(emb.requires_grad))
            out["embedding_table_checksum"] = emb.detach().double().sum().reshape(1)
            if emb.grad is not None:
                out["grad_embedding"] = emb.grad
    elif call == "waypoint":
        with torch.no_grad():
            pa = {k: v.clone() for k, v in prev.items()}
            (value, _acts, elems, modes, variances, logp, h, pdist) = policy.act(
                obs, h0, pa, masks, deterministic=True
            )
            out.update(value=value, logp=logp, rnn_states=h, pano_logits=pdist.logits)
            for k, v in elems.items():
                out["elem_" + k] = v
            for k, v in modes.items():
                out["mode_" + k] = v
            for k, v in variances.items():
                out["var_" + k] = v
            pa = {k: v.clone() for k, v in prev.items()}
            v2, lp2, ent, h2 = policy.evaluate_actions(obs, h0, pa, masks, elems)
            out.update(ev_value=v2, ev_logp=lp2, ev_rnn_states=h2)
            for k, v in ent.items():
                out["ent_" + k] = v
            pa = {k: v.clone() for k, v in prev.items()}
            out["get_value"] = policy.get_value(obs, h0, pa, masks)
    elif call == "ppo_update":
        # ppo_fn(policy, sample) -> the 6 floats WDDPPO.update returns; leaves .grad populated
        pa = {k: v.clone() for k, v in prev.items()}
        stats = ppo_fn(policy, ppo_sample(obs, pa, masks, extra, h0))
        out["ppo_stats"] = torch.tensor(list(stats), dtype=torch.float64)
        names, norms = [], []
        for n, p in policy.named_parameters():
            if p.grad is not None:
                names.append(n)
                norms.append(p.grad.double().norm().item())
        out["grad_names"] = np.array(names)
        out["grad_norms"] = torch.tensor(norms, dtype=torch.float64)
        out["grad_critic_w"] = policy.critic.fc.weight.grad
        out["grad_ins_w_hh"] = policy.net.instruction_encoder.encoder_rnn.weight_hh_l0.grad
    else:
        raise ValueError(call)
    if case["mode"] == "train" and not case.get("cached"):
        sd = policy.state_dict()
        for k in _BN_PROBES:
            if k in sd:
                out["bn/" + k] = sd[k].detach().clone().float()
        if case.get("outputs_only"):
            # one number per BatchNorm / GroupNorm-free layer: sum of every running_mean and of
            # every running_var of the RGB trunk, in state_dict order
            rm = [sd[k].double().sum() for k in sd if k.endswith("running_mean")]
            rv = [sd[k].double().sum() for k in sd if k.endswith("running_var")]
            out["bn_checksum/running_mean"] = torch.stack(rm)
            out["bn_checksum/running_var"] = torch.stack(rv)
    return {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}


def save_case(path, case_name, obs, prev, masks, extra, outputs):
    blob = {}
    if CASES.get(case_name, {}).get("outputs_only"):
        blob.update(to_numpy_tree(outputs, "out/"))
        # (a PPO case: the action components the reference's act() chose are inputs that do not
        # regenerate from the seed)
        blob.update(to_numpy_tree({k: v for k, v in extra.items() if k.startswith("act_")},
                                  "in/extra/"))
        np.savez_compressed(path, **blob)
        return
    ins = dict(obs=dict(obs), masks=masks, extra=extra)
    ins["prev"] = prev if isinstance(prev, dict) else {"_": prev}
    for k in list(ins["obs"].keys()):
        if k.startswith("rgb") and not k.endswith("features"):
            ins["obs"][k] = ins["obs"][k].to(torch.uint8)  # integer-valued: exact
    blob.update(to_numpy_tree(ins, "in/"))
    blob.update(to_numpy_tree(outputs, "out/"))
    np.savez_compressed(path, **blob)


def load_case(path, device="cpu"):
    """(obs, prev, masks, extra, expected outputs); an outputs-only fixture regenerates its inputs
    from the seed (build_inputs of the case named like the file)."""
    z = np.load(path, allow_pickle=False)
    import os
    name = os.path.splitext(os.path.basename(path))[0]
    if CASES.get(name, {}).get("outputs_only"):
        outs = {k[4:]: (z[k] if z[k].dtype.kind in "US" else torch.from_numpy(z[k]))
                for k in z.files if k.startswith("out/")}
        obs, prev, masks, extra = build_inputs(CASES[name])
        for k in z.files:
            if k.startswith("in/extra/"):
                extra[k[9:]] = torch.from_numpy(z[k])
        mv = lambda t: t.to(device) if isinstance(t, torch.Tensor) else t  # noqa: E731
        return ({k: mv(v) for k, v in obs.items()}, mv(prev), mv(masks),
                {k: mv(v) for k, v in extra.items()}, outs)
    obs, prev, extra, outs = {}, {}, {}, {}
    masks = None
    for k in z.files:
        v = z[k]
        if k.startswith("out/"):
            outs[k[4:]] = v if v.dtype.kind in "US" else torch.from_numpy(v)
            continue
        t = torch.from_numpy(v)
        if k.startswith("in/obs/"):
            name = k[7:]
            if t.dtype == torch.uint8:
                t = t.float()
            obs[name] = t.to(device)
        elif k.startswith("in/prev/"):
            prev[k[8:]] = t.to(device)
        elif k == "in/masks":
            masks = t.to(device)
        elif k.startswith("in/extra/"):
            extra[k[9:]] = t.to(device) if t.dim() > 0 else int(t)
    if list(prev.keys()) == ["_"]:
        prev = prev["_"]
    return obs, prev, masks, extra, outs
