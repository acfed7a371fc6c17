"""CPU tier: the data-parallel gradient exchange (vlnce_amd.distributed) with
world_size 2 over gloo: averaged shard gradients == single-process gradient of
the global batch, including a parameter that never receives a gradient."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from vlnce_amd.distributed import GradientAllReducer, shard_rows


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(12, 32)
        self.b = nn.Linear(32, 4)
        self.unused = nn.Linear(3, 3)  # WaypointPolicy.action_distribution analogue
        self.frozen = nn.Linear(5, 5)
        for p in self.frozen.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def il_loss(model, x, tgt, w):
    # per-episode normalised, then mean over episodes (base_il_trainer.py:159-165)
    T, N = tgt.shape
    logits = model(x).view(T, N, -1)
    ce = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    return ((w * ce).sum(0) / w.sum(0)).mean()


def make_data():
    g = torch.Generator().manual_seed(3)
    T, N = 3, 8
    return (torch.randn(T, N, 12, generator=g), torch.randint(0, 4, (T, N), generator=g),
            torch.rand(T, N, generator=g) + 0.5)


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Tiny()
    red = GradientAllReducer(model, bucket_bytes=1024)  # several buckets
    x, tgt, w = make_data()
    sl = shard_rows(x.size(1), rank, world)
    for _ in range(2):  # second pass checks the hooks re-arm
        model.zero_grad()
        il_loss(model, x[:, sl].reshape(-1, 12), tgt[:, sl], w[:, sl]).backward()
        red.finish()
    if rank == 0:
        torch.save({n: p.grad for n, p in model.named_parameters() if p.grad is not None}, out)
    dist.destroy_process_group()


def unused_first_worker(rank, world, port, out):
    """WaypointPolicy's layout: the unused head sits between used modules, so in reverse
    parameter order it lands in an EARLY bucket.  From the second step on it must not hold the
    other buckets back until finish(); a head that wakes up later must still be averaged."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Tiny()
    red = GradientAllReducer(model, bucket_bytes=64)  # about one bucket per tensor
    x, tgt, w = make_data()
    sl = shard_rows(x.size(1), rank, world)
    early = []
    for step in range(3):
        model.zero_grad()
        loss = il_loss(model, x[:, sl].reshape(-1, 12), tgt[:, sl], w[:, sl])
        if step == 2:   # the head is used after all (same on every rank: static mode)
            loss = loss + model.unused(x[0, sl, :3]).pow(2).mean() * (rank + 1)
        loss.backward()
        red.finish()
        early.append(red.launched_before_finish)
    if rank == 0:
        torch.save({"early": early, "n_buckets": len(red.buckets),
                    "grads": {n: p.grad for n, p in model.named_parameters() if p.grad is not None}},
                   out)
    dist.destroy_process_group()


def test_unused_head_in_an_early_bucket_does_not_serialise_the_collectives(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "early.pt")
    mp.spawn(unused_first_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    n = got["n_buckets"]
    assert n >= 4
    # step 0: the unused head's bucket (index 0 or 1 in reverse order) blocks the strict order
    assert got["early"][0] < n - 1
    # step 1: learned -- every bucket goes out from the hooks, none is left for finish()
    assert got["early"][1] == n, got["early"]
    # step 2: the head received a gradient after its bucket went out: reduced in finish()
    torch.manual_seed(0)
    model = Tiny()
    x, tgt, w = make_data()
    # reference: mean over the two ranks' losses = IL loss of the global batch + the rank-weighted extra
    il_loss(model, x.reshape(-1, 12), tgt, w).backward()
    extra = 0.0
    for r in range(2):
        sl = shard_rows(x.size(1), r, 2)
        extra = extra + model.unused(x[0, sl, :3]).pow(2).mean() * (r + 1) / 2
    extra.backward()
    for name, p in model.named_parameters():
        if p.grad is not None:
            assert torch.allclose(got["grads"][name], p.grad, atol=1e-6), name
    assert "unused.weight" in got["grads"]


def test_two_rank_gradients_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "grads.pt")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = Tiny()
    x, tgt, w = make_data()
    il_loss(model, x.reshape(-1, 12), tgt, w).backward()
    want = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(want) and "unused.weight" not in got
    for n in want:
        assert torch.allclose(got[n], want[n], atol=1e-6), n


# ------------------------------------------------------------------ the real policy, sharded
def _policy_and_batch():
    import cases
    import vlnce_amd
    from oracle import thirdparty as tp

    case = dict(cases.CASES["cma_update_64"], N=4, lengths=[6, 10, 3, 8], mode="eval")
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    policy.train()
    policy.net.rgb_encoder.eval()   # frozen trunks on running statistics: shard-independent
    policy.net.depth_encoder.eval()
    obs, prev, masks, extra = cases.build_inputs(case)
    return policy, obs, prev, masks, extra, case


def _shard(t, T, N, sl):
    """time-major rows t*N + n -> the rows of envs `sl`."""
    return t.view(T, N, *t.shape[1:])[:, sl].reshape(-1, *t.shape[1:])


def policy_worker(rank, world, port, out):
    import hostsim
    from vlnce_amd import _lib
    from vlnce_amd.il_harness import update_agent

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib._LIB = hostsim.HostSim()
    policy, obs, prev, masks, extra, case = _policy_and_batch()
    T, N = case["T"], case["N"]
    sl = shard_rows(N, rank, world)
    red = GradientAllReducer(policy, bucket_bytes=1 << 18)
    update_agent(policy, None, {k: _shard(v, T, N, sl) for k, v in obs.items()},
                 _shard(prev, T, N, sl), _shard(masks, T, N, sl), extra["targets"][:, sl],
                 extra["weights"][:, sl], 512, step_grad=False, grad_hook=red.finish)
    if rank == 0:
        torch.save({n: p.grad for n, p in policy.named_parameters() if p.grad is not None}, out)
    dist.destroy_process_group()


def test_sharded_policy_update_equals_full_batch(tmp_path, monkeypatch):
    """CMA DAgger update of 4 episodes on 2 gloo ranks (2 episodes each, the ABI simulator under
    the HIP host path) == the single-process update of all 4: the IL loss is normalised per
    episode, so equal shards + gradient averaging are exact (base_il_trainer.py:159-165)."""
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hostsim
    from vlnce_amd import _lib
    from vlnce_amd.il_harness import update_agent

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "policy_grads.pt")
    mp.spawn(policy_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    monkeypatch.setattr(_lib, "_LIB", hostsim.HostSim())
    policy, obs, prev, masks, extra, case = _policy_and_batch()
    update_agent(policy, None, obs, prev, masks, extra["targets"], extra["weights"], 512,
                 step_grad=False)
    want = {n: p.grad for n, p in policy.named_parameters() if p.grad is not None}
    assert set(got) == set(want) and len(want) > 20
    gmax = max(v.abs().max().item() for v in want.values())
    for n in want:
        assert torch.allclose(got[n], want[n], rtol=1e-4, atol=1e-6 * max(gmax, 1.0)), n


# ------------------------------------------------------------------ progress monitor under DP
def _pm_policy_and_batch():
    import cases
    import vlnce_amd
    from oracle import thirdparty as tp

    case = dict(cases.CASES["cma_update_64"], N=4, T=3, lengths=[6, 10, 3, 8], mode="eval",
                overrides={"PROGRESS_MONITOR.use": True})
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    policy.train()
    policy.net.rgb_encoder.eval()
    policy.net.depth_encoder.eval()
    obs, prev, masks, extra = cases.build_inputs(case)
    w = extra["weights"].clone()
    w[1:, 0] = 0.0  # unbalanced: rank 0's shard has fewer unmasked rows than rank 1's
    w[2, 1] = 0.0
    extra["weights"] = w
    return policy, obs, prev, masks, extra, case


def pm_worker(rank, world, port, out, exact):
    import hostsim
    import vlnce_amd
    from vlnce_amd import _lib
    from vlnce_amd.il_harness import update_agent

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib._LIB = hostsim.HostSim()
    policy, obs, prev, masks, extra, case = _pm_policy_and_batch()
    T, N = case["T"], case["N"]
    sl = shard_rows(N, rank, world)
    red = GradientAllReducer(policy, bucket_bytes=1 << 18)
    vlnce_amd.AuxLosses.activate()
    vlnce_amd.AuxLosses.set_data_parallel(exact)
    update_agent(policy, None, {k: _shard(v, T, N, sl) for k, v in obs.items()},
                 _shard(prev, T, N, sl), _shard(masks, T, N, sl), extra["targets"][:, sl],
                 extra["weights"][:, sl], 512, step_grad=False, grad_hook=red.finish)
    if rank == 0:
        torch.save({n: p.grad for n, p in policy.named_parameters() if p.grad is not None}, out)
    dist.destroy_process_group()


def test_progress_monitor_loss_is_a_global_masked_mean_under_data_parallelism(tmp_path, monkeypatch):
    """aux_losses.py:24-32 averages the progress loss over ALL unmasked rows (and, through the
    [B] x [B,1] broadcast of App. B-2, over the progress targets of the whole batch).  With
    unbalanced masks the mean of per-rank means differs; AuxLosses.set_data_parallel() restores
    the single-process gradient exactly."""
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hostsim
    import vlnce_amd
    from vlnce_amd import _lib
    from vlnce_amd.il_harness import update_agent

    def spawn(exact):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = str(tmp_path / f"pm_{int(exact)}.pt")
        mp.spawn(pm_worker, args=(2, port, out, exact), nprocs=2, join=True)
        return torch.load(out)

    monkeypatch.setattr(_lib, "_LIB", hostsim.HostSim())
    policy, obs, prev, masks, extra, case = _pm_policy_and_batch()
    vlnce_amd.AuxLosses.activate()
    update_agent(policy, None, obs, prev, masks, extra["targets"], extra["weights"], 512,
                 step_grad=False)
    vlnce_amd.AuxLosses.deactivate()
    want = {n: p.grad for n, p in policy.named_parameters() if p.grad is not None}
    key = "net.progress_monitor.weight"
    assert key in want
    gmax = max(v.abs().max().item() for v in want.values())

    def worst(got):
        return max((got[n] - want[n]).abs().max().item() for n in want)

    exact, naive = spawn(True), spawn(False)
    assert worst(exact) < 1e-5 * max(gmax, 1.0), worst(exact)
    # the documented deviation of the plain per-rank mean is real on this batch
    assert worst(naive) > 1e-3 * want[key].abs().max().item()


# ------------------------------------------------------------------ torch DDP + WaypointPolicy
def _wp_policy_and_sample():
    import cases
    import vlnce_amd
    from oracle import thirdparty as tp

    # equal instruction lengths: the waypoint net's instruction attention keeps PAD keys in its
    # softmax (App. B-3), so its output depends on the batch's longest instruction (App. B-8) --
    # upstream's DDP shards differ from the unsharded batch in exactly the same way
    case = dict(cases.CASES["waypoint_ppo_update_64"], N=4, T=2, lengths=[9, 9, 9, 9])
    policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config, vlnce_amd.make_spaces,
                                   tp.synth_state_dict)
    obs, prev, masks, extra = cases.build_inputs(case)
    with torch.no_grad():
        B = obs["rgb"].size(0)
        out = policy.act(obs, torch.zeros(B, policy.net.num_recurrent_layers, 256),
                         {k: v.clone() for k, v in prev.items()}, masks, deterministic=True)
    for k, v in out[2].items():
        extra["act_" + k] = v.detach().clone()
    return policy, obs, prev, masks, extra, case


def _wp_loss(policy, obs, prev, masks, extra, h0):
    """the actor-critic terms of WDDPPO.update (ddppo_alg.py:65-132) -- plain means over rows"""
    import cases

    actions = {k[4:]: v for k, v in extra.items() if k.startswith("act_")}
    values, logp, entropy, _ = policy.evaluate_actions(obs, h0, prev, masks, actions)
    ratio = torch.exp(logp - extra["old_logp"])
    adv = extra["adv"]
    action_loss = -torch.min(ratio * adv, ratio.clamp(0.8, 1.2) * adv).mean()
    value_loss = 0.5 * ((extra["returns"] - values) ** 2).mean()
    return value_loss * cases.PPO["value_loss_coef"] + action_loss \
        - cases.PPO["entropy_coef"] * entropy["pano"].mean()


def ddp_worker(rank, world, port, out):
    import hostsim
    from torch.nn.parallel import DistributedDataParallel as DDP
    from vlnce_amd import _lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib._LIB = hostsim.HostSim()
    policy, obs, prev, masks, extra, case = _wp_policy_and_sample()
    T, N = case["T"], case["N"]
    sl = shard_rows(N, rank, world)
    sh = lambda t: _shard(t, T, N, sl)  # noqa: E731
    # habitat-lab v0.1.7 DecentralizedDistributedMixin.init_distributed: the policy is wrapped
    # only to obtain DDP's reducer; forward goes through evaluate_actions on the bare module and
    # reducer.prepare_for_backward([loss]) marks the unused action_distribution.* parameters
    # (ddppo_waypoint_trainer.py:370: find_unused_params=True)
    ddp = DDP(policy, find_unused_parameters=True)
    for _ in range(2):
        policy.zero_grad()
        ex = {k: (sh(v) if isinstance(v, torch.Tensor) and v.dim() > 0 and v.size(0) == T * N else v)
              for k, v in extra.items()}
        loss = _wp_loss(policy, {k: sh(v) for k, v in obs.items()}, {k: sh(v) for k, v in prev.items()},
                        sh(masks), ex, extra["h0"][sl])
        ddp.reducer.prepare_for_backward([loss])
        loss.backward()
    if rank == 0:
        torch.save({n: p.grad for n, p in policy.named_parameters() if p.grad is not None}, out)
    dist.destroy_process_group()


def test_waypoint_policy_under_torch_ddp_reducer(tmp_path, monkeypatch):
    """The reference's only distributed path: WaypointPolicy inside torch DDP with
    find_unused_parameters=True.  The custom autograd Functions of the HIP host path (here on the
    ABI simulator) must coexist with DDP's reducer hooks: 2 gloo ranks x 2 envs == 4 envs."""
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hostsim
    from vlnce_amd import _lib

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "ddp_grads.pt")
    mp.spawn(ddp_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    monkeypatch.setattr(_lib, "_LIB", hostsim.HostSim())
    policy, obs, prev, masks, extra, case = _wp_policy_and_sample()
    _wp_loss(policy, obs, prev, masks, extra, extra["h0"]).backward()
    want = {n: p.grad for n, p in policy.named_parameters() if p.grad is not None}
    assert "action_distribution.linear.weight" not in want  # the unused head
    assert set(want) <= set(got) and len(want) > 20
    gmax = max(v.abs().max().item() for v in want.values())
    for n in want:
        assert torch.allclose(got[n], want[n], rtol=1e-4, atol=1e-6 * max(gmax, 1.0)), n


class Branchy(nn.Module):
    """A head that only SOME ranks use in a given step (a data-dependent branch)."""

    def __init__(self):
        super().__init__()
        self.trunk = nn.Linear(12, 64)
        self.sometimes = nn.Linear(64, 64)   # early in parameter order, late or never in backward
        self.head = nn.Linear(64, 4)

    def forward(self, x, use_branch):
        h = torch.relu(self.trunk(x))
        if use_branch:
            h = h + torch.tanh(self.sometimes(h))
        return self.head(h)


def divergent_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Branchy()
    red = GradientAllReducer(model, bucket_bytes=512, divergent_unused=True)  # head | sometimes | trunk
    assert len(red.buckets) >= 3
    x, tgt, w = make_data()
    sl = shard_rows(x.size(1), rank, world)
    issued = []
    launch = red._launch
    red._launch = lambda b: (issued.append(red.buckets.index(b)), launch(b))[1]
    for step in range(3):
        model.zero_grad()
        # the ranks disagree on which parameters get a gradient: rank 0 takes the branch on even
        # steps, rank 1 on odd steps
        use = (step + rank) % 2 == 0
        logits = model(x[:, sl].reshape(-1, 12), use).view(3, -1, 4)
        ce = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), tgt[:, sl], reduction="none")
        ((w[:, sl] * ce).sum(0) / w[:, sl].sum(0)).mean().backward()
        red.finish()
    # every step issued the buckets in index order on this rank (hence on every rank)
    n = len(red.buckets)
    assert issued == list(range(n)) * 3, issued
    if rank == 0:
        torch.save({k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}, out)
    dist.destroy_process_group()


def test_reducer_issues_buckets_in_order_when_ranks_disagree_on_unused_parameters(tmp_path):
    """ROUND-2 review: a bucket whose last gradient never arrives used to be launched in finish(),
    i.e. AFTER later buckets; with rank-divergent unused parameters the ranks then issue their
    collectives in different orders and hang.  Buckets are now launched strictly in index order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "g.pt")
    mp.spawn(divergent_worker, args=(2, port, out), nprocs=2, join=True)
    g = torch.load(out)
    assert g["head.weight"] is not None and torch.isfinite(g["head.weight"]).all()
    # last step (2): rank 0 took the branch, rank 1 did not -- and on step 1 the other way round;
    # either way the parameter ends up with the averaged gradient on BOTH ranks
    assert g["sometimes.weight"] is not None and g["sometimes.weight"].abs().max() > 0


def _fake_sysfs(tmp_path, bdf, node, cpulist, l3=None, other_gpus=()):
    """l3: {cpu: "shared_cpu_list"}; other_gpus: (bdf, node) of further AMD GPUs of the host."""
    for name, nn in ((bdf, node),) + tuple(other_gpus):
        d = tmp_path / "bus/pci/devices" / name
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{nn}\n")
        (d / "vendor").write_text("0x1002\n")
        (d / "class").write_text("0x120000\n")
    nic = tmp_path / "bus/pci/devices/0000:01:00.0"   # not a GPU: never counted
    nic.mkdir(parents=True)
    (nic / "numa_node").write_text(f"{max(node, 0)}\n")
    (nic / "vendor").write_text("0x15b3\n")
    (nic / "class").write_text("0x020000\n")
    if node >= 0:
        n = tmp_path / f"devices/system/node/node{node}"
        n.mkdir(parents=True)
        (n / "cpulist").write_text(cpulist + "\n")
    for cpu, shared in (l3 or {}).items():
        c = tmp_path / f"devices/system/cpu/cpu{cpu}/cache/index3"
        c.mkdir(parents=True)
        (c / "shared_cpu_list").write_text(shared + "\n")
    return str(tmp_path)


def test_cpulist_parser():
    from vlnce_amd.distributed import _parse_cpulist

    assert _parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert _parse_cpulist("5") == {5}
    assert _parse_cpulist("") == set()


def test_bind_host_threads_to_gpu_socket(tmp_path, monkeypatch):
    """One process per GPU on the socket its GPU hangs off: the node comes from the GPU's PCI
    address, every thread of the process gets the node's CPUs (intersected with the mask the
    process already had), and unknown topologies change nothing."""
    import os
    import types

    from vlnce_amd import distributed as D

    monkeypatch.delenv("VLNCE_BIND_SOCKET", raising=False)
    before = os.sched_getaffinity(0)
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0xD9, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: props)
    keep = sorted(before)[: max(1, len(before) // 2)]
    sysfs = _fake_sysfs(tmp_path / "a", "0000:d9:00.0", 1, ",".join(str(c) for c in keep) + ",100000")
    try:
        assert D.gpu_numa_node(0, sysfs) == 1
        # the default scope is the whole socket (worker processes inherit the mask)
        monkeypatch.setattr(D, "_AFFINITY_BEFORE_BIND", None)
        assert D.bind_host_threads_to_gpu_socket(0, sysfs=sysfs) == 1
        assert os.sched_getaffinity(0) == set(keep)
        for tid in os.listdir("/proc/self/task"):
            assert os.sched_getaffinity(int(tid)) == set(keep)
        # ... and a worker's init function gets the trainer's original mask back
        assert D.restore_worker_affinity() == before
        assert os.sched_getaffinity(0) == before
    finally:
        for tid in os.listdir("/proc/self/task"):
            os.sched_setaffinity(int(tid), before)
    # single-socket hosts report -1; a node whose CPUs are all outside the mask; the switch
    assert D.bind_host_threads_to_gpu_socket(0, sysfs=_fake_sysfs(tmp_path / "b", "0000:d9:00.0", -1, "")) is None
    assert D.bind_host_threads_to_gpu_socket(0, sysfs=_fake_sysfs(tmp_path / "c", "0000:d9:00.0", 0, "100000-100003")) is None
    monkeypatch.setenv("VLNCE_BIND_SOCKET", "0")
    assert D.bind_host_threads_to_gpu_socket(0, sysfs=sysfs) is None
    assert os.sched_getaffinity(0) == before


def test_bind_scope_l3_spreads_the_gpus_of_a_socket_over_its_l3_domains(tmp_path, monkeypatch):
    """scope "l3" (what bench.py asks for): ONE L3 domain of the GPU's socket; the GPUs of a socket
    (PCI order, other jobs' GPUs included -- sysfs shows the whole host) get different domains."""
    import os
    import types

    from vlnce_amd import distributed as D

    monkeypatch.delenv("VLNCE_BIND_SOCKET", raising=False)
    before = os.sched_getaffinity(0)
    cpus = sorted(before)
    if len(cpus) < 4:
        pytest.skip("needs four usable CPUs")
    a, b = cpus[:2], cpus[2:4]                       # two fake L3 domains of two CPUs each
    as_list = lambda xs: ",".join(str(c) for c in xs)
    l3 = {a[0]: as_list(a), a[1]: as_list(a), b[0]: as_list(b), b[1]: as_list(b)}
    gpus = (("0000:c9:00.0", 1), ("0000:05:00.0", 0))  # one more GPU on node 1 (before ours), one on node 0
    sysfs = _fake_sysfs(tmp_path, "0000:d9:00.0", 1, as_list(a + b), l3=l3, other_gpus=gpus)
    assert D._l3_domains(set(a + b), sysfs) == [set(a), set(b)]
    assert D._gpu_slot_on_node("0000:d9:00.0", 1, sysfs) == (1, 2)
    assert D._gpu_slot_on_node("0000:c9:00.0", 1, sysfs) == (0, 2)
    try:
        for bus, want in ((0xD9, b), (0xC9, a)):
            props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=bus, pci_device_id=0)
            monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i, p=props: p)
            for tid in os.listdir("/proc/self/task"):
                os.sched_setaffinity(int(tid), before)
            assert D.bind_host_threads_to_gpu_socket(0, sysfs=sysfs, scope="l3") == 1
            assert os.sched_getaffinity(0) == set(want)
        for tid in os.listdir("/proc/self/task"):
            os.sched_setaffinity(int(tid), before)
        assert D.bind_host_threads_to_gpu_socket(0, sysfs=sysfs, scope="socket") == 1
        assert os.sched_getaffinity(0) == set(a + b)
    finally:
        for tid in os.listdir("/proc/self/task"):
            os.sched_setaffinity(int(tid), before)
