"""policy-steps/sec of the VLN-CE policy hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*) the process is one of the N ranks; started plainly, `python bench.py
--gpus N` launches the N ranks itself (self_launch) and rank 0 prints the one JSON line.

One "step" = one DAgger inner-loop update (BaseVLNCETrainer._update_agent,
base_il_trainer.py:134-180: build_distribution on raw frames, inflection-
weighted cross-entropy, backward, Adam step) of the CMA policy over a batch of
num_envs=64 rollouts PER GPU of synthetic 256x256 RGB + 256x256 depth +
80-token instructions, resident in HBM before the timed region.  The policy is
exactly as the reference constructs it: frozen visual encoders whose
BatchNorm runs on batch statistics.  value = envs
processed by all ranks per second (weak scaling, data parallel; gradients are
all-reduced over RCCL by vlnce_amd.distributed when N > 1).  `value` is the loop the UNCHANGED
trainers issue: one update call per batch, every step running its own trunk passes, tail forward,
backward and Adam, over 4 distinct resident batches in rotation.  Beside it
(config.encode_ahead_*) the same steps with the optional policy.encode_ahead() API: the frozen
trunks of batch k+1 issued on side HIP streams before batch k's update is enqueued.

The JSON line also carries `roofline` (dominant kernel = the implicit-GEMM convolution:
conv_p3 / conv_u3 / conv_s3 / conv_x3 / conv_m3 / stem7 kernels, fp32 operands split into 16-bit planes
(round 6 default: two fp16 planes per activation, three plane products per multiply; VLNCE_CONV_MATH=bf16:
three bf16 planes, six products) on the 16-bit matrix pipe, plus the fp32-MFMA igemm_kernel
for the depth stem and the small layers; every launch timed with HIP events on the launch stream and
attributed to the kernel the library dispatched it to; `frac` prices the ALGORITHMIC fp32 FLOPs
against the fp32 MFMA peak, `bf16_pipe.frac` the hardware FLOPs of the bf16-plane launches
against the bf16 MFMA peak) and, on rank 0 at N=1, `cpu_baseline` (the CPU oracle restatement of the reference
policy timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 / _f16, dense
HBM_PEAK_TBS = 8.0           # MI355X_MICROARCH.md: HBM3E spec
HBM_ACHIEVABLE_TBS = 6.3     # MI355X_MICROARCH.md: measured float4 copy (79 % of spec)
# algorithmic FLOPs per policy-step (one env), CMA 256x256 L=80
CMA_FWD_GFLOP = 11.461
CMA_FWD_BWD_FROZEN_GFLOP = 11.630
CONV_GFLOP_PER_ENV = 10.677 + 0.699  # RGB ResNet-50 + depth ResNet-50 trunks (conv MACs x2)
STEM_GFLOP_PER_ENV = 0.308 + 0.051   # 7x7/s2 stems at 256x256 (3->64 and 1->32 channels)


# Set by a TEST driver only (tests/bench_on_cpu_simulator.py): the library binding it has put behind
# the package, named in the JSON line's `data` field.  With it the ranks run on the CPU over gloo:
# the launcher, the rank set-up, the sharded update through GradientAllReducer and the
# max-over-ranks timing are this file's real code, nothing is measured (value = None, no roofline).
# The product harness itself imports nothing from tests/.
SIMULATED_BACKEND = None


def plane_products():
    """16-bit MFMA products the plane kernels issue per fp32 multiply: 3 in plane format 2 (fp16
    planes, the default), 6 in format 1 (three bf16 planes; VLNCE_CONV_MATH=bf16)"""
    from vlnce_amd import ops
    return 3 if ops.plane_format() == 2 else 6


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def synth_batch(N, hw, L, device, seed=1):
    g = torch.Generator().manual_seed(seed)
    obs = {"rgb": torch.randint(0, 256, (N, hw, hw, 3), generator=g).float(),
           "depth": torch.rand(N, hw, hw, 1, generator=g),
           "instruction": torch.zeros(N, 200, dtype=torch.long)}
    obs["instruction"][:, :L] = torch.randint(1, 2504, (N, L), generator=g)
    prev = torch.randint(0, 4, (N, 1), generator=g)
    masks = (torch.rand(N, 1, generator=g) > 0.1).to(torch.uint8)
    targets = torch.randint(0, 4, (1, N), generator=g)
    weights = torch.rand(1, N, generator=g) + 0.5
    mv = lambda t: t.to(device)  # noqa: E731
    return ({k: mv(v) for k, v in obs.items()}, mv(prev), mv(masks), mv(targets), mv(weights))


def host_cpu_facts():
    """CPU model, sockets, physical cores and logical CPUs of this host (/proc/cpuinfo), plus
    the logical CPUs this process may run on."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
                phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"model": model, "sockets": len({p for p, _ in phys}) or None,
            "physical_cores": len(phys) or None, "logical_cpus": logical or None,
            "usable_logical_cpus": usable}


_UNBOUND_AFFINITY = None  # the process's CPU mask before the rank was bound to its GPU's socket


class whole_host:
    """The CPU baseline is the host's best, not one socket's: its child process starts from the
    mask this process had before bind_host_threads_to_gpu_socket()."""

    def __enter__(self):
        self.bound = os.sched_getaffinity(0) if _UNBOUND_AFFINITY else None
        if self.bound:
            os.sched_setaffinity(0, _UNBOUND_AFFINITY)

    def __exit__(self, *exc):
        if self.bound:
            os.sched_setaffinity(0, self.bound)


def cpu_baseline_worker(num_envs, hw, L, threads, which="cma"):
    """Runs in a child process: CPU oracle (port of the reference policy) timed on the host
    cores on a bounded sample of the bench workload.  `threads` is a comma list: each count
    gets 1 warm-up + 2 timed iterations and the best count is reported (oneDNN/OpenMP on a
    many-core shared host is not monotone in the thread count).  which: cma | seq2seq (one
    `_update_agent`) | waypoint (one WDDPPO minibatch update, encoders in eval mode)."""
    from oracle import policy_cpu as oc
    from oracle import thirdparty as tp

    n = num_envs   # the bench workload itself (num_envs = 64): ~3-6 s per iteration on this host
    oc.AuxLosses.activate()
    if which == "waypoint":
        pol = oc.WaypointPolicy.from_config(tp.make_config("WaypointPolicy"),
                                            *tp.make_spaces(hw, hw, pano=True))
        pol.train()
        pol.net.rgb_encoder.eval()
        pol.net.depth_encoder.eval()
        opt = torch.optim.Adam([q for q in pol.parameters() if q.requires_grad], lr=2.5e-4)
        obs, prev, ex = synth_pano_batch(n, hw, L, "cpu")
        masks = torch.ones(n, 1, dtype=torch.uint8)
        h0 = torch.zeros(n, pol.net.num_recurrent_layers,
                         pol.net.model_config.STATE_ENCODER.hidden_size)
        torch.set_num_threads(int(str(threads).split(",")[0]))
        with torch.no_grad():   # action components inside the truncated-normal supports
            out = pol.act(obs, h0, {k: v.clone() for k, v in prev.items()}, masks, deterministic=True)
        actions = {k: v.clone() for k, v in out[2].items()}
        actions["pano"] = ex["pano_action"]
        sample = (obs, h0, actions, prev, ex["value_preds"], ex["value_preds"] + 0.3, masks,
                  torch.full((n, 1), -2.0), ex["adv"])

        def one_update():
            smp = list(sample)
            smp[3] = {k: v.clone() for k, v in prev.items()}
            oc.ppo_update(pol, opt, tuple(smp))
        what = (f"WaypointPolicy WDDPPO minibatch update (oracle/policy_cpu.py), {n} envs x 13 "
                f"frames {hw}x{hw} RGB-D, L={L}")
    else:
        name = "CMAPolicy" if which == "cma" else "Seq2SeqPolicy"
        pol = getattr(oc, name).from_config(tp.make_config(name), *tp.make_spaces(hw, hw))
        opt = torch.optim.Adam(pol.parameters(), lr=2.5e-4)
        obs, prev, masks, tgt, w = synth_batch(n, hw, L, "cpu")

        def one_update():
            oc.il_update(pol, opt, obs, prev, masks, tgt, w, 512)
        what = (f"{'CMA' if which == 'cma' else 'Seq2Seq'} fwd+bwd+Adam (oracle/policy_cpu.py), "
                f"{n} envs x {hw}x{hw} RGB-D, L={L}")
    sweep = {}
    t_start = time.time()
    for th in [int(t) for t in str(threads).split(",")]:
        # bounded (the default bench run finishes within minutes): counts come in ascending
        # order and the sweep ends once a larger count is slower than the best so far -- on the
        # 2 x 64-core hosts 32 threads win and 128 threads cost 12-17 s per iteration
        if sweep and (time.time() - t_start > 60 or sweep[max(sweep)] > 1.1 * min(sweep.values())):
            break
        torch.set_num_threads(th)
        times = []
        for i in range(3):
            t0 = time.time()
            one_update()
            dt = time.time() - t0
            log(f"cpu_baseline {th} threads iter {i}: {dt:.2f}s")
            if i > 0:
                times.append(dt)
        sweep[th] = min(times)
    best_th = min(sweep, key=sweep.get)
    facts = host_cpu_facts()
    print(json.dumps({
        "value": round(n / sweep[best_th], 2), "unit": "policy-steps/sec", "cores": best_th,
        "physical_cores": facts["physical_cores"], "kind": "port",
        "sample": f"{what}, "
                  f"min of 2 iters after 1 warm-up, torch CPU fp32, best of thread counts "
                  f"{sorted(sweep)} = {best_th} threads",
        "host_cpu": facts,
        "steps_per_sec_by_threads": {str(k): round(n / v, 2) for k, v in sorted(sweep.items())}}))


def cpu_baseline(num_envs, hw, L, timeout_s=150, which="cma"):
    """Bounded: the child is killed after `timeout_s` (the default bench run must finish within
    minutes).  The child sweeps a few thread counts up to the host's physical cores and reports
    the best one as `cores`; the host's CPU model / socket / core counts ride along in
    `host_cpu` (north_star: "core count stated")."""
    import subprocess

    with whole_host():
        return _cpu_baseline(num_envs, hw, L, timeout_s, subprocess, which)


def _cpu_baseline(num_envs, hw, L, timeout_s, subprocess, which="cma"):
    facts = host_cpu_facts()
    usable = facts["usable_logical_cpus"]
    phys = min(facts["physical_cores"] or usable, usable)
    counts = sorted({min(c, usable) for c in (32, 64, phys)})
    threads = max(counts)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--num-envs",
           str(num_envs), "--hw", str(hw), "--tokens", str(L), "--policy", which, "--threads",
           ",".join(str(c) for c in counts)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        sys.stderr.write(r.stderr[-2000:])
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # timeout / parse error: report, never block the GPU line
        return {"value": None, "unit": "policy-steps/sec", "cores": threads, "kind": "port",
                "sample": f"cpu baseline did not finish within {timeout_s}s ({type(e).__name__})",
                "host_cpu": facts}


def f32_mfma_compare(args):
    """The same bench with every convolution on v_mfma_f32_32x32x2_f32 (VLNCE_CONV_MATH=f32; the
    switch is read once per process, hence a child process): what the step costs without the
    bf16-plane kernel.  Reported beside `value`, never as `value`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--num-envs", str(args.num_envs), "--no-cpu-baseline",
           "--no-f32-compare"]
    try:
        out = subprocess.run(cmd, env=dict(os.environ, VLNCE_CONV_MATH="f32"), capture_output=True,
                             text=True, timeout=600).stdout.strip().splitlines()
        d = json.loads(out[-1])
        return {"value": d["value"], "ms_per_step": d["ms_per_step"],
                "encode_ahead_ms_per_step": d["config"]["encode_ahead_ms_per_step"],
                "roofline_frac_of_fp32_mfma_peak": d["roofline"]["fp32_mfma_peak"]["frac"],
                "conv_kernel_ms_per_step": d["roofline"]["kernel_ms_per_step"]}
    except Exception as e:  # never block the main line
        return {"value": None, "error": type(e).__name__}


def pmc_traffic(n_conv):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes of this same
    workload (profiles/r06_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate
    passes of `bench.py --pmc-step`, gfx950 corrections applied as MI355X_MICROARCH.md
    prescribes).  None when the file is absent or was taken for a different launch count."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    rec = None
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json"):   # newest first
        try:
            rec = json.load(open(os.path.join(prof, name)))
            break
        except OSError:
            continue
    if rec is None:
        return None
    if rec.get("conv_launches_per_step") != n_conv:
        return None
    return rec.get("hbm_bytes_per_launch")


def conv_kernel_time(policy, obs, dev, repeats=3, trunk_pass=None):
    """GPU-paced duration of every convolution launch of the two visual trunks' forward
    (the 107 convolution launches of a step).

    The trunks run eagerly on ONE stream (graph replay hides the individual launches from
    the host; concurrent branches would stretch each other's kernels) with a HIP event
    before and after each conv launch.  Events are device-side timestamps, but an event
    pair also contains whatever time the stream sat idle waiting for the HOST to issue the
    launch -- on a slow host that idle time used to be booked as kernel time.  So the pass is
    issued behind a spin kernel that keeps the stream busy until the host has queued
    everything: all launches then execute back to back and the pairs measure device time
    only.  The marker cost of an empty event pair (measured in the same backlog) is
    subtracted per launch.  Checks: every per-launch time is the minimum over `repeats`
    passes; the sum must not exceed the single-stream duration of the whole eager pass
    (one event pair around everything) -- otherwise None is returned with the reason."""
    import vlnce_amd.encoders.resnet_encoders as enc
    from vlnce_amd import ops

    saved = {k: os.environ.get(k) for k in ("VLNCE_HIP_GRAPHS", "VLNCE_SIDE_STREAMS")}
    os.environ["VLNCE_HIP_GRAPHS"] = "0"
    os.environ["VLNCE_SIDE_STREAMS"] = "0"
    orig_conv = ops.conv2d_nhwc
    orig_conv_bn = ops.conv2d_bn_sums   # convolution that also adds its BatchNorm column sums
    events = []

    meta = []  # per launch: (algorithmic FLOPs, kernel path the library dispatched to)

    def _timed(call, x, w, k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = call()
        e1.record()
        events.append((e0, e1))
        y = out[0] if isinstance(out, tuple) else out
        # algorithmic HBM bytes of the launch: input (+ second input and the materialised block
        # output of a fused block end), weights, output -- each once
        nbytes = 4.0 * (x.numel() + w.numel() + y.numel())
        if k.get("x2") is not None:
            nbytes += 4.0 * x.numel() * (2 if k.get("side_out") is not None else 1)
        meta.append((2.0 * y.numel() * w[0].numel(), ops.L().conv2d_last_path(), nbytes))
        return out

    def timed_conv(x, w, stride, pad, *a, **k):
        return _timed(lambda: orig_conv(x, w, stride, pad, *a, **k), x, w, k)

    def timed_conv_bn(x, w, stride, pad, acc, **k):
        return _timed(lambda: orig_conv_bn(x, w, stride, pad, acc, **k), x, w, k)

    orig_stem7 = ops.stem7   # the 7x7 RGB stem from the frames (one of the conv launches)

    def timed_stem7(fr, wf, Cout, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_stem7(fr, wf, Cout, *a, **k)
        e1.record()
        events.append((e0, e1))
        frame_bytes = fr["images"] * fr["H"] * fr["W"] * 3 * fr["x"].element_size()
        meta.append((2.0 * y.numel() * 147, 2, frame_bytes + 4.0 * y.numel() + 4.0 * Cout * 147))
        return y

    def trunks():
        with torch.no_grad():
            if trunk_pass is not None:   # (a policy whose encoders take something else than `obs`)
                trunk_pass()
                return
            policy.net.rgb_encoder.trunk_features(obs)
            policy.net.depth_encoder.trunk_features(obs)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    try:
        trunks()  # eager warm-up of this mode
        torch.cuda.synchronize()
        # host issue time of one eager pass and the spin kernel's rate
        t0 = time.perf_counter()
        trunks()
        host_ms = 1e3 * (time.perf_counter() - t0)
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        torch.cuda._sleep(2_000_000)
        b.record()
        torch.cuda.synchronize()
        cyc_per_ms = 2_000_000 / max(a.elapsed_time(b), 1e-3)
        backlog_ms = min(3.0 * host_ms + 30.0, 2000.0)
        per_launch, totals, empties = None, [], []
        enc.ops.conv2d_nhwc = timed_conv
        ops.conv2d_bn_sums = timed_conv_bn
        ops.stem7 = timed_stem7
        for _ in range(repeats):
            events.clear()
            meta.clear()
            torch.cuda._sleep(int(backlog_ms * cyc_per_ms))
            nul = [(ev(), ev()) for _ in range(8)]
            for e0, e1 in nul:
                e0.record()
                e1.record()
            w0, w1 = ev(), ev()
            w0.record()
            trunks()
            w1.record()
            torch.cuda.synchronize()
            empties.append(sorted(e0.elapsed_time(e1) for e0, e1 in nul)[len(nul) // 2])
            cur = [e0.elapsed_time(e1) for e0, e1 in events]
            per_launch = cur if per_launch is None else [min(x, y) for x, y in zip(per_launch, cur)]
            totals.append(w0.elapsed_time(w1))
    finally:
        enc.ops.conv2d_nhwc = orig_conv
        ops.conv2d_bn_sums = orig_conv_bn
        ops.stem7 = orig_stem7
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    empty = min(empties)
    n = len(per_launch)
    conv_ms = sum(max(t - empty, 0.0) for t in per_launch)
    whole = min(totals)
    log(f"conv attribution: {n} launches, {conv_ms:.3f} ms of a {whole:.3f} ms eager single-stream "
        f"trunk pair (empty event pair {1e3 * empty:.1f} us, host issue {host_ms:.1f} ms)")
    by_path = {}
    # per-launch floor = max(time of its algorithmic bytes at the achievable HBM rate, time of its
    # instruction FLOPs on the pipe it runs on); `hbm_bound` = launches whose floor is the HBM term
    floor_ms, hbm_bound_ms, hbm_bound_n, hbm_floor_ms = 0.0, 0.0, 0, 0.0
    for t, (fl, path, nb) in zip(per_launch, meta):
        d = by_path.setdefault(path, {"launches": 0, "ms": 0.0, "flop": 0.0})
        d["launches"] += 1
        d["ms"] += max(t - empty, 0.0)
        d["flop"] += fl
        t_hbm = nb / (HBM_ACHIEVABLE_TBS * 1e12) * 1e3
        t_pipe = (fl / (FP32_MFMA_PEAK_TFLOPS * 1e12) if path == 0
                  else plane_products() * fl / (BF16_MFMA_PEAK_TFLOPS * 1e12)) * 1e3
        floor_ms += max(t_hbm, t_pipe)
        if t_hbm >= t_pipe:
            hbm_bound_n += 1
            hbm_bound_ms += max(t - empty, 0.0)
            hbm_floor_ms += t_hbm
    res = {"n": n, "conv_ms": conv_ms, "eager_trunks_ms": whole, "empty_pair_us": 1e3 * empty,
           "host_issue_ms": host_ms, "reason": None, "by_path": by_path,
           "flop": sum(m[0] for m in meta), "bytes": sum(m[2] for m in meta),
           "floor_ms": floor_ms, "hbm_bound_n": hbm_bound_n, "hbm_bound_ms": hbm_bound_ms,
           "hbm_floor_ms": hbm_floor_ms}
    if not (0.0 < conv_ms <= whole * 1.001):
        res["reason"] = (f"per-launch event sum {conv_ms:.3f} ms is not inside the eager "
                         f"single-stream pass {whole:.3f} ms")
    return res


def act_latency(policy, batch, dev, sizes=(1, 4, 8), iters=20):
    """forward-only act() (eval, no_grad) at the small batches of the eval / inference loops
    (base_il_trainer.py:284-331): mean wall time per call in ms, graphs as the policy uses them."""
    obs, prev, masks = batch[0], batch[1], batch[2]
    out = {}
    with torch.no_grad():
        for n in sizes:
            if n > prev.size(0):
                continue
            o = {k: v[:n].contiguous() for k, v in obs.items()}
            h0 = torch.zeros(n, policy.net.num_recurrent_layers, 512, device=dev)
            for _ in range(4):  # eager pass, capturing pass, replays
                policy.act(o, h0, prev[:n], masks[:n], deterministic=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                policy.act(o, h0, prev[:n], masks[:n], deterministic=True)
            torch.cuda.synchronize()
            out[str(n)] = round(1e3 * (time.perf_counter() - t0) / iters, 3)
    return out


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE in the environment):
    re-run this very command line as N ranks, one per GPU, under torch.distributed.run on this
    node -- how the reference starts its distributed trainer (`python -u -m
    torch.distributed.launch --use_env --nproc_per_node N run.py ...`,
    sbatch_scripts/waypoint_train_single_node.sh:24-30; init at ddppo_waypoint_trainer.py:310-312).
    The ranks inherit stdout: rank 0's JSON line is the one line printed there.  Returns the
    launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(sys.argv[0])] + sys.argv[1:]   # (this file, or the test driver around it)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL)
    env.setdefault("OMP_NUM_THREADS", "8")
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:9])} ...")
    return subprocess.call(cmd, env=env)


def exchange_fields(ex):
    """`allreduce_ms` / `allreduce_hidden_frac` of the JSON line (world > 1 or --force-dist): the
    bucket collectives of the last timed step on rank 0 and the share of them that ran before
    backward had ended (GradientAllReducer.stats)."""
    if not ex:
        return {}
    return {"allreduce_ms": round(ex["allreduce_ms"], 3),
            "allreduce_hidden_frac": (round(ex["hidden_frac"], 4) if ex["hidden_frac"] is not None
                                      else None),
            "allreduce": {"buckets": ex["buckets"], "issued_from_backward_hooks": ex["issued_from_hooks"],
                          "exposed_ms": round(ex["exposed_ms"], 3),
                          "what": "rank 0, last timed step; hidden = ran under backward"}}


def synth_pano_batch(n, hw, tokens, device, seed=1):
    """WaypointPolicy inputs (ddppo_waypoint_trainer.py:283-306 after ObsStack): 12 panorama
    frames + the history frame per env, angle features, a `tokens`-token instruction."""
    g = torch.Generator().manual_seed(seed)
    obs = {"rgb": torch.randint(0, 256, (n, 12, hw, hw, 3), generator=g).float(),
           "depth": torch.rand(n, 12, hw, hw, 1, generator=g),
           "rgb_history": torch.randint(0, 256, (n, hw, hw, 3), generator=g).float(),
           "depth_history": torch.rand(n, hw, hw, 1, generator=g),
           "angle_features": torch.randn(n, 12, 4, generator=g),
           "instruction": torch.zeros(n, 200, dtype=torch.long)}
    obs["instruction"][:, :tokens] = torch.randint(1, 2504, (n, tokens), generator=g)
    prev = {"pano": torch.randint(0, 12, (n, 1), generator=g),
            "offset": (torch.rand(n, 1, generator=g) - 0.5) * 0.4,
            "distance": 0.25 + torch.rand(n, 1, generator=g) * 2.0}
    extra = {"pano_action": torch.randint(0, 12, (n, 1), generator=g),
             "value_preds": torch.randn(n, 1, generator=g) * 0.5,
             "adv": torch.randn(n, 1, generator=g)}
    mv = lambda t: t.to(device)  # noqa: E731
    return ({k: mv(v) for k, v in obs.items()}, {k: mv(v) for k, v in prev.items()},
            {k: mv(v) for k, v in extra.items()})


def secondary_policy_bench(args, dev, rank, world, sim, use_dist, dev_sync):
    """`--policy seq2seq | waypoint`: BASELINE.json configs[1] / configs[4] under the same launcher,
    rank set-up, barrier + max-over-ranks timing and GradientAllReducer as the headline.  One step =
    one `_update_agent` (Seq2Seq; base_il_trainer.py:134-180) or one WDDPPO minibatch update
    (ddppo_alg.py:53-141: evaluate_actions, the clipped losses, backward, gradient exchange,
    clipping, Adam; encoders in eval mode as ddppo_waypoint_trainer.py:528-530 puts them; the
    policy's unused `action_distribution` head never receives a gradient, :370), every step
    ending with the host read-backs the reference makes."""
    import vlnce_amd
    from vlnce_amd.il_harness import update_agent
    from vlnce_amd.ppo_harness import PPOConfig, wddppo_minibatch_update

    torch.manual_seed(0)
    n, hw, L = args.num_envs, args.hw, args.tokens
    way = args.policy == "waypoint"
    name = "WaypointPolicy" if way else "Seq2SeqPolicy"
    policy = vlnce_amd.build_model(vlnce_amd.make_config(name),
                                   *vlnce_amd.make_spaces(hw, hw, pano=way)).to(dev)
    if way:
        policy.train()
        policy.net.rgb_encoder.eval()
        policy.net.depth_encoder.eval()
        opt = torch.optim.Adam([p for p in policy.parameters() if p.requires_grad], lr=2.5e-4)
    else:
        opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
    reducer = grad_hook = None
    if use_dist:
        from vlnce_amd.distributed import GradientAllReducer

        reducer = GradientAllReducer(policy, timing=True)
        grad_hook = reducer.finish
    vlnce_amd.AuxLosses.activate()
    NB = 2 if way else 4   # distinct resident batches in rotation (a waypoint batch is 330 MB)
    it = [0]
    if way:
        hid = policy.net.model_config.STATE_ENCODER.hidden_size
        samples = []
        for i in range(NB):
            obs, prev, ex = synth_pano_batch(n, hw, L, dev, seed=1 + rank + 101 * i)
            masks = torch.ones(n, 1, dtype=torch.uint8, device=dev)
            h0 = torch.zeros(n, policy.net.num_recurrent_layers, hid, device=dev)
            with torch.no_grad():   # action components inside the truncated-normal supports
                out = policy.act(obs, h0, {k: v.clone() for k, v in prev.items()}, masks,
                                 deterministic=True)
            actions = {k: v.clone() for k, v in out[2].items()}
            actions["pano"] = ex["pano_action"]
            samples.append((obs, h0, actions, prev, ex["value_preds"], ex["value_preds"] + 0.3,
                            masks, torch.full((n, 1), -2.0, device=dev), ex["adv"]))

        def step():
            it[0] += 1
            s = list(samples[it[0] % NB])
            s[3] = {k: v.clone() for k, v in s[3].items()}   # the policy mutates prev_actions
            stats = wddppo_minibatch_update(policy, opt, tuple(s), PPOConfig(), grad_hook=grad_hook)
            return [v.item() for v in stats]   # ddppo_alg.py:133-140 accumulates six .item()s
        frames = 13
    else:
        batches = [synth_batch(n, hw, L, dev, seed=1 + rank + 101 * i) for i in range(NB)]

        def step():
            it[0] += 1
            obs, prev, masks, tgt, w = batches[it[0] % NB]
            return update_agent(policy, opt, obs, prev, masks, tgt, w, 512, grad_hook=grad_hook)
        frames = 1

    def sync():
        if use_dist:
            dist.barrier()
        dev_sync()

    log(f"{name} built, starting warm-up")
    for i in range(args.warmup):
        t0 = time.perf_counter()
        step()
        dev_sync()
        log(f"warm-up step {i}: {1e3 * (time.perf_counter() - t0):.1f} ms")
    sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    sync()
    elapsed = time.perf_counter() - t0
    exchange = None
    if use_dist:
        exchange = reducer.stats()
        tmax = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    log(f"timed region: {args.steps} steps in {elapsed:.3f}s; last step's read-backs {last}")
    roof = None
    if rank == 0 and not sim:
        if way:   # the 12 panorama frames + the masked history frame, as _encode_frames hands them over
            wobs, wmask = samples[0][0], samples[0][6].reshape(-1)

            def way_trunks():
                policy.net.rgb_encoder.trunk_features(
                    {"rgb": (wobs["rgb"], wobs["rgb_history"], wmask)})
                policy.net.depth_encoder.trunk_features(
                    {"depth": (wobs["depth"], wobs["depth_history"], wmask)})
            c = conv_kernel_time(policy, None, dev, trunk_pass=way_trunks)
        else:
            c = conv_kernel_time(policy, batches[0][0], dev)
        if c["reason"] is None:
            roof = {"schema": 3, "bound": "mfma",
                    "kernel": "conv2d fwd launches of the visual trunks (the plane kernels: "
                              f"{plane_products()} 16-bit MFMA products per fp32 multiply)",
                    "achieved": round(c["flop"] / (c["conv_ms"] * 1e-3) / 1e12, 2),
                    "peak": round(c["flop"] / (c["floor_ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                    "frac": round(c["floor_ms"] / c["conv_ms"], 4),
                    "peak_is": "per-launch max(algorithmic bytes / 6.3 TB/s, instruction FLOPs / the "
                               "pipe's peak) floor, as in the headline line; frac = floor_ms / "
                               "kernel_ms_per_step",
                    "floor_ms": round(c["floor_ms"], 3),
                    "algorithmic_GFLOP": round(c["flop"] / 1e9, 1),
                    "algorithmic_GB": round(c["bytes"] / 1e9, 3),
                    "hbm_bound": {"launches": c["hbm_bound_n"], "ms": round(c["hbm_bound_ms"], 3),
                                  "floor_ms": round(c["hbm_floor_ms"], 3)},
                    "by_kernel": {{0: "fp32_mfma", 1: "planes_x3", 2: "planes_p3", 3: "planes_m3"}.get(k, str(k)):
                                  {"launches": v["launches"], "ms": round(v["ms"], 3),
                                   "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] else None}
                                  for k, v in c["by_path"].items()},
                    "launches_per_step": c["n"], "kernel_ms_per_step": round(c["conv_ms"], 3),
                    "eager_single_stream_trunks_ms": round(c["eager_trunks_ms"], 3),
                    "timing": "HIP events per launch behind a device-side backlog (GPU-paced), "
                              "empty-pair cost subtracted, min of 3 passes",
                    "traffic": None}
        else:
            roof = {"schema": 3, "invalid_reason": c["reason"]}
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        line = {
            "metric": "policy-steps/sec (fwd+bwd)",
            "value": None if sim else round(n * world * args.steps / elapsed, 1),
            "unit": "policy-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": (f"{SIMULATED_BACKEND} over gloo -- a test of "
                     "the launcher and the rank logic, NOT a measurement") if sim else "synthetic",
            "config": {"workload": (
                f"WaypointPolicy WDDPPO minibatch update (evaluate_actions + losses + backward + "
                f"clip + Adam), encoders in eval mode, num_envs={n}/GPU x 13 frames "
                f"{hw}x{hw} RGB-D, {L}-token instruction" if way else
                f"Seq2Seq policy DAgger update (fwd+bwd+Adam), num_envs={n}/GPU, {hw}x{hw} RGB-D, "
                f"{L}-token instruction") + f", {NB} distinct batches in rotation",
                "global_batch": n * world, "parallelism": f"dp{world}",
                "frames_per_sec": None if sim else round(frames * n * world * args.steps / elapsed, 1)},
            **exchange_fields(exchange)}
        if roof:
            line["roofline"] = roof
        if world == 1 and not sim and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, hw, L, timeout_s=240 if way else 150,
                                                which=args.policy)
        final = json.dumps(line)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(final, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--policy", choices=["cma", "seq2seq", "waypoint"], default="cma",
                    help="cma (default) = the headline workload (BASELINE.json configs[2]/[3]); "
                         "seq2seq = configs[1] (DAgger update, num_envs 32); waypoint = configs[4] "
                         "(WaypointPolicy WDDPPO minibatch update, num_envs 32 x 13 frames, 200 tokens): "
                         "the same rank logic, timing and gradient all-reducer, a shorter JSON line")
    ap.add_argument("--num-envs", type=int, default=None, help="per GPU; default 64 (cma) / 32")
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=None, help="default 80 (cma, seq2seq) / 200 (waypoint)")
    ap.add_argument("--bn", choices=["train", "eval"], default="train",
                    help="train = as constructed by the reference (batch statistics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-compare", action="store_true",
                    help="skip the child run with every convolution on the fp32 MFMA instruction")
    ap.add_argument("--trainable-encoders", action="store_true",
                    help="MODEL.{RGB,DEPTH}_ENCODER.trainable=True: trunks get dgrad/wgrad too")
    ap.add_argument("--pmc-step", action="store_true",
                    help="for rocprofv3 --pmc passes: 1 warm-up + 1 eager single-stream step, "
                         "nothing else")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--threads", default="8", help=argparse.SUPPRESS)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="skip the secondary encode_ahead measurement (`value` is the plain "
                         "trainer loop either way)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gradient all-reducer even with one rank "
                         "(exercises the N>1 code path on a 1-GPU box)")
    args = ap.parse_args()
    if args.num_envs is None:
        args.num_envs = 64 if args.policy == "cma" else 32
    if args.tokens is None:
        args.tokens = 200 if args.policy == "waypoint" else 80
    if args.cpu_baseline_only:
        cpu_baseline_worker(args.num_envs, args.hw, args.tokens, args.threads, args.policy)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    sim = SIMULATED_BACKEND is not None
    if sim:
        dev = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but this node shows "
                             f"{torch.cuda.device_count()} GPU(s)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        # one process per GPU, on one L3 domain of the CPU socket that GPU hangs off (what a
        # production launcher does with numactl); VLNCE_BIND_SOCKET=0 leaves it to the scheduler
        from vlnce_amd.distributed import bind_host_threads_to_gpu_socket

        global _UNBOUND_AFFINITY
        before = os.sched_getaffinity(0)
        # (scope "l3": this process forks no simulator / loader workers; the library default is
        # the whole socket)
        node = bind_host_threads_to_gpu_socket(
            local, scope=os.environ.get("VLNCE_BIND_SOCKET") or "l3")
        if node is not None:
            _UNBOUND_AFFINITY = before
            now = os.sched_getaffinity(0)
            log(f"rank {rank}: host threads bound to NUMA node {node} (cuda:{local}'s socket), "
                f"{len(now)} of {len(before)} CPUs"
                + (f" {sorted(now)}" if len(now) <= 32 else f" ({min(now)}..{max(now)})"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if sim:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def dev_sync():
        if not sim:
            torch.cuda.synchronize()

    import vlnce_amd
    from vlnce_amd import ops
    from vlnce_amd.il_harness import update_agent

    if args.policy != "cma":
        secondary_policy_bench(args, dev, rank, world, sim, use_dist, dev_sync)
        return
    torch.manual_seed(0)
    over = {}
    if args.trainable_encoders:
        over = {"RGB_ENCODER.trainable": True, "DEPTH_ENCODER.trainable": True}
    cfg = vlnce_amd.make_config("CMAPolicy", **over)
    if args.pmc_step:
        os.environ["VLNCE_HIP_GRAPHS"] = "0"
        os.environ["VLNCE_SIDE_STREAMS"] = "0"
    policy = vlnce_amd.build_model(cfg, *vlnce_amd.make_spaces(args.hw, args.hw)).to(dev)
    if args.bn == "eval":
        policy.net.rgb_encoder.eval()
    opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
    grad_hook = None
    if use_dist:
        from vlnce_amd.distributed import GradientAllReducer

        reducer = GradientAllReducer(policy, timing=True)
        grad_hook = reducer.finish
    vlnce_amd.AuxLosses.activate()
    # NB distinct resident batches, used in rotation: one batch fed every step would stay warm in
    # the 256 MiB Infinity Cache (67 MB of frames)
    NB = 4
    batches = [synth_batch(args.num_envs, args.hw, args.tokens, dev, seed=1 + rank + 101 * i)
               for i in range(NB)]
    batch = batches[0]
    it = [0]

    def next_batch():
        it[0] += 1
        return batches[it[0] % NB]

    losses = []  # (loss, action_loss, aux_loss) floats of every step, as the trainers log them

    def step():
        obs, prev, masks, tgt, w = next_batch()
        losses.append(update_agent(policy, opt, obs, prev, masks, tgt, w, 512, grad_hook=grad_hook))

    if os.environ.get("VLNCE_BENCH_CACHED_DEPTH"):  # diagnostic: how much the depth trunk costs
        with torch.no_grad():
            feats = policy.net.depth_encoder.trunk_features(batch[0]).clone()
        batch[0]["depth_features"] = feats

    # Frozen encoders do not depend on the weights an update changes, so the trunks of step
    # k+1 are issued (on their own streams) BEFORE step k's update is enqueued and overlap its
    # latency-bound tail.  Every step still runs its own trunk pass, tail forward, backward and
    # Adam; trainable encoders cannot run ahead and fall back to the plain loop.
    pipeline = not (args.no_pipeline or args.trainable_encoders or sim)

    def run_steps(n, ahead):
        if not ahead:
            for _ in range(n):
                step()
            return
        b = next_batch()
        nxt = policy.encode_ahead(b[0])
        for k in range(n):
            cur, (_, prev, masks, tgt, w) = nxt, b
            if k + 1 < n:
                b = next_batch()
                nxt = policy.encode_ahead(b[0])
            losses.append(update_agent(policy, opt, cur, prev, masks, tgt, w, 512,
                                       grad_hook=grad_hook))

    if args.pmc_step:
        # layout of the profiled run:  warm-up | marker | calibration copy (a known 256 MiB
        # read + 256 MiB write) | marker | the two visual trunks' forward, eager, one stream
        # (= exactly the conv launches the roofline line times) | marker | one whole step
        step()
        torch.cuda.synchronize()
        obs = batch[0]
        src = torch.randn(1 << 26, device=dev)
        torch.cuda._sleep(1000)
        dst = src.clone()
        torch.cuda._sleep(1000)
        with torch.no_grad():
            policy.net.rgb_encoder(obs)
            policy.net.depth_encoder(obs)
        torch.cuda._sleep(1000)
        step()
        torch.cuda.synchronize()
        del dst
        return
    log("policy built, starting warm-up")
    for i in range(args.warmup):
        t0 = time.perf_counter()
        step()
        dev_sync()
        log(f"warm-up step {i}: {1e3 * (time.perf_counter() - t0):.1f} ms")

    def sync():
        if use_dist:
            dist.barrier()
        dev_sync()

    # ---- the timed region: the loop the UNCHANGED trainers issue (base_il_trainer.py:134-180
    # per batch): build_distribution on raw frames -> loss -> backward -> Adam, one call per step
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps, ahead=False)
    sync()
    elapsed = time.perf_counter() - t0
    log(f"timed region (plain trainer loop): {args.steps} steps in {elapsed:.3f}s; "
        f"loss of the last step {losses[-1][0]:.5f} (every step ends with the reference's "
        f"loss.item() / action_loss.item() read-backs)")
    timed_losses = [l[0] for l in losses[-args.steps:]]
    from vlnce_amd import streams as _st
    if _st.TIMING:
        per = {}
        for idx, e0, e1 in _st.TIMING[-3 * args.steps:]:
            per.setdefault(idx, []).append(e0.elapsed_time(e1))
        log("side-stream branch durations (ms, mean over the timed launches): "
            + ", ".join(f"stream{idx}: {sum(v) / len(v):.2f} x{len(v)}" for idx, v in sorted(per.items())))
    exchange = None
    if use_dist:
        exchange = reducer.stats()   # the last timed step's gradient exchange on this rank
        tmax = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    if sim:
        if rank == 0:
            print(json.dumps({
                "metric": "policy-steps/sec (fwd+bwd)", "value": None, "unit": "policy-steps/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": f"{SIMULATED_BACKEND} over gloo -- a test "
                        "of the launcher and the rank logic, NOT a measurement",
                "config": {"workload": "launcher self-test", "global_batch": args.num_envs * world,
                           "parallelism": f"dp{world}"}, **exchange_fields(exchange)}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- beside it: the same steps with the optional policy.encode_ahead() API (the frozen
    # trunks of batch k+1 issued before batch k's update is enqueued), same step count
    ahead_ms = None
    if pipeline:
        run_steps(3, ahead=True)  # warms the run-ahead path (its graphs live on the side streams)
        sync()
        t1 = time.perf_counter()
        run_steps(args.steps, ahead=True)
        sync()
        ahead_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        if use_dist:
            tm = torch.tensor([ahead_ms], device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ahead_ms = tm.item()
        log(f"with encode_ahead: {ahead_ms:.3f} ms/step")

    # ---- dominant-kernel attribution: separate, untimed-for-throughput pass
    conv = conv_kernel_time(policy, batch[0], dev)
    # secondary figure: forward-only act() under no_grad in eval mode
    policy.eval()
    h0 = torch.zeros(args.num_envs, 2, 512, device=dev)
    with torch.no_grad():
        for _ in range(3):
            policy.act(batch[0], h0, batch[1], batch[2], deterministic=True)
        sync()
        ta = time.perf_counter()
        for _ in range(args.steps):
            policy.act(batch[0], h0, batch[1], batch[2], deterministic=True)
        sync()
        act_s = (time.perf_counter() - ta) / args.steps
    act_small = act_latency(policy, batch, dev) if rank == 0 else {}
    policy.train()

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = args.num_envs * world * args.steps / elapsed
        # the attributed launches are the forward convolutions of the two trunks (in the
        # trainable-encoder variant the data-gradient launches are the same kernel again)
        conv_flop = CONV_GFLOP_PER_ENV * 1e9 * args.num_envs
        step_gflop = CMA_FWD_BWD_FROZEN_GFLOP
        if args.trainable_encoders:
            step_gflop += 2 * CONV_GFLOP_PER_ENV - STEM_GFLOP_PER_ENV
        conv_ms, n_conv = conv["conv_ms"], conv["n"]
        ok = conv["reason"] is None
        achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if ok else None
        floor_rate = conv_flop / (conv["floor_ms"] * 1e-3) / 1e12 if ok else None
        # per kernel family, from the library's own dispatch record (vlnce_conv2d_last_path)
        fam = {0: "fp32_mfma", 1: "bf16_planes_x3", 2: "bf16_planes_p3", 3: "bf16_planes_m3"}
        paths = {fam.get(k, str(k)): v for k, v in conv["by_path"].items()}
        bf = [v for k, v in paths.items() if k.startswith("bf16")]
        bf_ms, bf_flop = sum(v["ms"] for v in bf), sum(v["flop"] for v in bf)
        f32 = paths.get("fp32_mfma", {"ms": 0.0, "flop": 0.0, "launches": 0})

        def tf(v):
            return round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None

        line = {
            "metric": "policy-steps/sec (fwd+bwd)", "value": round(value, 1),
            "unit": "policy-steps/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CMA policy DAgger update (fwd+bwd+Adam) as the unchanged "
                                   "trainers call it (one _update_agent per batch, no run-ahead), "
                                   + ("trainable" if args.trainable_encoders else "frozen")
                                   + " encoders, "
                                   f"BatchNorm={args.bn}, num_envs={args.num_envs}/GPU, "
                                   f"{args.hw}x{args.hw} RGB-D, {args.tokens}-token instruction, "
                                   f"{NB} distinct batches in rotation",
                       "global_batch": args.num_envs * world, "parallelism": f"dp{world}",
                       "host_threads": (
                           f"rank bound to one L3 domain of its GPU's CPU socket "
                           f"({len(os.sched_getaffinity(0))} of {len(_UNBOUND_AFFINITY)} CPUs; "
                           f"VLNCE_BIND_SOCKET=socket for the whole socket, 0 to leave it to the "
                           f"scheduler)" if _UNBOUND_AFFINITY else "placed by the scheduler"),
                       "host_readbacks_per_step": "loss.item(), action_loss.item() [, aux_loss.item()] "
                                                  "as base_il_trainer.py:176-180",
                       "loss_first_last_timed_step": [round(timed_losses[0], 5),
                                                      round(timed_losses[-1], 5)],
                       "whole_step_tflops": round(step_gflop * value / 1e3, 2),
                       "encode_ahead_ms_per_step": round(ahead_ms, 3) if ahead_ms else None,
                       "encode_ahead_steps_per_sec": (
                           round(1e3 * args.num_envs * world / ahead_ms, 1) if ahead_ms else None),
                       "encode_ahead_is": "optional API policy.encode_ahead(obs): the frozen "
                                          "trunks of the next batch issued on side streams before "
                                          "this batch's update; same results "
                                          "(tests/test_policy_gpu.py::test_encode_ahead_pipeline_"
                                          "equals_plain_loop); NOT what `value` measures",
                       "act_fwd_only_eval_steps_per_sec_per_gpu": round(args.num_envs / act_s, 1),
                       "act_latency_ms_by_num_envs": act_small},
            "roofline": {"schema": 3,
                         "schema_is": "3 (round 6): the plane kernels issue `bf16_pipe."
                                      "hw_flops_per_algorithmic_flop` 16-bit MFMA FLOPs per fp32 FLOP "
                                      "(3 with fp16 planes, 6 with bf16 planes) and the per-launch "
                                      "floor prices that count; 2 (round 5): `frac` / `peak` = the "
                                      "per-launch max(HBM, MFMA) floor, the fp32-MFMA-peak figure "
                                      "under `fp32_mfma_peak`; 1 (rounds 1-4): frac = achieved / fp32 "
                                      "MFMA peak",
                         "bound": "mfma",
                         "kernel": "conv2d fwd of the two visual trunks: conv_p3_kernel (stride-1 "
                                   "KxK: A transformed once per workgroup into an LDS patch, B "
                                   "fragments from L2), conv_u3_kernel (wide 1x1: no producer "
                                   "waves, the matrix waves transform the next K-chunk between "
                                   "their MFMAs), conv_s3_kernel (short-K 1x1 expansions: whole K "
                                   "resident, tile n+1's loads under tile n's stores), "
                                   "conv_x3_kernel (the other 1x1 / strided: im2col K-tiles through "
                                   "LDS, 8 producer + 8 matrix waves), conv_m3_kernel (small "
                                   "launches -- the depth trunk: A fragments straight from global "
                                   "memory, reduction split over a workgroup's waves) and "
                                   "stem7_kernel (the 7x7/s2 RGB stem straight from the frames); "
                                   "all six: fp32 operands "
                                   + ("split into fp16 planes (activations a1 + a2, weights b1 + b2, "
                                      "11 + 11 mantissa bits each), 3 x v_mfma_f32_32x32x16_f16 "
                                      "(a1 b1 + a1 b2 + a2 b1) per 32x32x16 block"
                                      if plane_products() == 3 else
                                      "split exactly into 3 bf16 planes, 6 x "
                                      "v_mfma_f32_32x32x16_bf16 per 32x32x16 block")
                                   + ", fp32 accumulate, "
                                   "train-mode BatchNorm column sums taken in the epilogue; "
                                   "igemm_kernel (v_mfma_f32_32x32x2_f32) for the depth stem and "
                                   "the handful-of-tiles layers",
                         "achieved": round(achieved, 2) if ok else None,
                         "peak": round(floor_rate, 2) if ok else None, "unit": "TFLOP/s",
                         "frac": round(conv["floor_ms"] / conv_ms, 4) if ok else None,
                         "peak_is": "the algorithmic fp32 rate these launches would reach if EACH sat "
                                    "on its own hardware roofline, max(algorithmic bytes / achievable "
                                    "HBM rate, instruction FLOPs / peak of the pipe it runs on) -- "
                                    "`per_launch_floor` below has the terms; frac = achieved / peak = "
                                    "floor_ms / kernel_ms_per_step.  The same `achieved` against the "
                                    "fp32 MFMA peak (the reference arithmetic's roofline, which the "
                                    "bf16-plane kernels can exceed) is `fp32_mfma_peak`; the hardware "
                                    "FLOPs of the bf16-plane launches against their pipe `bf16_pipe`",
                         "fp32_mfma_peak": {"peak": FP32_MFMA_PEAK_TFLOPS,
                                            "frac": (round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)
                                                     if ok else None)},
                         "bf16_pipe": {"instruction": ("v_mfma_f32_32x32x16_f16" if plane_products() == 3
                                                       else "v_mfma_f32_32x32x16_bf16"),
                                       "hw_flops_per_algorithmic_flop": plane_products(),
                                       "peak": BF16_MFMA_PEAK_TFLOPS,
                                       "launches": sum(v["launches"] for v in bf),
                                       "kernel_ms_per_step": round(bf_ms, 3),
                                       "achieved_algorithmic_tflops": (
                                           round(bf_flop / (bf_ms * 1e-3) / 1e12, 2) if bf_ms else None),
                                       "frac": (round(plane_products() * bf_flop / (bf_ms * 1e-3) / 1e12
                                                      / BF16_MFMA_PEAK_TFLOPS, 4) if bf_ms and ok
                                                else None),
                                       "by_kernel": {k: {"launches": v["launches"],
                                                         "ms": round(v["ms"], 3), "tflops": tf(v)}
                                                     for k, v in paths.items() if k.startswith("bf16")}},
                         "fp32_mfma": {"instruction": "v_mfma_f32_32x32x2_f32",
                                       "peak": FP32_MFMA_PEAK_TFLOPS, "launches": f32["launches"],
                                       "kernel_ms_per_step": round(f32["ms"], 3),
                                       "achieved_tflops": tf(f32),
                                       "frac": (round(tf(f32) / FP32_MFMA_PEAK_TFLOPS, 4)
                                                if f32["ms"] and ok else None)},
                         "flop_check": {"per_launch_geometry_gflop": round(conv["flop"] / 1e9, 2),
                                        "survey_a3_gflop": round(conv_flop / 1e9, 2)},
                         "arithmetic": (
                             "fp32-class: a = a1 + a2 with a1 = fp16(a), a2 = fp16((a - a1) 2^11) / 2^11 "
                             "(11 + 11 mantissa bits and the sign of a2, residual <= 2^-22 |a|), the "
                             "same for the weights; a1 b1 + a1 b2 + a2 b1 accumulated in fp32 at the "
                             "common scale 2^11 (dropped a2 b2 <= 2^-22); against an fp64 convolution "
                             "as close as the six-product bf16 form (half as many fp32 accumulator "
                             "updates; tests/test_kernels_gpu.py::test_conv_p3_matches_fp64_better_"
                             "than_1e_6 holds both to 1e-6 rms); |x| < 65504, |w| < 32; "
                             "VLNCE_CONV_MATH=bf16 selects three bf16 planes / six products, =f32 the "
                             "fp32-MFMA kernel everywhere" if plane_products() == 3 else
                             "fp32-class: operands are represented exactly by three bf16 "
                             "planes (round-to-nearest split), products are exact, six of "
                             "the nine are kept (dropped <= 2^-26 relative), accumulation "
                             "is fp32; measured relative rms error vs an fp64 convolution "
                             "1.6-2.8x torch's own fp32 convolution on the same operands "
                             "(profiles/archive/r03_e_conv_accuracy_*.txt); VLNCE_CONV_MATH=f32 "
                             "selects the fp32-MFMA kernel everywhere"),
                         "per_launch_floor": {
                             "what": "sum over the conv launches of max(algorithmic bytes / "
                                     "achievable HBM rate, instruction FLOPs / the peak of the pipe "
                                     "the launch runs on): the time the launches would take if "
                                     "each sat on its own roofline; hbm_bound = the launches whose "
                                     "HBM term is the larger one",
                             "hbm_achievable_TBps": HBM_ACHIEVABLE_TBS, "hbm_peak_TBps": HBM_PEAK_TBS,
                             "floor_ms": round(conv["floor_ms"], 3),
                             "frac": round(conv["floor_ms"] / conv_ms, 4) if ok else None,
                             "algorithmic_GB": round(conv["bytes"] / 1e9, 3),
                             "hbm_bound": {"launches": conv["hbm_bound_n"],
                                           "ms": round(conv["hbm_bound_ms"], 3),
                                           "floor_ms": round(conv["hbm_floor_ms"], 3),
                                           "achieved_TBps": (round(
                                               conv["hbm_floor_ms"] * HBM_ACHIEVABLE_TBS
                                               / conv["hbm_bound_ms"], 2)
                                               if conv["hbm_bound_ms"] > 0 else None)}},
                         "launches_per_step": n_conv,
                         "avg_launch_ms": round(conv_ms / max(n_conv, 1), 4),
                         "kernel_ms_per_step": round(conv_ms, 3),
                         "eager_single_stream_trunks_ms": round(conv["eager_trunks_ms"], 3),
                         "timing": "HIP events per launch behind a device-side backlog "
                                   "(GPU-paced), empty-pair cost subtracted, min of 3 passes",
                         "invalid_reason": conv["reason"],
                         "traffic": pmc_traffic(n_conv)},
        }
        line.update(exchange_fields(exchange))
        if world == 1 and not args.no_f32_compare and "VLNCE_CONV_MATH" not in os.environ:
            line["config"]["fp32_mfma_only"] = f32_mfma_compare(args)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.num_envs, args.hw, args.tokens)
        final = json.dumps(line)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C runtime's
        # stdout buffer until it is flushed (at exit it would land behind the line)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(final, flush=True)


if __name__ == "__main__":
    main()
