"""Data-parallel gradient exchange for the policy update.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The env / episode dimension N is sharded
across ranks, weights are replicated, the forward needs no communication and
the only exchange step is a sum-all-reduce (then / world) of the gradients of
the TRAINABLE tensors (25.1 MB for CMA with frozen encoders).

Buckets are filled in reverse parameter order -- the order backward produces
gradients -- and each bucket's (coalesced, in-place) all-reduce is launched
asynchronously from a post-accumulate-grad hook the moment its last gradient
lands, so it overlaps with the rest of backward.  xGMI is point-to-point (7 links x ~153 GB/s per
GPU): few, large collectives beat many small ones, hence 8 MiB buckets rather
than DDP's NVSwitch-era 25 MB first/1 MB rest heuristics being copied.
Parameters that never receive a gradient (WaypointPolicy's unused
action_distribution, ddppo_waypoint_trainer.py:370 `find_unused_params=True`)
contribute zeros on every rank, so ranks stay in lock-step.

Semantics preserved (base_il_trainer.py:159-165): the IL loss is normalised
per episode and then .mean()'d over episodes, so equal-sized shards + gradient
averaging reproduce the single-process gradient exactly.
"""
import time
import warnings

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "pending", "work", "tensors", "flags")


class GradientAllReducer:
    """Zero-copy bucketed gradient averaging: each bucket is ONE coalesced collective over the
    .grad tensors themselves (ncclGroupStart/End underneath, no flatten / unflatten copies and,
    on RCCL, ReduceOp.AVG so there is no scaling pass either)."""

    def __init__(self, module, bucket_bytes=8 << 20, process_group=None, divergent_unused=False,
                 timing=False):
        """timing=True: every bucket's collective is bracketed by events (host clock without a
        communication stream), finish() stamps the end of backward, and stats() reports how much
        of the exchange ran under backward (bench.py: `allreduce_ms`, `allreduce_hidden_frac`)."""
        self.timing = bool(timing)
        self._stamps = []   # per bucket of the last step: (start, end) events or host times
        self._bwd_end = None
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        # gloo (CPU tests) has no AVG: sum, then one fused multiply
        self.avg = dist.get_backend(process_group) == "nccl"
        params = [p for p in module.parameters() if p.requires_grad]
        # RCCL's kernels go on a stream of OURS: c10d's own communication stream is a pool
        # stream whose hardware queue we do not control -- if it shares the queue of the stream
        # the next batch's RGB trunk runs on (streams.BranchStreams), every bucket would wait
        # for a whole trunk.  Collectives issued with async_op=False run on the current stream
        # (ProcessGroupNCCL), so they are issued under side stream 0, which is measured to be
        # concurrent with the main stream and which the trunk streams are measured against.
        self.comm = None
        if self.avg and params and params[0].is_cuda:
            from .streams import BranchStreams

            self.comm = BranchStreams()._stream(0, params[0].device)
        params.reverse()
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in params:
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(self._make_bucket(cur))
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self._bucket_of = {}
        self._next = 0  # index of the first bucket whose collective has not been issued yet
        # rank-divergent parameter use: every rank must put the SAME tensors into a bucket's
        # coalesced collective, so the used-flag vector is sent by all ranks for every bucket
        self._divergent = bool(divergent_unused)
        self._handles = []
        self._zeros = {}
        # parameters that received no gradient in the previous step.  With a static graph (the
        # default: the unused set is the same on every rank, e.g. WaypointPolicy's
        # action_distribution under WDDPPO) they are not waited for again: their bucket's pending
        # count starts without them, so strict bucket order no longer serialises every
        # collective into finish() because bucket 0 holds an unused head (round-3 ADVICE).  One
        # that does receive a gradient after its bucket went out is reduced in finish() (_late).
        self._skip = frozenset()
        self._late = []
        self.launched_before_finish = 0  # buckets of the last step issued from the hooks (tests)
        # the hooks pin every AccumulateGrad node to the stream current here, which is what
        # orders gradients produced on side streams before the collective; the engine's
        # warning about that (intended) stream hand-over is noise
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        for b in self.buckets:
            for p in b.params:
                self._bucket_of[p] = b
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    @staticmethod
    def _make_bucket(params):
        b = _Bucket()
        b.params = list(params)
        b.pending = len(params)
        b.work = None
        b.tensors = None
        b.flags = None
        return b

    def _zero_like(self, p):
        z = self._zeros.get(p)
        if z is None:
            z = self._zeros[p] = torch.zeros_like(p)
        else:
            z.zero_()
        return z

    def _launch(self, b):
        # a parameter without a gradient this step contributes zeros (and keeps .grad None)
        b.tensors = [p.grad if p.grad is not None else self._zero_like(p) for p in b.params]
        b.flags = None
        if self._divergent:
            # which parameters received a gradient on SOME rank: a parameter this rank did not use
            # but another one did must end up with the averaged gradient here too.  The flag
            # vector rides in the same coalesced collective and is sent by EVERY rank for every
            # bucket (all ranks must put the same tensors into a collective).  Without
            # divergent_unused the set of unused parameters is the same on all ranks (e.g.
            # WaypointPolicy.action_distribution.*): they contribute zeros and keep .grad None.
            b.flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in b.params],
                                   device=b.tensors[0].device)
            b.tensors = b.tensors + [b.flags]
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if self.comm is None:
                if self.timing:
                    self._stamps.append([time.perf_counter(), None, b])
                b.work = dist.all_reduce_coalesced(b.tensors, op=op, group=self.group,
                                                   async_op=True)
                return
            self.comm.wait_stream(torch.cuda.current_stream(self.comm.device))
            with torch.cuda.stream(self.comm):
                if self.timing:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(self.comm)
                dist.all_reduce_coalesced(b.tensors, op=op, group=self.group, async_op=False)
                b.work = torch.cuda.Event(enable_timing=self.timing)
                b.work.record(self.comm)
                if self.timing:
                    self._stamps.append([t0, b.work, b])

    def _on_grad(self, p):
        b = self._bucket_of[p]
        if p in self._skip:
            # counted as unused when the step was armed.  Not launched yet: its gradient simply
            # rides in the bucket.  Already launched (with zeros in its place): reduce it in
            # finish() -- in static mode every rank sees the same thing and does the same.
            if b.work is not None:
                self._late.append(p)
            return
        b.pending -= 1
        # Collectives are issued STRICTLY in bucket order: a complete bucket waits for every bucket
        # before it.  If the set of parameters that receive a gradient ever differs between ranks
        # (data-dependent branches, unused heads), a bucket one rank can only launch in finish()
        # must not be overtaken by later buckets there while another rank launches it early --
        # ranks issuing collectives in different orders hang.
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """Call after loss.backward(): waits for every bucket (the gradients were averaged in
        place) and re-arms the hooks for the next step."""
        self.launched_before_finish = self._next
        if self.timing:
            if self.comm is None:
                self._bwd_end = time.perf_counter()
            else:
                self._bwd_end = torch.cuda.Event(enable_timing=True)
                self._bwd_end.record(torch.cuda.current_stream(self.comm.device))
            self._done_stamps = self._stamps   # this step's buckets (those launched below join it)
            self._stamps = []
            stamps_of_step = self._done_stamps
        for b in self.buckets[self._next:]:  # held back by a parameter without a gradient
            self._launch(b)
        self._next = 0
        if self._late:
            late = [p.grad for p in self._late]
            if self.comm is not None:
                torch.cuda.current_stream(self.comm.device).wait_stream(self.comm)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                dist.all_reduce_coalesced(late, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM,
                                          group=self.group)
            if not self.avg:
                torch._foreach_mul_(late, 1.0 / self.world)
            self._late = []
        if self.timing:
            stamps_of_step.extend(self._stamps)
            self._stamps = []
        for b in self.buckets:
            if self.comm is None:
                b.work.wait()
                if self.timing:
                    for st in stamps_of_step:
                        if st[2] is b and st[1] is None:
                            st[1] = time.perf_counter()
            else:
                torch.cuda.current_stream(self.comm.device).wait_event(b.work)
            if not self.avg:
                have = [t for t, p in zip(b.tensors, b.params) if p.grad is not None]
                if have:
                    torch._foreach_mul_(have, 1.0 / self.world)
            if b.flags is not None:
                used = b.flags.tolist()  # (host sync: only on the unused-parameter path)
                for t, p, u in zip(b.tensors, b.params, used):
                    if p.grad is None and u > 0:   # used elsewhere: adopt the averaged gradient
                        p.grad = t.clone() if self.avg else t * (1.0 / self.world)
            b.work = None
            b.tensors = None
            b.flags = None
        # re-arm: in static mode the parameters left without a gradient are not waited for again
        if not self._divergent:
            self._skip = frozenset(p for b in self.buckets for p in b.params if p.grad is None)
        for b in self.buckets:
            b.pending = sum(1 for p in b.params if p not in self._skip)

    def stats(self):
        """Exchange of the LAST finished step (call after a device synchronisation): total time of
        the bucket collectives, the part of it that ran after backward had ended (`exposed_ms`:
        from the backward-end stamp finish() takes to the end of the last collective), and
        hidden_frac = 1 - exposed / total.  None without timing=True or before the first step."""
        st = getattr(self, "_done_stamps", None)
        if not self.timing or not st or self._bwd_end is None:
            return None
        if self.comm is None:
            total = sum(e - s for s, e, _ in st) * 1e3
            exposed = max(0.0, max(e for _, e, _ in st) - self._bwd_end) * 1e3
        else:
            total = sum(s.elapsed_time(e) for s, e, _ in st)
            exposed = max(0.0, max(self._bwd_end.elapsed_time(e) for _, e, _ in st))
        exposed = min(exposed, total)
        return {"allreduce_ms": total, "exposed_ms": exposed, "buckets": len(st),
                "issued_from_hooks": self.launched_before_finish,
                "hidden_frac": (1.0 - exposed / total) if total > 0 else None}

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def shard_rows(n_total, rank, world):
    """Contiguous, equal-sized env shard of rank `rank` (N must divide evenly so the
    per-episode-normalised IL loss averages exactly; see module docstring)."""
    assert n_total % world == 0, "num_envs must be divisible by the number of ranks"
    per = n_total // world
    return slice(rank * per, (rank + 1) * per)


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_bdf(device_index):
    p = torch.cuda.get_device_properties(device_index)
    try:
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except AttributeError:
        return None


def _read(path):
    with open(path) as f:
        return f.read().strip()


def gpu_numa_node(device_index=0, sysfs="/sys"):
    """NUMA node of the host socket GPU `device_index` hangs off (PCI sysfs), or None when the
    platform does not say (single-socket hosts report -1)."""
    import os

    bdf = _gpu_bdf(device_index)
    if bdf is None:
        return None
    try:
        node = int(_read(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")))
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def _l3_domains(cpus, sysfs):
    """The L3 domains (CCDs on EPYC) the CPU set `cpus` spans, each cut down to `cpus`, ordered by
    their lowest CPU; [] when the cache topology is not readable."""
    import os

    doms = {}
    for c in sorted(cpus):
        if any(c in d for d in doms.values()):
            continue
        try:
            d = _parse_cpulist(_read(os.path.join(
                sysfs, "devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % c))) & cpus
        except (OSError, ValueError):
            return []
        if d:
            doms[min(d)] = d
    return [doms[k] for k in sorted(doms)]


def _gpu_slot_on_node(bdf, node, sysfs):
    """(k, n): this GPU is the k-th (PCI order) of the n AMD GPUs on NUMA node `node`.  sysfs
    shows the host's PCI devices to every container, so ranks of DIFFERENT jobs on one host get
    different slots too."""
    import os

    root = os.path.join(sysfs, "bus/pci/devices")
    gpus = []
    try:
        names = sorted(os.listdir(root))
    except OSError:
        names = []
    for name in names:
        try:
            if (_read(os.path.join(root, name, "vendor")) == "0x1002"
                    and _read(os.path.join(root, name, "class"))[:4] in ("0x03", "0x12")
                    and int(_read(os.path.join(root, name, "numa_node"))) == node):
                gpus.append(name)
        except (OSError, ValueError):
            continue
    return (gpus.index(bdf), len(gpus)) if bdf in gpus else (0, 1)


_AFFINITY_BEFORE_BIND = None   # the mask this process had before the first bind (inherited by forks)


def restore_worker_affinity():
    """For the init function of worker processes started after bind_host_threads_to_gpu_socket()
    (`DataLoader(worker_init_fn=...)`, habitat VectorEnv workers): give the calling process the
    CPU mask the trainer had BEFORE it bound its own threads.  Returns the mask, or None when the
    parent never bound (or under spawn, where the module state is not inherited: nothing to undo
    there either -- a spawned child starts from the parent's mask at exec time, so bind after
    creating the workers in that case)."""
    import os

    if _AFFINITY_BEFORE_BIND is None:
        return None
    try:
        os.sched_setaffinity(0, _AFFINITY_BEFORE_BIND)
    except OSError:
        return None
    return set(_AFFINITY_BEFORE_BIND)


def bind_host_threads_to_gpu_socket(device_index=0, node=None, sysfs="/sys", scope=None):
    """One process per GPU: keep this rank's threads (the issuing thread, autograd's, the HIP
    runtime's) on the CPU socket its GPU is attached to -- with scope "l3" on ONE L3 domain of it.
    Half of a step's phases are paced by the issuing thread and its hand-overs to autograd's
    thread; on the two-socket hosts of the MI355X nodes the scheduler
    otherwise places those threads on either socket, measured as two speeds of the same loop
    (profiles/r05_zz_step_jitter*.txt: 9.4-9.7 ms unbound in the slow mode, 9.0-9.2 on one socket,
    8.94-8.98 on one L3 domain).

    scope (default: VLNCE_BIND_SOCKET, else "socket"): "socket" / "1" = every CPU of the socket;
    "l3" = one L3 domain (a CCD: 8 cores + their SMT siblings) of the GPU's socket, the GPUs of a
    socket spread evenly over its domains by PCI order; "0" = do nothing.  The library default is
    the whole socket: processes forked or spawned AFTERWARDS -- habitat's VectorEnv
    workers, DataLoader workers -- inherit the mask, and a trainer that builds 32-64 simulator
    workers must not pin them to eight cores; `bench.py` (no workers) asks for "l3" itself, and
    `restore_worker_affinity()` in a worker's init function undoes the inheritance.  Returns the
    NUMA node bound to, or None when nothing was changed (unknown topology, or an affinity mask
    that already excludes the node)."""
    import os

    scope = (scope or os.environ.get("VLNCE_BIND_SOCKET") or "socket").lower()
    if scope == "0":
        return None
    if node is None:
        node = gpu_numa_node(device_index, sysfs)
    if node is None:
        return None
    try:
        cpus = _parse_cpulist(_read(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)))
    except (OSError, ValueError):
        return None
    cpus &= os.sched_getaffinity(0)
    if not cpus:
        return None
    if scope == "l3":
        doms = _l3_domains(cpus, sysfs)
        if doms:
            k, n = _gpu_slot_on_node(_gpu_bdf(device_index), node, sysfs)
            cpus = doms[(k * len(doms)) // max(n, 1) % len(doms)]
    global _AFFINITY_BEFORE_BIND
    if _AFFINITY_BEFORE_BIND is None:
        _AFFINITY_BEFORE_BIND = os.sched_getaffinity(0)
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:
            pass  # a thread that exited between the listing and the call
    return node
