"""Side HIP streams for the independent branches of a policy step.

The three encoders of a policy (RGB trunk, depth trunk, instruction RNN) do
not depend on each other.  The RGB trunk is a train of large MFMA kernels;
the depth trunk is ~200 small launches and the packed instruction RNN occupies
only (directions x batch/16) workgroups for ~1 ms -- both leave most of the 256
CUs idle when run alone.  Running them on side streams overlaps them with the
RGB trunk; autograd replays each branch's backward on the stream its forward
used, so the BPTT kernel overlaps the tail's backward as well.
Set VLNCE_SIDE_STREAMS=0 to serialise everything on the current stream."""
import gc
import os
from collections import OrderedDict

import torch


TIMING = [] if os.environ.get("VLNCE_STREAM_TIMING") else None  # (idx, start, end) events
_SIDE_STREAMS = {}  # (idx, device index) -> stream; one set per process, shared by all policies
_READY = {}  # storage address -> completion event of tensors produced ahead on a side stream


class capture_guard:
    """Around every HIP-graph capture: collect cyclic garbage first and keep the collector off
    while the capture is open.  torch.cuda.graph no longer collects on entry, and a cycle that
    dies mid-capture (an earlier policy with its graphs, events and pool memory) runs
    hipGraphDestroy / hipEventDestroy inside the capture, which aborts the process."""

    def __enter__(self):
        gc.collect()
        self.was_enabled = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self.was_enabled:
            gc.enable()
        return False


READY_KEY = "_vlnce_ready"  # observation-dict entry: {feature key: completion event}
_READY_MAX = 64


def mark_ready(t, event):
    """remember that the memory of `t` is complete once `event` has fired (see wait_ready).
    Keyed on the STORAGE address, so any view of `t` (slice, index, permute, reshape) finds the
    event; a stale entry whose storage was freed and re-used only costs a wait on an event that
    has long fired.  Copies made on another stream (`.to(dtype)`, `.clone()`) before wait_ready
    are not covered -- the event also rides in the observation dict (READY_KEY) for consumers
    that rebuild tensors from the dict."""
    _READY[t.untyped_storage().data_ptr()] = event
    while len(_READY) > _READY_MAX:
        _READY.pop(next(iter(_READY)))


def wait_ready(t, observations=None, key=None):
    """orders the current stream after the side-stream producer of `t` (no-op for ordinary
    tensors); every consumer of a tensor handed out by encode_ahead() calls this."""
    ev = _READY.get(t.untyped_storage().data_ptr()) if t.is_cuda else None
    if ev is None and observations is not None:
        ev = (observations.get(READY_KEY) or {}).get(key)
    if ev is not None:
        cur = torch.cuda.current_stream(t.device)
        cur.wait_event(ev)
        t.record_stream(cur)
    return t


def _elapsed_two_spins(a, b, cycles):
    """wall time (ms) of one spin kernel on stream `a` and, if given, one on `b`, started together."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(a)
    if b is not None:
        b.wait_event(e0)
        with torch.cuda.stream(b):
            torch.cuda._sleep(cycles)
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    if b is not None:
        a.wait_stream(b)
    e1.record(a)
    e1.synchronize()
    return e0.elapsed_time(e1)


def pick_concurrent_stream(device, others=(), priority=-1, candidates=12):
    """A stream whose kernels really overlap with the current stream's (and `others`').

    HIP streams are multiplexed onto a handful of hardware queues (4 by default); two streams
    that land on the same queue serialise, and which pool stream lands where depends on how
    many streams the process created before (graph capture, RCCL, other policies).  Instead of
    guessing the runtime's mapping this measures it: a candidate is accepted when two spin
    kernels, one per stream, take the time of one."""
    with torch.cuda.device(device):
        cur = torch.cuda.current_stream(device)
        cycles = 200_000
        _elapsed_two_spins(cur, None, cycles)  # warm-up (module load)
        t1 = min(_elapsed_two_spins(cur, None, cycles) for _ in range(2))
        last = None
        for _ in range(candidates):
            cand = torch.cuda.Stream(device=device, priority=priority)
            _elapsed_two_spins(cur, cand, cycles)  # first use binds the stream to a queue
            ok = all(min(_elapsed_two_spins(ref, cand, cycles) for _ in range(2)) < 1.5 * t1
                     for ref in (cur, *others))
            last = cand
            if ok:
                return cand
        return last  # no concurrency available (single hardware queue): still correct


class BranchStreams:
    # 0: inline branches of a forward (instruction RNN, depth trunk); 1 / 2: the RGB / depth
    # trunks of a batch that runs ahead (encode_visual_ahead).  With the main stream that is
    # one stream per hardware queue of the default HIP runtime configuration.
    NUM_STREAMS = 3

    def __init__(self):
        self._streams = _SIDE_STREAMS

    # set by ActGraph while it captures: the branches then fork / join INSIDE the capture (event
    # record / wait on capturing streams become graph edges), so the one graph of a whole act()
    # keeps the three encoders side by side
    in_capture = False

    @staticmethod
    def enabled(device):
        return (device.type == "cuda" and os.environ.get("VLNCE_SIDE_STREAMS", "1") != "0"
                and (BranchStreams.in_capture or not torch.cuda.is_current_stream_capturing()))

    def _stream(self, idx, device):
        key = (idx, device.index)
        if key not in self._streams:
            # high priority for all three: the branch kernels are small and would otherwise only
            # be admitted at the boundaries of the saturating RGB-trunk kernels.  Measured
            # (steps/s, pipelined | plain loop): "-1,-1,-1" 5888 | 4716, "0,-1,-1" 5951 | 4385,
            # "0,0,0" 5282 | -
            prios = os.environ.get("VLNCE_SIDE_PRIORITY", "-1,-1,-1").split(",")
            prio = int(prios[min(idx, len(prios) - 1)])
            others = [st for (i, d), st in self._streams.items() if d == device.index]
            self._streams[key] = pick_concurrent_stream(device, others, prio)
        return self._streams[key]

    def fork(self, device):
        """Marks the current point of the current stream; branches started with this
        token wait for it (and NOT for work enqueued on the current stream afterwards)."""
        if not self.enabled(device):
            return None
        # both side streams are created (and their hardware-queue placement measured) on the
        # very first use, i.e. before any HIP graph of this process is captured
        for idx in range(self.NUM_STREAMS):
            self._stream(idx, device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        return ev

    def run(self, token, idx, device, fn):
        """fn() on side stream `idx`, ordered after the fork point.  Returns
        (result, join) -- call join() before consuming the result on the current stream."""
        if token is None:
            return fn(), (lambda: None)
        out, done = self.launch(token, idx, device, fn)

        def join():
            # waits for THIS branch only (an event, not the whole side stream: work that was
            # queued behind it, e.g. the next step's trunk, must not hold the consumer back)
            cur = torch.cuda.current_stream(device)
            cur.wait_event(done)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)

        return out, join

    def launch(self, token, idx, device, fn):
        """fn() on side stream `idx` after the fork point; returns (result, completion event)."""
        side = self._stream(idx, device)
        side.wait_event(token)
        timing = TIMING is not None
        with torch.cuda.stream(side):
            if timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(side)
            out = fn()
            done = torch.cuda.Event(enable_timing=timing)
            done.record(side)
        if timing:
            TIMING.append((idx, t0, done))
        return out, done

    def encode_visual_ahead(self, net, observations):
        """Starts the FROZEN visual trunks of `observations` on side streams and returns a copy
        of the dict carrying their outputs as `rgb_features` / `depth_features` -- the same
        bypass keys the reference's DAgger feature cache feeds (resnet_encoders.py:70-72,
        193-195), so the policy consumes them through its normal path.  Nothing downstream of
        the trunks runs here: every trainable module still sees the weights of the step that
        consumes the features.  Stream-ordered after the caller's current stream; the consumer
        waits on the completion events, so the call can be issued a whole update earlier
        (e.g. for batch k+1 before the update on batch k is enqueued) and the trunk then
        overlaps that update's latency-bound tail."""
        out = dict(observations)
        probe = next((v for v in observations.values() if isinstance(v, torch.Tensor)), None)
        if probe is None or not self.enabled(probe.device):
            return out
        dev = probe.device
        fork = None
        plan = (("rgb", "rgb_features", 1, getattr(net, "rgb_encoder", None)),
                ("depth", "depth_features", 2, getattr(net, "depth_encoder", None)))
        for src, key, idx, enc in plan:
            if (enc is None or key in out or src not in out or getattr(enc, "is_blind", False)
                    or not hasattr(enc, "trunk_features")
                    or any(p.requires_grad for p in enc.trunk_parameters())):
                continue
            if not enc.trunk_ready(observations):
                # a new input signature: its eager pass and its graph-capturing pass run right
                # here on the caller's stream (capture starts from an idle device), so trunk
                # passes -- and the BatchNorm running-stat updates -- keep the call order
                out[key] = enc.trunk_features(observations)
                continue
            if fork is None:
                fork = self.fork(dev)
            feats, done = self.launch(fork, idx, dev, lambda e=enc: e.trunk_features(observations))
            mark_ready(feats, done)
            out[key] = feats
            out[READY_KEY] = {**(out.get(READY_KEY) or {}), key: done}
        return out


class GraphedTail:
    """HIP-graph replay of a policy's trainable tail (everything downstream of the three
    encoders): ~100 small launches forward and ~200 backward whose host-side issue cost
    is on the critical path of a step.  Uses torch.cuda.make_graphed_callables, i.e. the
    forward AND the backward of the tail are captured (autograd-aware) per input-shape
    signature; the 1st call with a signature runs eagerly, the 2nd captures.
    Set VLNCE_HIP_GRAPHS=0 to disable."""

    MAX_GRAPHS = 8
    CAPTURE_AFTER = 2  # sightings of a signature before it is captured (1 eager pass first)

    def __init__(self, make_module):
        # make_graphed_callables patches the module's forward in place, so every captured
        # signature gets its own (cheap: it only references the shared sub-modules) instance
        self.make_module = make_module
        self.modules = {}             # static configuration -> eager instance
        self.entries = OrderedDict()  # signature -> graphed callable; LRU over CAPTURED graphs
        self.sightings = OrderedDict()  # signature -> times seen before capture (bounded, no graphs)

    @property
    def module(self):
        return self._eager(())

    def _eager(self, static):
        m = self.modules.get(static)
        if m is None:
            m = self.modules[static] = self.make_module(*static)
        return m

    def __call__(self, *tensors, static=()):
        """`static`: hashable configuration handed to make_module(*static) -- part of the graph key
        (values the tail's Python control flow depends on that are not visible in the tensors)."""
        t0 = tensors[0]
        if (not t0.is_cuda or os.environ.get("VLNCE_HIP_GRAPHS", "1") == "0"
                or torch.cuda.is_current_stream_capturing()):
            return self._eager(static)(*tensors)
        key = tuple((tuple(t.shape), t.dtype, t.requires_grad) for t in tensors) + (
            torch.is_grad_enabled(), static)
        ent = self.entries.get(key)
        if ent is None:
            # first sightings are counted apart from the captured graphs: a stream of new
            # signatures (e.g. the waypoint tail's batch-dependent Lmax) must not evict graphs
            seen = self.sightings.pop(key, 0) + 1
            if seen < self.CAPTURE_AFTER:
                while len(self.sightings) >= 8 * self.MAX_GRAPHS:
                    self.sightings.popitem(last=False)
                self.sightings[key] = seen
                return self._eager(static)(*tensors)
            sample = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in tensors)
            with capture_guard():
                ent = torch.cuda.make_graphed_callables(self.make_module(*static), sample,
                                                        allow_unused_input=True)
            while len(self.entries) >= self.MAX_GRAPHS:
                self.entries.popitem(last=False)  # least recently used captured graph
            self.entries[key] = ent
        self.entries.move_to_end(key)
        return ent(*tensors)


class ActGraph:
    """The WHOLE forward-only act() of an IL policy (three encoders on forked streams, tail,
    action head, mode / sample) as ONE HIP graph per input signature, for the inference and
    evaluation loops that call it with a handful of environments (base_il_trainer.py:284-331,
    dagger_trainer.py:183-193): there a call is ~300 launches and three host syncs (instruction
    lengths, distribution validation, the caller's own), i.e. host time.

    Protocol per key: 1st call eager, 2nd call captures, later calls copy the inputs into the
    graph's static buffers and replay.  The key carries the input signature, the sampling mode and
    every parameter version / normalisation state of the policy, because tensors DERIVED from
    parameters outside the capture (folded BatchNorm vectors, packed convolution weights) are
    baked into the graph: an optimizer step or load_state_dict makes a new key.
    Inside the capture the instruction runs at its static padded length (lengths stay on the
    device; steps past a row's length emit zeros, which the text attention masks exactly) and the
    empty-instruction check of the eager path (one host sync) is not made; the action
    distribution is built without argument validation (another sync).
    Used under no_grad, in eval mode, for at most MAX_ENVS rows, and only when VLNCE_ACT_GRAPH=1:
    measured on MI355X (profiles/archive/r03_l_*) the call is bound by the GPU, not the host -- the bare
    replay of the captured graph is 1.37 ms at 1 environment against 1.52 ms for the eager call
    (three smaller graphs + an eager instruction encoder), and with the input / output copies
    the graphed call is 1.61 / 1.84 / 2.40 ms at 1 / 4 / 8 environments against 1.52 / 1.90 /
    2.47 -- so it is off by default and there for hosts that are busy with a simulator."""

    MAX_ENVS = 16
    MAX_GRAPHS = 6

    def __init__(self, policy):
        self.policy = policy
        self.entries = OrderedDict()
        self._tracked = None  # (parameters + buffers, modules with a _graph_key): walked once

    def _hooked(self):
        """forward hooks on any module of the policy (DAgger's feature capture registers them on
        `.net.rgb_encoder.cnn` / `.net.depth_encoder.visual_encoder`, dagger_trainer.py:300-314, and
        reads `o.cpu()` inside): a hook runs when Python calls the module, i.e. once, at capture --
        and a device-to-host copy inside a capture is an error.  With hooks the call stays eager."""
        import torch.nn.modules.module as _m
        if _m._global_forward_hooks or _m._global_forward_pre_hooks:
            return True
        mods = self.__dict__.get("_modules_list")
        if mods is None:
            mods = self._modules_list = list(self.policy.modules())
        return any(m._forward_hooks or m._forward_pre_hooks for m in mods)

    def usable(self, observations, rnn_states):
        return (rnn_states.is_cuda and rnn_states.size(0) <= self.MAX_ENVS
                and not self._hooked()
                and not torch.is_grad_enabled() and not self.policy.training
                and os.environ.get("VLNCE_ACT_GRAPH", "0") == "1"
                and os.environ.get("VLNCE_HIP_GRAPHS", "1") != "0"
                and not torch.cuda.is_current_stream_capturing()
                and READY_KEY not in observations
                and all(isinstance(v, torch.Tensor) and v.is_cuda for v in observations.values()))

    def _key(self, observations, rnn_states, prev_actions, masks, deterministic):
        pol = self.policy
        sig = tuple(sorted((k, tuple(v.shape), v.dtype, v.is_contiguous())
                           for k, v in observations.items()))
        if self._tracked is None:
            self._tracked = ([t for t in list(pol.parameters()) + list(pol.buffers()) if t is not None],
                             [m for m in pol.modules() if hasattr(m, "_graph_key")])
        state = tuple(t._version for t in self._tracked[0])
        try:
            trunks = tuple(m._graph_key(None) for m in self._tracked[1])
        except TypeError:  # a trunk that has not run yet (its input transform is set by its first call)
            return None
        return (sig, tuple(rnn_states.shape), tuple(prev_actions.shape), prev_actions.dtype,
                tuple(masks.shape), masks.dtype, bool(deterministic), state, trunks)

    def __call__(self, observations, rnn_states, prev_actions, masks, deterministic):
        key = self._key(observations, rnn_states, prev_actions, masks, deterministic)
        if key is None:
            return self.policy._act_eager(observations, rnn_states, prev_actions, masks, deterministic)
        ent = self.entries.get(key)
        if ent is None:
            while len(self.entries) >= self.MAX_GRAPHS:
                self.entries.popitem(last=False)
            self.entries[key] = "seen"
            return self.policy._act_eager(observations, rnn_states, prev_actions, masks, deterministic)
        self.entries.move_to_end(key)
        names = sorted(observations)
        if ent == "seen":
            static = ({k: observations[k].clone() for k in names}, rnn_states.clone(),
                      prev_actions.clone(), masks.clone())
            graph = torch.cuda.CUDAGraph()
            BranchStreams.in_capture = True
            try:
                with capture_guard(), torch.cuda.graph(graph):
                    out = self.policy._act_eager(*static, deterministic)
            finally:
                BranchStreams.in_capture = False
            ent = [graph, static, out]
            self.entries[key] = ent
        else:
            sobs, sstate, sprev, smask = ent[1]
            for k in names:
                sobs[k].copy_(observations[k])
            sstate.copy_(rnn_states)
            sprev.copy_(prev_actions)
            smask.copy_(masks)
        ent[0].replay()
        return ent[2][0].clone(), ent[2][1].clone()


class DropsGraphsOnApply:
    """Mixin (in front of nn.Module in the bases) for modules that hold captured HIP graphs in
    `_graphs` / `_tail` / `_act_graph`.  nn.Module._apply -- .to(), .cuda(), .float(), .half() --
    gives the parameters new storage WITHOUT bumping their version counters, which are what the
    graph keys carry; a graph captured before the move would replay on the old pointers.  The
    captured graphs are dropped instead; the next calls run eagerly and capture again."""

    _graph_holders = ("_graphs", "_tail", "_act_graph")

    def _plist(self, checked=True):
        """the module's parameters as a list built once: the trunks' graph keys read every
        parameter's version on every call, and walking the module tree for that
        (`self.parameters()`: ~0.1 ms for a ResNet-50 trunk, twice per call) was host time at the
        very start of a step, with the GPU idle (profiles/r05_h_*).  The trunks' module structure
        is fixed after construction; _apply drops the list with the graphs."""
        pl = self.__dict__.get("_param_list")
        if pl is not None and checked:   # (checked=False: the caller validated it in this call already)
            #  a Parameter REBOUND behind the module's back -- load_state_dict(assign=True),
            # `conv.weight = nn.Parameter(...)`, a child-only .to() under
            # torch.__future__.set_overwrite_module_params_on_conversion -- leaves the list pointing at
            # the old tensors: the graph keys would keep reading their versions and the graphs keep
            # replaying on their storage.  One dict lookup per parameter (~10 us per trunk) instead of
            # the module-tree walk.  (An in-place storage swap of a CHILD module, `trunk[4].to(...)`,
            # keeps identity and version: move the policy as a whole, or call .to() on the trunk.)
            for owner, name, par in self.__dict__["_param_slots"]:
                if owner._parameters.get(name) is not par:
                    pl = None
                    self._drop_graphs()
                    break
        if pl is None:
            slots = [(m, n, p) for m in self.modules() for n, p in m._parameters.items() if p is not None]
            seen, pl = set(), []
            for _, _, p in slots:   # (order and de-duplication of nn.Module.parameters())
                if id(p) not in seen:
                    seen.add(id(p))
                    pl.append(p)
            object.__setattr__(self, "_param_slots", slots)
            object.__setattr__(self, "_param_list", pl)
        return pl

    def _drop_graphs(self):
        self.__dict__.pop("_param_list", None)
        self.__dict__.pop("_param_slots", None)
        for name in self._graph_holders:
            holder = self.__dict__.get(name)
            if holder is not None:
                holder.entries.clear()
                if hasattr(holder, "sightings"):
                    holder.sightings.clear()
                if hasattr(holder, "_tracked"):
                    holder._tracked = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._drop_graphs()
        return out


def bucket_rows(n, step=8):
    """smallest multiple of `step` >= n: instruction lengths are padded to buckets before a
    graphed tail so that batches whose longest instruction differs by a few tokens share one
    captured graph (the pad rows are all-zero, which the text attention masks out exactly)."""
    return ((int(n) + step - 1) // step) * step
