"""Side HIP streams for the independent branches of a policy step.

The three encoders of a policy (RGB trunk, depth trunk, instruction RNN) do
not depend on each other.  The RGB trunk is a train of large MFMA kernels;
the depth trunk is ~200 small launches and the packed instruction RNN occupies
only (directions x batch/16) workgroups for ~1 ms -- both leave most of the 256
CUs idle when run alone.  Running them on side streams overlaps them with the
RGB trunk; autograd replays each branch's backward on the stream its forward
used, so the BPTT kernel overlaps the tail's backward as well.
Set VLNCE_SIDE_STREAMS=0 to serialise everything on the current stream."""
import os

import torch


_SIDE_STREAMS = {}  # (idx, device index) -> stream; one set per process, shared by all policies


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
    def __init__(self):
        self._streams = _SIDE_STREAMS

    @staticmethod
    def enabled(device):
        return (device.type == "cuda" and os.environ.get("VLNCE_SIDE_STREAMS", "1") != "0"
                and not torch.cuda.is_current_stream_capturing())

    def _stream(self, idx, device):
        key = (idx, device.index)
        if key not in self._streams:
            # high priority: the branch kernels are small; without it the hardware only
            # admits them at the boundaries of the saturating RGB-trunk kernels
            prio = int(os.environ.get("VLNCE_SIDE_PRIORITY", "-1"))
            others = [st for (i, d), st in self._streams.items() if d == device.index]
            self._streams[key] = pick_concurrent_stream(device, others, prio)
        return self._streams[key]

    def fork(self, device):
        """Marks the current point of the current stream; branches started with this
        token wait for it (and NOT for work enqueued on the current stream afterwards)."""
        if not self.enabled(device):
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        return ev

    def run(self, token, idx, device, fn):
        """fn() on side stream `idx`, ordered after the fork point.  Returns
        (result, join) -- call join() before consuming the result on the current stream."""
        if token is None:
            return fn(), (lambda: None)
        cur = torch.cuda.current_stream(device)
        side = self._stream(idx, device)
        side.wait_event(token)
        with torch.cuda.stream(side):
            out = fn()

        def join():
            cur.wait_stream(side)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)

        return out, join


class GraphedTail:
    """HIP-graph replay of a policy's trainable tail (everything downstream of the three
    encoders): ~100 small launches forward and ~200 backward whose host-side issue cost
    is on the critical path of a step.  Uses torch.cuda.make_graphed_callables, i.e. the
    forward AND the backward of the tail are captured (autograd-aware) per input-shape
    signature; the 1st call with a signature runs eagerly, the 2nd captures.
    Set VLNCE_HIP_GRAPHS=0 to disable."""

    MAX_GRAPHS = 8

    def __init__(self, make_module):
        # make_graphed_callables patches the module's forward in place, so every captured
        # signature gets its own (cheap: it only references the shared sub-modules) instance
        self.make_module = make_module
        self.module = make_module()
        self.entries = {}

    def __call__(self, *tensors):
        t0 = tensors[0]
        if (not t0.is_cuda or os.environ.get("VLNCE_HIP_GRAPHS", "1") == "0"
                or torch.cuda.is_current_stream_capturing()):
            return self.module(*tensors)
        key = tuple((tuple(t.shape), t.dtype, t.requires_grad) for t in tensors) + (
            torch.is_grad_enabled(),)
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) >= self.MAX_GRAPHS:
                self.entries.pop(next(iter(self.entries)))
            self.entries[key] = "seen"
            return self.module(*tensors)
        if ent == "seen":
            sample = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in tensors)
            ent = torch.cuda.make_graphed_callables(self.make_module(), sample,
                                                    allow_unused_input=True)
            self.entries[key] = ent
        return ent(*tensors)
