"""The RGB trunk and the depth trunk on DISJOINT CUs (CU-masked streams) against today's shared
CUs: GPU event stamps, graph replay, num_envs 64.

    python scripts/cumask_overlap_probe.py

Needs a library build with the (since removed) dispatch option "cus" -- the persistent kernels'
grid size for a masked stream; kept for the record of the experiment
(profiles/archive/r04_zr_cu_masked_streams_trunks_on_disjoint_cus.txt: the RGB trunk alone takes 8.0 ms on
224 CUs and 11.0 ms on 240 against 6.0 ms on all 256 -- a masked queue does not place one
workgroup per CU --, so partitioning the CUs between the trunks loses to sharing them, 6.7 ms).
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vlnce_amd  # noqa: E402
from vlnce_amd import ops  # noqa: E402
from vlnce_amd.streams import pick_concurrent_stream  # noqa: E402

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
KEEP = []


def masked_stream(lo, hi):
    """stream on mask bits [lo, hi): bit i = XCD i % 8, CU slot i // 8 (scripts/cumask_probe.py)"""
    words = (NCU + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask) == 0
    KEEP.append(s)
    return torch.cuda.ExternalStream(s.value, device=dev)


def ev():
    return torch.cuda.Event(enable_timing=True)


obs = bench.synth_batch(64, 256, 80, dev, seed=1)[0]
lib = ops.L()


def run(name, rgb_stream, dep_stream, rgb_cus, dep_cus):
    torch.manual_seed(0)
    policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
    net = policy.net

    def rgb():
        with torch.cuda.stream(rgb_stream), lib.options(cus=rgb_cus):
            net.rgb_encoder(obs)

    def dep():
        with torch.cuda.stream(dep_stream), lib.options(cus=dep_cus):
            net.depth_encoder(obs)

    with torch.no_grad():
        for _ in range(3):   # eager, capture, replay
            rgb()
            dep()
            torch.cuda.synchronize()
        res = []
        for which in ("rgb alone", "depth alone", "both"):
            best = None
            for rep in range(3):
                torch.cuda.synchronize()
                e0, a1, b1 = ev(), ev(), ev()
                main = torch.cuda.current_stream(dev)
                e0.record(main)
                rgb_stream.wait_event(e0)
                dep_stream.wait_event(e0)
                if which != "depth alone":
                    rgb()
                a1.record(rgb_stream)
                if which != "rgb alone":
                    dep()
                b1.record(dep_stream)
                torch.cuda.synchronize()
                t = (e0.elapsed_time(a1), e0.elapsed_time(b1))
                best = t if best is None or max(t) < max(best) else best
            res.append(f"{which}: rgb done {best[0]:.2f} depth done {best[1]:.2f}")
    print(f"{name:44s} | " + " | ".join(res), flush=True)


main = torch.cuda.current_stream(dev)
run("shared: main stream + side stream, 256 CUs", main, pick_concurrent_stream(dev), 0, 0)
for n_dep in (16, 32, 48, 64):
    run(f"disjoint: RGB {NCU - n_dep} CUs, depth {n_dep} CUs", masked_stream(0, NCU - n_dep),
        masked_stream(NCU - n_dep, NCU), NCU - n_dep, n_dep)
run("masked RGB 256 (all bits) + unmasked side", masked_stream(0, NCU), pick_concurrent_stream(dev), 0, 0)
