"""Test driver (tests/test_bench_launcher.py): bench.py's HOST logic -- self-launch, rank set-up,
sharded update through GradientAllReducer, max-over-ranks timing, the one JSON line -- on CPU, with
tests/hostsim.py (a tensor-level simulator of the C ABI binding) behind the package and gloo behind
torch.distributed.  bench.py re-runs THIS script as its ranks (sys.argv[0]).  Never a measurement:
the line says so, carries value = None and no roofline."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import hostsim  # noqa: E402
from vlnce_amd import _lib  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    _lib._LIB = hostsim.HostSim()
    bench.SIMULATED_BACKEND = "hostsim: CPU simulator of the C ABI (tests/hostsim.py)"
    bench.main()
