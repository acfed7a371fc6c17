"""Shows the stream -> hardware-queue aliasing that pick_concurrent_stream() works around:
for the first 10 pool streams, the time of two simultaneous spin kernels (current stream +
pool stream) relative to one.  ~1.0 = concurrent, ~2.0 = same hardware queue."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vlnce_amd import streams

dev = torch.device("cuda", 0)
cur = torch.cuda.current_stream(dev)
streams._elapsed_two_spins(cur, None, 200_000)
t1 = min(streams._elapsed_two_spins(cur, None, 200_000) for _ in range(3))
print(f"one spin kernel: {t1*1e3:.0f} us")
for i in range(10):
    st = torch.cuda.Stream(device=dev, priority=-1)
    streams._elapsed_two_spins(cur, st, 200_000)
    t2 = min(streams._elapsed_two_spins(cur, st, 200_000) for _ in range(3))
    print(f"pool stream {i}: ratio {t2/t1:.2f}")
picked = streams.pick_concurrent_stream(dev)
t2 = min(streams._elapsed_two_spins(cur, picked, 200_000) for _ in range(3))
print(f"picked: ratio {t2/t1:.2f}")
