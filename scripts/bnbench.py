"""Timing of the BatchNorm finalize launch on the (tiles_m, C) pairs of the RGB trunk at num_envs=64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vlnce_amd import ops

dev = "cuda:0"
for tiles, rows, C in ((4096, 64, 64), (2048, 128, 256), (2048, 128, 64), (512, 128, 128),
                       (512, 128, 512), (128, 128, 1024), (256, 64, 256), (64, 64, 2048), (64, 64, 512)):
    M = tiles * rows
    part = torch.rand(tiles, C, 2, device=dev)
    g, b = torch.rand(C, device=dev), torch.rand(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    for _ in range(3):
        ops.bn_finalize((part, tiles, rows), M, g, b, 1e-5, 0.1, rm, rv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.bn_finalize((part, tiles, rows), M, g, b, 1e-5, 0.1, rm, rv)
    e1.record()
    torch.cuda.synchronize()
    print(f"tiles={tiles:5d} C={C:5d}: {e0.elapsed_time(e1) * 20:.1f} us per call (incl. launch + 3 small allocs)")
