"""Time the instruction encoder (packed bidirectional LSTM, hidden 128) forward and forward+backward.
    python scripts/rnnbench.py [--b 64] [--l 80]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vlnce_amd

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=64)
ap.add_argument("--l", type=int, default=80)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
pol = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(256, 256)).to(dev)
enc = pol.net.instruction_encoder
tok = torch.randint(1, 2000, (a.b, a.l), device=dev)
obs = {"instruction": tok}
for mode in ("fwd", "fwd+bwd"):
    for _ in range(3):
        y = enc(obs)
        if mode != "fwd":
            y.sum().backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = enc(obs)
        if mode != "fwd":
            y.sum().backward()
    e1.record()
    torch.cuda.synchronize()
    print(f"instruction encoder B={a.b} L={a.l} {mode}: {e0.elapsed_time(e1) / 20:.3f} ms")
