"""Per-step cost of the persistent sequence kernels, launched back to back (HIP events):
the packed instruction LSTM (rnn_seq_fwd / _bwd, H = 128, both directions) and the one-launch
GRU rollout (gru_rollout_fwd / _bwd, H = 512).

    python scripts/seqbench.py [--reps 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vlnce_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = ops.L()
torch.manual_seed(0)


def timed(fn, reps=a.reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


for B, Lm in ((5, 80), (64, 80), (5, 200)):
    H, G = 128, 4
    gis = [torch.randn(Lm, B, G * H, device=dev) * 0.5 for _ in range(2)]
    whh = [torch.randn(G * H, H, device=dev) * H ** -0.5 for _ in range(2)]
    bhh = [torch.randn(G * H, device=dev) * 0.1 for _ in range(2)]
    lengths = torch.full((B,), Lm, dtype=torch.int32, device=dev)
    outs = [torch.zeros(Lm, B, H, device=dev) for _ in range(2)]
    hfin = [torch.empty(B, H, device=dev) for _ in range(2)]
    gates = [torch.empty(Lm, B, G * H, device=dev) for _ in range(2)]
    aux = [torch.empty(Lm, B, H, device=dev) for _ in range(2)]
    us = timed(lambda: lib.rnn_seq_fwd(0, 2, gis, whh, bhh, lengths, outs, hfin, gates, aux, B, Lm, H))
    whh_t = [w.t().contiguous() for w in whh]
    douts = [torch.randn(Lm, B, H, device=dev) for _ in range(2)]
    dgi = [torch.zeros(Lm, B, G * H, device=dev) for _ in range(2)]
    usb = timed(lambda: lib.rnn_seq_bwd(0, 2, whh_t, lengths, outs, gates, aux, douts, None, dgi, None,
                                        B, Lm, H))
    print(f"rnn_seq LSTM H=128 x2 dirs B={B:3d} L={Lm:3d}: fwd {us:8.1f} us = {us / Lm:5.2f} us/step, "
          f"bwd {usb:8.1f} us = {usb / Lm:5.2f} us/step")

for T, N in ((100, 5), (100, 16), (20, 5)):
    H = 512
    GH = 3 * H
    gi = torch.randn(T, N, GH, device=dev) * 0.7
    h0 = torch.randn(N, H, device=dev) * 0.4
    w = torch.randn(GH, H, device=dev) * H ** -0.5
    b = torch.randn(GH, device=dev) * 0.1
    mask = (torch.rand(T, N, device=dev) > 0.1).to(torch.uint8)
    hp, out, aux = (torch.empty(T, N, H, device=dev) for _ in range(3))
    gates = torch.empty(T, N, GH, device=dev)
    ws = torch.empty(lib.gru_rollout_workspace_bytes(N, H), dtype=torch.uint8, device=dev)
    us = timed(lambda: lib.gru_rollout_fwd(gi, h0, mask, w, b, hp, out, gates, aux, ws, T, N, H))
    wt = w.t().contiguous()
    dout = torch.randn(T, N, H, device=dev)
    dgi, dgh = torch.empty(T, N, GH, device=dev), torch.empty(T, N, GH, device=dev)
    dh0 = torch.empty(N, H, device=dev)
    usb = timed(lambda: lib.gru_rollout_bwd(dout, None, gates, aux, hp, mask, wt, dgi, dgh, dh0, ws, T, N, H))

    def steps():
        h = h0
        for t in range(T):
            lib.rnn_step_fwd(False, gi[t], h, None, mask[t], w, b, hp[t], out[t], aux[t], gates[t], N, H)
            h = out[t]
    uss = timed(steps, 5)
    print(f"GRU rollout H=512 T={T:3d} N={N:2d}: fwd {us:8.1f} us = {us / T:5.2f} us/step, "
          f"bwd {usb:8.1f} us = {usb / T:5.2f} us/step; {T} step launches fwd {uss:8.1f} us")
