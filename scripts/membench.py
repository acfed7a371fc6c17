"""HBM micro-benchmarks that bound the store-heavy conv layers: write-only, read-only, copy and a
row-strided write (512 B of every 1 KiB row = what one N-tile of a 256-channel output writes).
    python scripts/membench.py"""
import torch

dev = "cuda:0"


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it  # us


for mb in (67, 268, 1072):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    z = torch.empty(1, device=dev)
    tw = timeit(lambda: y.fill_(1.0))
    tz = timeit(lambda: y.zero_())
    tc = timeit(lambda: y.copy_(x))
    tr = timeit(lambda: torch.sum(x, dim=0, out=z[0]))
    ta = timeit(lambda: torch.add(x, 1.0, out=y))
    b = n * 4 / 1e6  # MB
    print(f"{mb:5d} MiB: fill {b/tw:.2f} TB/s ({tw:.1f} us) | zero {b/tz:.2f} | copy {2*b/tc:.2f} TB/s "
          f"total ({tc:.1f} us) | read(sum) {b/tr:.2f} | add-scalar r+w {2*b/ta:.2f}")
    y2 = y[: (n // 256) * 256].view(-1, 256)
    ts = timeit(lambda: y2[:, :128].fill_(1.0))
    print(f"            strided fill of 512 B per 1 KiB row: {b/2/ts:.2f} TB/s ({ts:.1f} us)")
