import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlnce_amd  # noqa: F401  (registers the package alias)
from vlnce_amd import ops
dev = torch.device("cuda:0")
for n, dt in ((64, torch.float32), (64, torch.uint8), (416, torch.float32)):
    x = torch.randint(0, 256, (n, 256, 256, 3)).to(dt).to(dev)
    fr = ops.frames(x)
    w = torch.randn(64, 7, 7, 3, device=dev) * 0.08
    wf = ops.stem7_pack_weights(w)
    acc = torch.zeros((16, 64, 2), device=dev, dtype=torch.float64)
    sc = torch.full((3,), 1 / 255.0, device=dev); sh = torch.zeros(3, device=dev)
    for mode in ("bn", "eval"):
        kw = dict(bn_acc=acc) if mode == "bn" else dict(scale=torch.ones(64, device=dev), shift=torch.zeros(64, device=dev), act=1)
        for _ in range(3): ops.stem7(fr, wf, 64, sc, sh, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.stem7(fr, wf, 64, sc, sh, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"stem7 {n} frames {dt} {mode}: {us:.1f} us = {2*n*128*128*147*64/us/1e6:.1f} TF/s algorithmic, output {n*128*128*64*4/us/1e6:.2f} TB/s")
