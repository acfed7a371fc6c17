#!/bin/bash
# attn_bwd with 16-byte row accesses: tests + time of the text attention's backward at the step's and the cached update's shapes
O=$GRAFT_REPO_ROOT/gpurun_out/r04_65
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" -p no:cacheprovider 2>&1 | tail -2
timeout 200 python - <<'P' | tee $O/attn_bwd_times.txt
import torch, sys
sys.path.insert(0, '.')
from vlnce_amd import ops
dev = 'cuda:0'
for B, P, Dk, Dv, U in ((64, 80, 256, 256, 0), (64, 16, 256, 256, 0), (500, 200, 256, 256, 5), (500, 16, 256, 256, 0)):
    q = torch.randn(B, Dk, device=dev, requires_grad=True)
    n = U or B
    K = torch.randn(n, P, Dk, device=dev, requires_grad=True)
    V = torch.randn(n, P, Dv, device=dev, requires_grad=True)
    idx = (torch.arange(B, device=dev) % U) if U else None
    g = torch.randn(B, Dv, device=dev)
    best = 1e9
    for rep in range(5):
        out = ops.attention(q, K, V, None, 1, Dk ** -0.5, index=idx)
        torch.cuda._sleep(int(2e7))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out.backward(g)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    print(f"attention backward B={B} P={P} Dk={Dk} Dv={Dv} shared={U}: {best:.1f} us (incl. autograd's own launches)")
P
