import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, vlnce_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = vlnce_amd.make_config("CMAPolicy")
pol = vlnce_amd.build_model(cfg, *vlnce_amd.make_spaces(256, 256)).to(dev); pol.eval()
batch = bench.synth_batch(8, 256, 80, dev, seed=1)
obs, prev, masks = batch[0], batch[1], batch[2]
n = 1
o = {k: v[:n].contiguous() for k, v in obs.items()}
h0 = torch.zeros(n, pol.net.num_recurrent_layers, 512, device=dev)
with torch.no_grad():
    ahead = pol.encode_ahead(o); torch.cuda.synchronize()
    oo = dict(o, rgb_features=ahead["rgb_features"], depth_features=ahead["depth_features"])
    for _ in range(10): pol.act(oo, h0, prev[:n], masks[:n], deterministic=True)
    torch.cuda.synchronize()
    # host-only issue time (no sync inside the loop)
    t0 = time.perf_counter()
    for _ in range(200): pol.act(oo, h0, prev[:n], masks[:n], deterministic=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"issue {1e3*(t1-t0)/200:.3f} ms/call, drained after {1e3*(t2-t1):.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): pol.act(oo, h0, prev[:n], masks[:n], deterministic=True)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(45)
