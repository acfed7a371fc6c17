"""repro harness for the run-ahead pipeline + graph capture interplay (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch

import vlnce_amd
from vlnce_amd.il_harness import update_agent

mode = sys.argv[1]
dev = "cuda:0"
torch.manual_seed(0)
policy = vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"), *vlnce_amd.make_spaces(64, 64)).to(dev)
opt = torch.optim.Adam(policy.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)


def batch(seed):
    g = torch.Generator().manual_seed(seed)
    obs = {"rgb": torch.randint(0, 256, (6, 64, 64, 3), generator=g).float().to(dev),
           "depth": torch.rand(6, 64, 64, 1, generator=g).to(dev),
           "instruction": torch.zeros(6, 200, dtype=torch.long)}
    obs["instruction"][:, :9] = torch.randint(1, 2504, (6, 9), generator=g)
    obs["instruction"] = obs["instruction"].to(dev)
    return obs


prev = torch.randint(0, 4, (6, 1), generator=g).to(dev)
masks = torch.ones(6, 1, dtype=torch.uint8, device=dev)
tgt = torch.randint(0, 4, (3, 2), generator=g).to(dev)
w = (torch.rand(3, 2, generator=g) + 0.5).to(dev)
obs = [batch(s) for s in range(6)]
if mode == "plain_first":
    for k in range(2):
        update_agent(policy, opt, obs[k], prev, masks, tgt, w, 512)
    torch.cuda.synchronize()
if mode == "sync_between":
    os.environ["X"] = "1"
def ahead(o, k):
    if mode == "identity_first2" and k < 2:
        return o
    if mode == "dictcopy_first2" and k < 2:
        return dict(o)
    return policy.encode_ahead(o)


nxt = ahead(obs[0], 0)
for k in range(5):
    cur = nxt
    if k + 1 < 5:
        nxt = ahead(obs[k + 1], k)
    if mode == "sync_between":
        torch.cuda.synchronize()
    loss, _, _ = update_agent(policy, opt, cur, prev, masks, tgt, w, 512)
    if mode != "noprint":
        print(mode, k, float(loss), flush=True)
torch.cuda.synchronize()
if mode == "new_sig":
    # a new tail signature (no_grad act) appears while run-ahead is in use
    h0 = torch.zeros(6, 2, 512, device=dev)
    with torch.no_grad():
        for k in range(4):
            cur = policy.encode_ahead(obs[k])
            a, h = policy.act(cur, h0, prev, masks, deterministic=True)
            print("act", k, a.view(-1).tolist(), flush=True)
    loss, _, _ = update_agent(policy, opt, policy.encode_ahead(obs[0]), prev, masks, tgt, w, 512)
    torch.cuda.synchronize()
print(mode, "OK", flush=True)
