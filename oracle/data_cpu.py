"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's cached-feature DAgger data
path.  Only tests/ (and benchmarks' cpu baselines) may import this.

Restates, in plain torch on the host:
  * collate_fn                      vlnce_baselines/dagger_trainer.py:39-114
  * the fp32 cast of every sensor   vlnce_baselines/dagger_trainer.py:559-566
  * inflection weights              vlnce_baselines/dagger_trainer.py:196-208
  * _block_shuffle / bucketed load  vlnce_baselines/dagger_trainer.py:117-121,174-184
Pinned by tests/golden/collate_*.npz, produced by executing the reference's own collate_fn /
_block_shuffle source (extracted with ast, tests/golden/make_goldens_data.py).
"""
import random

import torch


def pad_to(t, length, fill):
    if t.size(0) == length:
        return t
    pad = torch.full((length - t.size(0),) + tuple(t.shape[1:]), fill, dtype=t.dtype)
    return torch.cat([t, pad], dim=0)


def inflection_weights(oracle_actions, coef=1.0, use_iw=True):
    table = torch.tensor([1.0, coef if use_iw else 1.0])
    infl = torch.cat([torch.ones(1, dtype=torch.long),
                      (oracle_actions[1:] != oracle_actions[:-1]).long()])
    return table[infl]


def collate(batch):
    """batch: list of (obs dict, prev_actions, oracle_actions, weights); returns the 5-tuple of
    the reference's collate_fn with every sensor cast to fp32 as its train loop does."""
    B = len(batch)
    T = max(s[1].size(0) for s in batch)
    obs = {}
    for sensor in batch[0][0]:
        cols = [pad_to(s[0][sensor], T, 1.0) for s in batch]  # fill_val=1.0 for EVERY sensor
        st = torch.stack(cols, dim=1)
        obs[sensor] = st.view(-1, *st.shape[2:]).to(torch.float32)
    prev = torch.stack([pad_to(s[1], T, 0) for s in batch], dim=1)
    corrected = torch.stack([pad_to(s[2], T, 0) for s in batch], dim=1)
    weights = torch.stack([pad_to(s[3], T, 0) for s in batch], dim=1)
    masks = torch.ones_like(corrected, dtype=torch.uint8)
    masks[0] = 0
    return obs, prev.view(-1, 1), masks.view(-1, 1), corrected, weights


def block_shuffle(lst, block_size, rng=random):
    blocks = [lst[i:i + block_size] for i in range(0, len(lst), block_size)]
    rng.shuffle(blocks)
    return [e for blk in blocks for e in blk]


def bucketed_order(lengths, batch_size, rng=random):
    """sequence in which _load_next pops a preload chunk (lines 174-184, then .pop())."""
    prio = list(range(len(lengths)))
    rng.shuffle(prio)
    order = list(range(len(lengths)))
    order.sort(key=lambda k: (lengths[k], prio[k]))
    return block_shuffle(order, batch_size, rng)[::-1]
