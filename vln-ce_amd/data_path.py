"""Cached-feature DAgger data path on the device.

Upstream, `collate_fn` (dagger_trainer.py:39-114) pads every trajectory of a batch to the
longest one on the host (observations with 1.0, actions / weights with 0), interleaves them
time-major (row t*B + b), and the train loop then copies the padded tensors to the GPU casting
every sensor to fp32 (:559-583); the inflection weights come from
IWTrajectoryDataset.__next__ (:196-208) and the length-bucketed ordering from _load_next
(:174-184).  Here the host only concatenates the RAGGED rows (pinned, in the storage dtype --
fp16 if the LMDB cache was written with IL.DAGGER.lmdb_fp16) and ships them once; padding,
interleave, widening, inflection weights and masks are one kernel pass per sensor on the GPU.
The returned 5-tuple is what `_update_agent` takes, already on the device.
"""
import random

import torch

from . import ops


_STAGE = {}         # slot -> pinned uint8 buffer
_STAGE_EVENTS = {}  # slot -> event recorded after the last H2D copy out of that buffer


def _staging(slot, dtype, shape):
    n = 1
    for s in shape:
        n *= s
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    ev = _STAGE_EVENTS.get(slot)
    if ev is not None:
        ev.synchronize()
    buf = _STAGE.get(slot)
    if buf is None or buf.numel() < nbytes:
        buf = _STAGE[slot] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
    return buf[:nbytes].view(dtype).view(shape)


def inflection_weights_host(oracle_actions, coef):
    """reference values on the host (used by tests and by callers that stay on the CPU path)."""
    infl = torch.ones_like(oracle_actions, dtype=torch.bool)
    infl[1:] = oracle_actions[1:] != oracle_actions[:-1]
    return torch.where(infl, torch.tensor(float(coef)), torch.tensor(1.0))


def bucketed_order(lengths, batch_size, rng=random):
    """order in which IWTrajectoryDataset._load_next hands out a preload chunk: sort by
    (length, random tie-break), shuffle blocks of `batch_size`, consumed from the END (.pop())."""
    n = len(lengths)
    prio = list(range(n))
    rng.shuffle(prio)
    order = sorted(range(n), key=lambda k: (lengths[k], prio[k]))
    blocks = [order[i:i + batch_size] for i in range(0, n, batch_size)]
    rng.shuffle(blocks)
    flat = [k for blk in blocks for k in blk]
    return flat[::-1]


def collate_trajectories(batch, device, inflection_coef=1.0, pin=True):
    """batch: list of (obs dict of [T_b, ...] tensors / arrays, prev_actions [T_b],
    oracle_actions [T_b]) as stored in the LMDB feature cache.  Returns
    (observations {sensor: [Tmax*B, ...] fp32}, prev_actions [Tmax*B, 1] int64,
     not_done_masks [Tmax*B, 1] uint8, corrected_actions [Tmax, B] int64, weights [Tmax, B] fp32)
    on `device`."""
    lib = ops.L()
    B = len(batch)
    lens = [int(torch.as_tensor(tr[1]).shape[0]) for tr in batch]
    Tmax = max(lens)
    off = torch.zeros(B + 1, dtype=torch.int32)
    off[1:] = torch.tensor(lens, dtype=torch.int32).cumsum(0)

    use_pinned = pin and str(device) != "cpu" and torch.cuda.is_available()

    def ship(parts, slot):
        parts = [torch.as_tensor(p) for p in parts]
        if not use_pinned:
            return torch.cat(parts, dim=0).contiguous().to(device)
        # rows go straight into a persistent pinned staging buffer (one per slot, grown on
        # demand), then one asynchronous H2D copy; the buffer is reused by the next batch only
        # after that copy has completed
        shape = (sum(p.shape[0] for p in parts),) + tuple(parts[0].shape[1:])
        stage = _staging(slot, parts[0].dtype, shape)
        row = 0
        for p in parts:
            stage[row:row + p.shape[0]].copy_(p)
            row += p.shape[0]
        out = stage.to(device, non_blocking=True)
        _STAGE_EVENTS[slot] = torch.cuda.Event()
        _STAGE_EVENTS[slot].record()
        return out

    off_d = off.to(device, non_blocking=True)
    observations = {}
    for sensor in batch[0][0]:
        rows = ship([tr[0][sensor] for tr in batch], "obs/" + sensor)
        if rows.dtype not in (torch.float32, torch.float16, torch.int64):
            rows = rows.to(torch.int64 if not rows.is_floating_point() else torch.float32)
        tail = tuple(rows.shape[1:])
        D = 1
        for s in tail:
            D *= s
        dst = torch.empty((Tmax * B,) + tail, device=device, dtype=torch.float32)
        lib.ragged_pad_rows(rows, off_d, B, Tmax, D, 1.0, dst)  # observations pad = 1.0 (:77)
        observations[sensor] = dst
    prev = ship([torch.as_tensor(tr[1]).to(torch.int64) for tr in batch], "prev")
    prev_out = torch.empty((Tmax * B, 1), device=device, dtype=torch.int64)
    lib.ragged_pad_rows_i64(prev, off_d, B, Tmax, 1, 0, prev_out)
    oracle = ship([torch.as_tensor(tr[2]).to(torch.int64) for tr in batch], "oracle")
    corrected = torch.empty((Tmax, B), device=device, dtype=torch.int64)
    weights = torch.empty((Tmax, B), device=device, dtype=torch.float32)
    masks = torch.empty((Tmax * B, 1), device=device, dtype=torch.uint8)
    lib.dagger_targets(oracle, off_d, B, Tmax, inflection_coef, corrected, weights, masks)
    return observations, prev_out, masks, corrected, weights
