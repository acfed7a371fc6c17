"""Seeded ragged trajectories for the data-path goldens (shape of what the LMDB feature cache
holds: dagger_trainer.py:340-362 -- obs dict of per-step arrays, prev_actions, oracle_actions)."""
import numpy as np

CASES = {
    # fp16 feature cache (IL.DAGGER.lmdb_fp16), inflection weighting on
    "collate_fp16_iw": dict(lengths=[5, 9, 3, 9], fp16=True, coef=3.2, seed=11),
    # fp32 cache, no inflection weighting, a single-step trajectory and a batch of one long one
    "collate_fp32_plain": dict(lengths=[1, 7, 2], fp16=False, coef=1.0, seed=12),
    "collate_single": dict(lengths=[6], fp16=True, coef=2.0, seed=13),
}


def build_trajectories(spec):
    rng = np.random.RandomState(spec["seed"])
    ft = np.float16 if spec["fp16"] else np.float32
    out = []
    for T in spec["lengths"]:
        obs = {
            "rgb_features": rng.rand(T, 8, 2, 2).astype(ft),
            "depth_features": rng.rand(T, 4, 2, 2).astype(np.float32),
            "instruction": np.tile(rng.randint(1, 2504, size=(1, 10)), (T, 1)).astype(np.int64),
            "progress": rng.rand(T, 1).astype(np.float32),  # D = 1: scalar (non-vector) path
        }
        oracle = rng.randint(0, 4, size=T).astype(np.int64)
        oracle[T // 2:] = oracle[T // 2]  # a run of equal actions: weights of 1.0 in it
        prev = np.concatenate([[0], oracle[:-1]]).astype(np.int64)
        out.append((obs, prev, oracle))
    return out


def load(path):
    z = np.load(path, allow_pickle=False)
    n = 1 + max(int(k.split("/")[1]) for k in z.files if k.startswith("in/"))
    trajs = []
    for i in range(n):
        obs = {k.split("/", 3)[3]: z[k] for k in z.files if k.startswith(f"in/{i}/obs/")}
        trajs.append((obs, z[f"in/{i}/prev"], z[f"in/{i}/oracle"]))
    outs = {k[4:]: z[k] for k in z.files if k.startswith("out/")}
    return trajs, outs
