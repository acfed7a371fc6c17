"""Golden vectors for the cached-feature DAgger data path, produced by the
reference's OWN code: `collate_fn`, `_block_shuffle` and `IWTrajectoryDataset.__next__` are
extracted from vlnce_baselines/dagger_trainer.py with `ast` (the module itself cannot be imported:
lmdb / msgpack_numpy / habitat) and executed on seeded ragged trajectories.  CPU container only.

    python tests/golden/make_goldens_data.py
"""
import ast
import os
import random
import sys
import types
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases_data  # noqa: E402
from tools.oracle import shims  # noqa: E402


def extract(names):
    path = os.path.join(shims.REFERENCE_ROOT, "vlnce_baselines/dagger_trainer.py")
    tree = ast.parse(open(path).read())
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            found[node.name] = node
    scope = {"torch": torch, "np": np, "random": random, "defaultdict": defaultdict,
             "ObservationsDict": dict}
    mod = ast.Module(body=[found[n] for n in names], type_ignores=[])
    exec(compile(mod, path, "exec"), scope)
    return [scope[n] for n in names]


def main():
    collate_fn, block_shuffle, dataset_next = extract(["collate_fn", "_block_shuffle", "__next__"])
    for name, spec in cases_data.CASES.items():
        trajs = cases_data.build_trajectories(spec)
        samples = []
        for obs, prev, oracle in trajs:
            fake = types.SimpleNamespace(
                _load_next=lambda o=obs, p=prev, a=oracle: ({k: v.copy() for k, v in o.items()},
                                                            p.copy(), a.copy()),
                inflec_weights=torch.tensor([1.0, spec["coef"]]))
            samples.append(dataset_next(fake))  # (obs tensors, prev, oracle, inflection weights)
        obs_b, prev_b, masks_b, corrected_b, weights_b = collate_fn(samples)
        # the train loop's cast (dagger_trainer.py:559-566): every sensor -> float32
        obs_b = {k: v.to(dtype=torch.float32) for k, v in obs_b.items()}
        blob = {}
        for i, (obs, prev, oracle) in enumerate(trajs):
            for k, v in obs.items():
                blob[f"in/{i}/obs/{k}"] = v
            blob[f"in/{i}/prev"] = prev
            blob[f"in/{i}/oracle"] = oracle
        for k, v in obs_b.items():
            blob[f"out/obs/{k}"] = v.numpy()
        blob["out/prev_actions"] = prev_b.numpy()
        blob["out/not_done_masks"] = masks_b.numpy()
        blob["out/corrected_actions"] = corrected_b.numpy()
        blob["out/weights"] = weights_b.numpy()
        random.seed(spec["seed"])
        blob["out/block_shuffle"] = np.array(block_shuffle(list(range(23)), 4))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, {k: v.shape for k, v in blob.items() if k.startswith("out/")})


if __name__ == "__main__":
    main()
