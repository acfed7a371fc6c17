"""Golden vectors for DD-PPO returns: RolloutStorage.compute_returns is
extracted from vlnce_baselines/common/rollout_storage.py with `ast` and executed on a fake
storage object.  CPU container only.   python tests/golden/make_goldens_returns.py"""
import ast
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tools.oracle import shims  # noqa: E402


def main():
    path = os.path.join(shims.REFERENCE_ROOT, "vlnce_baselines/common/rollout_storage.py")
    fn = [n for n in ast.walk(ast.parse(open(path).read()))
          if isinstance(n, ast.FunctionDef) and n.name == "compute_returns"][0]
    scope = {"torch": torch, "Tensor": torch.Tensor}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), scope)
    g = torch.Generator().manual_seed(21)
    T, N = 16, 5
    blob = {}
    rewards = torch.randn(T, N, 1, generator=g) * 0.5
    vp = torch.randn(T + 1, N, 1, generator=g)
    masks = (torch.rand(T + 1, N, 1, generator=g) > 0.15).float()
    nv = torch.randn(N, 1, generator=g)
    blob.update(rewards=rewards.numpy(), value_preds=vp.numpy(), masks=masks.numpy(),
                next_value=nv.numpy(), gamma=np.float64(0.99), tau=np.float64(0.95))
    for use_gae in (True, False):
        st = types.SimpleNamespace(rewards=rewards.clone(), value_preds=vp.clone(),
                                   masks=masks.clone(), returns=torch.zeros(T + 1, N, 1), step=T)
        scope["compute_returns"](st, nv, use_gae, 0.99, 0.95)
        blob[f"returns_gae{int(use_gae)}"] = st.returns.numpy()
        blob[f"value_preds_after_gae{int(use_gae)}"] = st.value_preds.numpy()
    np.savez_compressed(os.path.join(HERE, "ppo_returns.npz"), **blob)
    print({k: getattr(v, "shape", v) for k, v in blob.items()})


if __name__ == "__main__":
    main()
