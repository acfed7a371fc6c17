"""Single-rank RCCL check, run as a script by test_policy_gpu.py (own process)."""
import os
import socket
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")]
import torch
import torch.distributed as dist

import cases
import vlnce_amd
from oracle import thirdparty as tp
from vlnce_amd.distributed import GradientAllReducer
from vlnce_amd.il_harness import update_agent

DEV = "cuda:0"
GOLD = os.path.join(HERE, "golden")


def to_dev(x):
    if isinstance(x, dict):
        return {k: to_dev(v) for k, v in x.items()}
    return x.to(DEV) if isinstance(x, torch.Tensor) else x


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    case = cases.CASES["cma_update_64"]
    obs, prev, masks, extra, _ = cases.load_case(os.path.join(GOLD, "cma_update_64.npz"))
    grads = []
    for use in (False, True):
        policy, _ = cases.build_policy(vlnce_amd, case, vlnce_amd.make_config,
                                       vlnce_amd.make_spaces, tp.synth_state_dict)
        policy.to(DEV)
        red = GradientAllReducer(policy, bucket_bytes=1 << 20) if use else None
        vlnce_amd.AuxLosses.activate()
        update_agent(policy, None, to_dev(obs), to_dev(prev), to_dev(masks),
                     to_dev(extra["targets"]), to_dev(extra["weights"]), 512, step_grad=False,
                     grad_hook=red.finish if use else None)
        vlnce_amd.AuxLosses.deactivate()
        torch.cuda.synchronize()
        grads.append({n: p.grad.clone() for n, p in policy.named_parameters()
                      if p.grad is not None})
        if red is not None:
            assert len(red.buckets) > 2
            red.remove()
    assert set(grads[0]) == set(grads[1]) and len(grads[0]) > 20
    for n in grads[0]:
        # two separate forward/backward passes: at this batch size the small-M trunk convolutions
        # take the split-K path, whose atomic summation order differs run to run (~1e-7 on the
        # features, amplified by batch-statistics BatchNorm), so equality is to 1e-4, not bitwise
        scale = grads[1][n].abs().max().item() + 1e-12
        assert (grads[0][n] - grads[1][n]).abs().max().item() <= 1e-4 * scale + 1e-7, n
    # trainable encoders: ~100 MB of gradients in 8 MiB buckets, several collectives in flight on
    # the communication stream while the trunk's backward convolutions still run; from the second
    # step on every bucket must have been issued from the hooks (none left for finish())
    grads_t = []
    for use in (False, False, True):   # two runs without the reducer = the run-to-run floor
        case_t = dict(case, overrides={**case["overrides"], "RGB_ENCODER.trainable": True,
                                       "DEPTH_ENCODER.trainable": True})
        policy, _ = cases.build_policy(vlnce_amd, case_t, vlnce_amd.make_config,
                                       vlnce_amd.make_spaces, tp.synth_state_dict)
        policy.to(DEV)
        red = GradientAllReducer(policy) if use else None
        vlnce_amd.AuxLosses.activate()
        for step in range(2):
            policy.zero_grad()
            update_agent(policy, None, to_dev(obs), to_dev(prev), to_dev(masks),
                         to_dev(extra["targets"]), to_dev(extra["weights"]), 512, step_grad=False,
                         grad_hook=red.finish if use else None)
            if use and step == 1:
                assert len(red.buckets) >= 8, len(red.buckets)
                assert red.launched_before_finish >= len(red.buckets) - 1, (
                    red.launched_before_finish, len(red.buckets))
        vlnce_amd.AuxLosses.deactivate()
        torch.cuda.synchronize()
        grads_t.append({n: p.grad.clone() for n, p in policy.named_parameters()
                        if p.grad is not None})
        if red is not None:
            red.remove()
    assert set(grads_t[0]) == set(grads_t[2]) and len(grads_t[0]) > 300

    gmax = max(t.abs().max().item() for t in grads_t[0].values())

    def worst(a, b):
        # relative to the tensor's own scale, floored at 1e-3 of the largest gradient (text_k.bias
        # has a mathematically zero gradient: its values are rounding noise)
        w, name = 0.0, ""
        for n in a:
            d = (a[n] - b[n]).abs().max().item() / max(b[n].abs().max().item(), 1e-3 * gmax)
            if d > w:
                w, name = d, n
        return w, name

    floor, fname = worst(grads_t[0], grads_t[1])
    got, gname = worst(grads_t[2], grads_t[0])
    print(f"trainable encoders: {len(grads_t[0])} gradients; worst relative difference between two "
          f"runs without the reducer {floor:.2e} ({fname}), with vs without {got:.2e} ({gname})",
          flush=True)
    # (a trainable 50-layer trunk with batch-statistics BatchNorm at 6 frames: split-K atomics
    # order differs run to run and is amplified down to the stem; the reducer -- AVG over one rank
    # -- must not add to that)
    assert got <= 3 * floor + 1e-3, (got, floor)
    print("RCCL-SINGLE-RANK-OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
