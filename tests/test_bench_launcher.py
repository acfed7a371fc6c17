"""`python bench.py --gpus N` must start its N ranks itself when no launcher set WORLD_SIZE
(the driver's multi-GPU command is the N=1 command with the number changed), keep working under
torch.distributed.run, and print exactly ONE JSON line on stdout from rank 0 with n_gpus = N.
Run here on CPU through tests/bench_on_cpu_simulator.py, which puts tests/hostsim.py behind the
package and gloo behind torch.distributed and then calls bench.main(): the launcher, the rank
set-up, the sharded update through GradientAllReducer and the max-over-ranks timing are bench.py's
real code; nothing is measured.  (bench.py itself imports nothing from tests/.)"""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--num-envs", "2", "--hw", "64", "--tokens", "7",
        "--no-cpu-baseline", "--no-f32-compare"]
DRIVER = os.path.join(REPO, "tests", "bench_on_cpu_simulator.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "4"
    return env


def _one_json_line(stdout, n):
    lines = [ln for ln in stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["parallelism"] == f"dp{n}" and d["config"]["global_batch"] == 2 * n
    assert d["ms_per_step"] > 0 and d["scaling"] == "weak"
    assert "hostsim" in d["data"] and d["value"] is None   # never mistaken for a measurement
    # the gradient exchange of the last step: total time of the bucket collectives and the share of
    # it that ran before backward had ended (GradientAllReducer.stats)
    assert d["allreduce_ms"] > 0 and 0.0 <= d["allreduce_hidden_frac"] <= 1.0
    assert d["allreduce"]["buckets"] >= 1
    return d


def test_bench_gpus_2_launches_its_own_ranks():
    r = subprocess.run([sys.executable, DRIVER, "--gpus", "2"] + ARGS,
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    _one_json_line(r.stdout, 2)
    assert "without a launcher: starting 2 ranks" in r.stderr


def test_bench_gpus_2_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), DRIVER, "--gpus", "2"] + ARGS,
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    _one_json_line(r.stdout, 2)
    assert "without a launcher" not in r.stderr


def test_bench_rejects_a_world_size_that_disagrees_with_gpus():
    env = dict(_clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, DRIVER, "--gpus", "4"] + ARGS,
                       env=env, capture_output=True, text=True, timeout=120, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_policy_waypoint_and_seq2seq_run_data_parallel():
    """BASELINE.json configs[4] (WaypointPolicy DD-PPO, `--policy waypoint`: evaluate_actions +
    WDDPPO minibatch update with the never-used `action_distribution` head in the reducer's first
    bucket, ddppo_waypoint_trainer.py:370) and configs[1] (`--policy seq2seq`) launch under
    `--gpus 2` like the headline and print the same one line."""
    for pol, workload in (("waypoint", "WaypointPolicy WDDPPO minibatch update"),
                          ("seq2seq", "Seq2Seq policy DAgger update")):
        r = subprocess.run([sys.executable, DRIVER, "--gpus", "2", "--policy", pol]
                           + ARGS, env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=REPO)
        assert r.returncode == 0, r.stderr[-3000:]
        d = _one_json_line(r.stdout, 2)
        assert workload in d["config"]["workload"]
