"""bench.py's multi-rank plumbing without a GPU: `python bench.py --gpus 2` with no launcher environment must spawn two ranks itself
(torch.distributed.run on 127.0.0.1), time K steps between barriers, take the MAX over ranks and print ONE JSON line with n_gpus = 2;
the same command under an existing launcher environment must not spawn again.  The GPU work itself is covered by
tests/test_gpu_unet.py::test_bench_two_ranks_gloo_end_to_end (-m gpu)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected exactly one JSON line, got {len(lines)}:\n{r.stdout[-2000:]}"
    return json.loads(lines[0])


def test_gpus_flag_self_launches_ranks():
    out = _run(["--gpus", "2", "--backend", "gloo", "--selftest-launch", "--steps", "5", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["config"]["parallelism"] == "frames2"
    # slowest rank sleeps 2 ms per step: MAX over ranks, not rank 0's own 1 ms
    assert out["ms_per_step"] >= 2.0, out


def test_single_rank_does_not_spawn():
    out = _run(["--gpus", "1", "--backend", "gloo", "--selftest-launch", "--steps", "3", "--warmup", "0"],
               env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29653"})
    assert out["n_gpus"] == 1
