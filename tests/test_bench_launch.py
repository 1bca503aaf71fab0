"""bench.py's launcher contract: `--gpus N` without a rendezvous in the environment re-executes itself
under `python -m torch.distributed.run --nproc-per-node N` (the reference's launcher, README.md:163), the job must BE N
ranks, the timed region takes the MAX over ranks and rank 0 prints one JSON line.  Runs on CPU ranks over gloo with the
test hook `--stub-step` (a step that sleeps; no kernel is involved and the line is marked "stub")."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout)


def test_gpus_2_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--stub-step"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["stub"] is True
    # MAX over ranks: rank 1 sleeps 4 ms per step, rank 0 only 2 ms
    assert out["ms_per_step"] >= 3.9, out


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--stub-step"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    # a launcher that started 1 rank for a run labelled --gpus 2 must fail loudly, not report a 1-GPU number
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-step"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in r.stderr
