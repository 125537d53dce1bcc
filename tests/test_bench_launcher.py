"""bench.py's own launcher: `python bench.py --gpus N` outside torchrun must start N ranks itself and
relay rank 0's JSON line; a --gpus / WORLD_SIZE mismatch must fail instead of running one rank.  CPU only: --dry runs the
N > 1 control flow (rendezvous on 127.0.0.1, barrier-bracketed timed region with the all-gather inside, MAX over ranks) on
gloo with a stub forward."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--dry", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["dry"] is True and j["launched_by"] == "bench.py"
    assert j["multi_gpu"]["rccl_world"] == 2 and j["multi_gpu"]["backend"] == "gloo" and j["multi_gpu"]["gather_in_timed_region"]
    assert j["value"] > 0 and j["scaling"] == "weak"


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--dry"], {"WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--dry"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0


def test_single_rank_dry_line():
    r = _run(["--dry", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["multi_gpu"] is None
