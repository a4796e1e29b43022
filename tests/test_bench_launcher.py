"""bench.py as a launcher (no GPU needed): `--gpus N` without a launcher spawns the ranks itself or refuses loudly;
a WORLD_SIZE that contradicts --gpus is an error - it never prints a line with another n_gpus."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_more_gpus_than_the_node_has_is_refused_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    r = _run(["--gpus", str(n)])
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "--gpus %d" % n in r.stderr


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE=4" in r.stderr
    r = _run(["--gpus", "0"])
    assert r.returncode != 0
