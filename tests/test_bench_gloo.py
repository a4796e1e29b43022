"""bench.py's OWN multi-rank code path on CPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` exactly as
the driver launches it, with the HIP library replaced by tests/bench_stub.py (JSTSP_BENCH_HOOKS) and RCCL by gloo.  What runs is
bench.py's rank / WORLD_SIZE handling, trial-id partition, barriers, the MAX all-reduce of the time, the SUM all-reduce of the
NMSE (headline mode), and sweep_mode's sharded run_sweep + its one all-reduce (--sweep): the first 8-GPU run then exercises
nothing untested but RCCL itself.  (montecarlo.run_sweep alone under gloo: tests/test_multiprocess_gloo.py.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra):
    env = dict(os.environ, JSTSP_BENCH_HOOKS="tests.bench_stub:HOOKS", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_headline_mode_two_ranks_partition_and_all_reduce():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bench_stub import trial_value
    batch, steps = 5, 3
    two = _run(2, ["--steps", str(steps), "--warmup", "1", "--batch", str(batch)])
    one = _run(1, ["--steps", str(steps), "--warmup", "1", "--batch", str(batch)])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["steps"] == steps and two["scaling"] == "weak"
    assert two["data"] == "stub"                               # never mistaken for a measurement
    # weak scaling: rank r solves global trials [r batch, (r + 1) batch); the reduced mean covers all 2 batch of them
    assert two["trial_ids_rank0"] == [0, batch - 1]
    assert abs(two["mean_nmse"] - float(trial_value(range(2 * batch)).mean())) < 1e-12
    assert abs(one["mean_nmse"] - float(trial_value(range(batch)).mean())) < 1e-12
    # value = whole-job trials / max-over-ranks time
    assert abs(two["value"] - 2 * batch * steps / (two["ms_per_step"] * 1e-3 * steps)) / two["value"] < 1e-2


@pytest.mark.timeout(900)
def test_sweep_mode_two_ranks_equals_one_rank():
    args = ["--sweep", "--small", "--sweep-trials", "3", "--batch", "2"]
    two = _run(2, args)
    one = _run(1, args)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["data"] == "stub"
    assert len(two["snr_db"]) == 11 and two["snr_db"] == one["snr_db"]
    np.testing.assert_allclose(two["mean_nmse_proposed"], one["mean_nmse_proposed"], rtol=0, atol=1e-6)   # (rounded to 6 places in the line)
    np.testing.assert_allclose(two["mean_nmse_angles"], one["mean_nmse_angles"], rtol=0, atol=1e-6)
    assert len(set(two["mean_nmse_proposed"])) > 3             # the points differ: the stub scored real, per-key inputs
    assert abs(two["value"] - 2 * 11 * 3 / (two["ms_per_step"] * 1e-3)) / two["value"] < 1e-2


@pytest.mark.timeout(900)
def test_sweep_mode_eight_ranks_5005_items_uneven_blocks_equal_one_rank():
    """The driver's 8-GPU command line (BASELINE configs[3]: plot_errorVSsnr.m:48-51,170 sharded over 8 ranks) under gloo: 11 points x
    455 realisations = 5005 (point, trial) items (the reference-native shape: the CPU builder of the stub makes real inputs per key,
    full-size ones would take 20 minutes here) over 8 ranks = blocks of 626 / 625 items, walked in calls of at most 48 trials - the
    last call of a rank is ragged, and a rank's block straddles sweep points.  Per-point means identical to the 1-rank run."""
    args = ["--sweep", "--small", "--sweep-trials", "455", "--batch", "48"]
    eight = _run(8, args)
    one = _run(1, args)
    assert eight["n_gpus"] == 8 and one["n_gpus"] == 1 and eight["scaling"] == "strong" and eight["data"] == "stub"
    assert eight["config"]["parallelism"].endswith("dp8")
    assert len(eight["snr_db"]) == 11 and eight["snr_db"] == one["snr_db"]
    np.testing.assert_allclose(eight["mean_nmse_proposed"], one["mean_nmse_proposed"], rtol=0, atol=1e-6)   # (6 decimals in the line)
    np.testing.assert_allclose(eight["mean_nmse_angles"], one["mean_nmse_angles"], rtol=0, atol=1e-6)
    assert abs(eight["value"] - 2 * 11 * 455 / (eight["ms_per_step"] * 1e-3)) / eight["value"] < 1e-2


def test_eight_gpus_requested_on_a_smaller_node_exits_2_before_touching_the_gpu():
    """`python bench.py --gpus 8` without a launcher on a node with fewer GPUs (here: none): exit code 2, a message on stderr, no
    JSON line - decided from torch.cuda.device_count(), which does not initialise the GPU (bench.py: launch_ranks)."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "JSTSP_BENCH_HOOKS"):
        env.pop(k, None)
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--sweep"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 8 requested" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
