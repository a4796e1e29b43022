"""bench.py prints ONE JSON line with the driver's contract fields (plus roofline / cpu_baseline / parity), and
__graft_entry__.smoke() runs — both on the reference-native shape so the test stays short."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line_small_shape():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--cpu-trials", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert j["value"] > 0 and abs(j["value"] - 8 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-3
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert j["parity"]["max_abs_dNMSE"] < 1e-6


def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()


def test_bench_host_path_and_sweep_mode_small_shape():
    """The line carries the drop-in (JSTSP_HOST) rate next to the device-resident value, and `--sweep` (BASELINE configs[3]
    at the reference-native shape here) prints one line with the per-point mean NMSE of both solvers."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--steps", "1", "--warmup", "0",
                        "--batch", "8", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    hp = j["host_path"]
    assert hp["value"] > 0 and hp["bit_identical_to_device_call"] is True and hp["h2d_gib"] > 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--sweep", "--sweep-trials", "16", "--batch",
                        "16"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["scaling"] == "strong" and "configs[3]" in j["config"]["workload"]
    assert len(j["snr_db"]) == 11 == len(j["mean_nmse_proposed"]) == len(j["mean_nmse_angles"])
    assert all(0 < a <= b <= 1 for a, b in zip(j["mean_nmse_angles"], j["mean_nmse_proposed"]))     # the genie support helps
    assert abs(j["value"] - 2 * 11 * 16 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3


def test_bench_under_the_launcher_with_one_rank_uses_rccl():
    """The multi-GPU code path with the one GPU a test box has: torch.distributed.run starts ONE rank, bench.py initialises
    the RCCL process group (WORLD_SIZE is set), barriers, all-reduces the timing MAX and the NMSE sum, and prints the line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in (["--steps", "1", "--warmup", "0", "--batch", "8", "--no-cpu-baseline", "--no-host-path"],
                  ["--sweep", "--sweep-trials", "8", "--batch", "8"]):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                            "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--small"] + extra,
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1 and json.loads(lines[0])["value"] > 0


def test_bench_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """The first N > 1 RCCL run of this code should not be the driver's 8-GPU scaling run: on any box with at least two GPUs
    this launches `bench.py --gpus 2` (headline step and configs[3] sweep; bench.py starts its two ranks itself) and checks
    the line - trials of both ranks counted, the sweep's per-point means equal to the one-rank run's (the trial -> rank
    partition must not change a result: inputs are keyed by global (point, trial) ids, the NMSE sums meet in ONE all-reduce).
    Self-skips on the 1-GPU boxes of the test pool."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = lambda args: subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--small"] + args, capture_output=True,
                                      text=True, timeout=1200, cwd=ROOT, env=env)
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert abs(j["value"] - 2 * 8 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3          # whole-job: both ranks' trials
    two = run(["--gpus", "2", "--sweep", "--sweep-trials", "12", "--batch", "8"])
    one = run(["--sweep", "--sweep-trials", "12", "--batch", "8"])
    assert two.returncode == 0 and one.returncode == 0, (two.stderr[-1500:], one.stderr[-1500:])
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j2["mean_nmse_proposed"] == pytest.approx(j1["mean_nmse_proposed"], abs=2e-6)
    assert j2["mean_nmse_angles"] == pytest.approx(j1["mean_nmse_angles"], abs=2e-6)


def test_bench_two_ranks_real_library_on_one_gpu_over_gloo():
    """N = 2 with the REAL library on the one GPU a test box has: both ranks on device 0, gloo instead of RCCL
    (tests/bench_stub.py ONE_GPU_HOOKS; everything else is bench.py as the driver launches it).  Rank r solves global trials
    [8 r, 8 r + 8): the reduced mean NMSE must be the one-rank run's over trials [0, 16) - a trial's result does not depend on
    the batch it is solved in, so the two differ by the summation order of float64 partial sums only - and the sweep's
    per-point means must be the one-rank sweep's.  What is left for the first 8-GPU run to exercise is RCCL itself."""
    import socket
    env = dict(os.environ, JSTSP_BENCH_HOOKS="tests.bench_stub:ONE_GPU_HOOKS", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def run(world, extra):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--small"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    head = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"]
    two = run(2, head + ["--batch", "8"])
    one = run(1, head + ["--batch", "16"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["data"] == "synthetic" and one["n_gpus"] == 1
    assert abs(two["value"] - 2 * 8 / (two["ms_per_step"] * 1e-3)) / two["value"] < 1e-3       # whole-job: both ranks' trials
    assert 0 < two["mean_nmse"] < 1 and abs(two["mean_nmse"] - one["mean_nmse"]) < 1e-12, (two["mean_nmse"], one["mean_nmse"])
    assert two["end_to_end"]["value"] > 0 and two["cpu_baseline"] is None     # (the CPU leg is rank 0 at N = 1 only)
    sw = ["--sweep", "--sweep-trials", "12", "--batch", "8"]
    s2, s1 = run(2, sw), run(1, sw)
    assert s2["n_gpus"] == 2 and s2["scaling"] == "strong" and s1["n_gpus"] == 1
    assert s2["mean_nmse_proposed"] == pytest.approx(s1["mean_nmse_proposed"], abs=2e-6)
    assert s2["mean_nmse_angles"] == pytest.approx(s1["mean_nmse_angles"], abs=2e-6)
