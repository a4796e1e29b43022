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
