import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max |b| — the tolerance metric used throughout the parity tests."""
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


# ---- parity tolerances of proposed_algorithm / proposed_algorithm_angles (include/jstsp.h "Accuracy"), round 6: about 4x the
#      largest error measured on MI355X over the whole -m gpu suite (profiles/r06_measured_tolerances.json), so that a 10x
#      regression of the device arithmetic turns a test red.
TOL_S = 1e-5        # S, Y: max|d| / max|ref|        (measured <= 2.0e-6 over the whole suite)
TOL_CE = 5e-4       # convergence_error, relative per finite entry (measured <= 1.05e-4)
TOL_NMSE = 1e-6     # |dNMSE| per trial: BASELINE.json north_star

_MEASURED = {}


def check_below(name, value, tol):
    """assert value < tol, and keep the largest value seen per name: written to gpurun_out/measured_tolerances.json at session end
    (the evidence the tolerances above were chosen from)."""
    value = float(value)
    m = _MEASURED.setdefault(name, {"max": 0.0, "tol": float(tol), "n": 0})
    m["max"] = max(m["max"], value); m["n"] += 1
    assert value < tol, (name, value, tol)


def ce_rel(ce, ref):
    """largest relative deviation over the finite entries of a convergence_error array (the Inf pattern must be equal)."""
    ce = np.asarray(ce, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    fin = np.isfinite(ref)
    assert ce.shape == ref.shape and np.array_equal(np.isfinite(ce), fin)
    return float(np.max(np.abs(ce[fin] - ref[fin]) / np.abs(ref[fin]))) if fin.any() else 0.0


def pytest_sessionfinish(session, exitstatus):
    if not _MEASURED:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "measured_tolerances.json"), "w") as f:
            json.dump(_MEASURED, f, indent=1, sort_keys=True)
    except OSError:
        pass
