"""Stand-in for the HIP library behind bench.py's hooks (TEST INFRASTRUCTURE, CPU tier): deterministic "solves" over the gloo
backend, so that bench.py's own rank / partition / barrier / all-reduce code runs under world_size 2 without a GPU
(tests/test_bench_gloo.py).  A trial's "NMSE" is a function of its GLOBAL trial index only - the reduced mean is then
independent of how trials are sharded, and wrong partitioning shows."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def trial_value(ids):
    t = torch.as_tensor(list(ids), dtype=torch.float64)
    return 0.05 + 0.9 * ((t * 0.6180339887498949) % 1.0)


def _sweep_solve(inp, Imax):
    """run_sweep's solver hook: per-trial numbers from the inputs the CPU builder made for exactly these trials (so a trial built
    for the wrong (point, trial) key changes the mean)."""
    e = inp["subY"].abs().double().mean((1, 2)) / (1.0 + inp["subY"].abs().double().amax((1, 2)))
    ea = inp["Omega"].double().mean((1, 2)) * 0.5 + 0.1 * e
    return e.clamp(max=1.0), ea.clamp(max=1.0)


class StubHooks:
    stub = True
    backend = "gloo"

    @staticmethod
    def device(local):
        torch.set_num_threads(2)
        return torch.device("cpu")

    @staticmethod
    def sync():
        pass

    @staticmethod
    def make_inputs(p, ids, device, shared_pilots):
        return {"ids": list(ids)}

    @staticmethod
    def solve(inp, imax, want_ce):
        import time
        time.sleep(0.02)                        # (so that the line's rounded ms_per_step resolves the step)
        v = trial_value(inp["ids"])
        return v, None, None

    @staticmethod
    def nmse(S, inp):
        return S

    @staticmethod
    def sweep_kw():
        tdir = os.path.join(ROOT, "tests")
        if tdir not in sys.path:
            sys.path.insert(0, tdir)
        from torch_builder import builder
        return {"solve_fn": _sweep_solve, "builder": builder}



def _one_gpu_hooks():
    """GPU tier, 1-GPU box: the REAL library behind bench.py's hooks, every rank on device 0 and gloo instead of RCCL (RCCL wants
    one device per rank).  What that run adds to the stub runs above: the library's per-process contexts, the device input builder
    keyed by the rank's GLOBAL trial ids and the device NMSE behind bench.py's partition and all-reduces - with real solves, so
    the reduced NMSE can be compared with a one-rank run over the same trials (tests/test_gpu_bench_contract.py)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench

    class OneGpuHooks(bench.HipHooks):
        backend = "gloo"

        @staticmethod
        def device(local):
            torch.cuda.set_device(0)
            return torch.device("cuda", 0)

    return OneGpuHooks


def __getattr__(name):              # (bench.py is imported only when these hooks are asked for)
    if name == "ONE_GPU_HOOKS":
        return _one_gpu_hooks()
    raise AttributeError(name)


HOOKS = StubHooks
