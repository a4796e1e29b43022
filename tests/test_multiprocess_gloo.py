"""N > 1 path on CPU: world_size-2 gloo run of the sweep runner (trial sharding + the single
all-reduce).  The HIP solver cannot run here, so the runner's solver hook is given the
oracle (tests may use it); what is under test is partitioning, RNG keying and the collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_solve(inp, Imax):
    from oracle import solvers as O
    T = inp["subY"].shape[0]
    e, ea = [], []
    A = inp["A"].numpy()
    for t in range(T):
        args = (inp["subY"][t].numpy(), inp["Omega"][t].numpy(), A, inp["B"][t].numpy(), Imax,
                float(inp["tau_Y"][t]), float(inp["tau_Z"][t]), float(inp["rho"][t]), "approximate")
        S, _, _ = O.proposed_algorithm(*args, want_ce=False)
        Sa, _, _ = O.proposed_algorithm(*args, indx_S=inp["indx_S"][t].numpy(), want_ce=False)
        zb = inp["Zbar"][t].numpy()
        e.append(O.nmse_capped(S, zb)); ea.append(O.nmse_capped(Sa, zb))
    return torch.tensor(e), torch.tensor(ea)


def _torch_builder():
    """The CPU-side input builder (tests/torch_builder.py) as the sweep runner's hook."""
    tdir = os.path.join(ROOT, "tests")
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    from torch_builder import builder
    return builder


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from jstsp19_amd.montecarlo import run_sweep
    from jstsp19_amd.system_model import SweepParams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    p = SweepParams(Nt=2, Nr=8, L=2, T=4, Mr=3)
    out = run_sweep(p, [-5.0, 5.0, 15.0], 5, Imax=15, batch=2, device=torch.device("cpu"),
                    solve_fn=_oracle_solve, dist=dist, builder=_torch_builder())
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _worker_approx(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from jstsp19_amd.montecarlo import run_approx_sweep
    from jstsp19_amd.system_model import TrainingParams
    from tests.test_system_model import _oracle_alg12
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    out = run_approx_sweep(TrainingParams(Nt=2, Nr=8, L=2, T=12), [0.0, 10.0], [5, 10], 3, batch=2,
                           device=torch.device("cpu"), solve_fn=_oracle_alg12, dist=dist, builder=_torch_builder())
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_everything_once():
    from jstsp19_amd.montecarlo import partition
    for n in (0, 1, 7, 15, 5000):
        for w in (1, 2, 3, 8):
            blocks = [partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(600)
def test_two_rank_gloo_sweep_equals_single_process():
    from jstsp19_amd.montecarlo import run_sweep
    from jstsp19_amd.system_model import SweepParams
    p = SweepParams(Nt=2, Nr=8, L=2, T=4, Mr=3)
    single = run_sweep(p, [-5.0, 5.0, 15.0], 5, Imax=15, batch=2, device=torch.device("cpu"),
                       solve_fn=_oracle_solve, dist=None, builder=_torch_builder()).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(q.get(timeout=500) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(res[0], res[1])                       # every rank holds the reduced result
    np.testing.assert_allclose(res[0], single, rtol=1e-12)      # independent of the number of ranks
    assert single.shape == (3, 2) and np.all(single > 0) and np.all(single <= 1)


@pytest.mark.timeout(600)
def test_two_rank_gloo_alg1_vs_alg2_sweep_equals_single_process():
    from jstsp19_amd.montecarlo import run_approx_sweep
    from jstsp19_amd.system_model import TrainingParams
    from tests.test_system_model import _oracle_alg12
    single = run_approx_sweep(TrainingParams(Nt=2, Nr=8, L=2, T=12), [0.0, 10.0], [5, 10], 3, batch=2,
                              device=torch.device("cpu"), solve_fn=_oracle_alg12, builder=_torch_builder()).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_approx, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(q.get(timeout=500) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(res[0], res[1])
    np.testing.assert_allclose(res[0], single, rtol=1e-12)
    assert single.shape == (2, 2, 2)
