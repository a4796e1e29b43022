"""Two host threads, two contexts, two streams of one GPU at the same time: the library keeps its per-call state in the context
(workspace, streams, events) and its per-call switches in a thread-local struct (csrc/runtime.hip: load_tuning), so concurrent
solves must return exactly what the same calls return one after the other."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_solves_on_two_contexts_equal_the_serial_ones():
    import torch
    import jstsp19_amd as J
    from jstsp19_amd import _lib
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
    inp = build_trials(p, 0, 64, seed=5, device=torch.device("cuda:0"))
    hyp = [inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")]
    torch.cuda.synchronize()
    halves = (slice(0, 32), slice(32, 64))

    def solve(sl, ctx, stream, out):
        with torch.cuda.stream(stream):
            r = J.proposed_algorithm(inp["subY"][sl], inp["Omega"][sl], inp["A"], inp["B"][sl], 25, *[h[sl] for h in hyp], "approximate", ctx=ctx)
            stream.synchronize()
        out.extend(r)

    ctxs = [_lib.Context(0), _lib.Context(0)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    serial = [[], []]
    for k in range(2):
        solve(halves[k], ctxs[k], streams[k], serial[k])
    for rep in range(3):
        conc = [[], []]
        th = [threading.Thread(target=solve, args=(halves[k], ctxs[k], streams[k], conc[k])) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in range(2):
            assert len(conc[k]) == 3
            for a, b in zip(serial[k], conc[k]):
                assert torch.equal(torch.view_as_real(a) if a.is_complex() else a, torch.view_as_real(b) if b.is_complex() else b)
    assert np.isfinite(serial[0][0].abs().sum().item()) and serial[0][0].abs().sum().item() > 0
