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


def test_two_phase_device_call_returns_before_the_solve_and_equals_the_one_phase_call():
    """jstsp_proposed_algorithm_begin_c32 / _end (include/jstsp.h): _begin enqueues the solve and returns while the GPU works -
    measured here as: the host gets control back in a fraction of the solve's duration; _end delivers bit for bit the outputs
    of the one-phase call; with every trial forced through the recovery (JSTSP_FUSED_KBACK=-20) _end re-solves them."""
    import os
    import time
    import torch
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
    inp = build_trials(p, 0, 64, seed=9, device=torch.device("cuda:0"))
    hyp = [inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")]
    args = (inp["subY"], inp["Omega"], inp["A"], inp["B"], 100, *hyp, "approximate")
    ref = J.proposed_algorithm(*args)                      # (also grows the workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ref = J.proposed_algorithm(*args); torch.cuda.synchronize(); t_full = time.perf_counter() - t0
    t0 = time.perf_counter(); h = J.proposed_algorithm_begin(*args); t_begin = time.perf_counter() - t0
    out = h.end()
    torch.cuda.synchronize()
    assert h.fallbacks == 0
    assert t_begin < 0.5 * t_full, (t_begin, t_full)       # measured: about 0.1 (setup with its two probe reads, then enqueue only)
    for a, b in zip(ref, out):
        assert torch.equal(torch.view_as_real(a) if a.is_complex() else a, torch.view_as_real(b) if b.is_complex() else b)
    with pytest.raises(J.JstspError):
        h.end()
    os.environ["JSTSP_FUSED_KBACK"] = "-20"
    try:
        h2 = J.proposed_algorithm_begin(*args)
        rec = h2.end()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("JSTSP_FUSED_KBACK", None)
    assert h2.fallbacks == 64
    os.environ["JSTSP_FUSED"] = "0"
    try:
        three = J.proposed_algorithm(*args)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("JSTSP_FUSED", None)
    for a, b in zip(three, rec):
        assert torch.equal(torch.view_as_real(a) if a.is_complex() else a, torch.view_as_real(b) if b.is_complex() else b)
