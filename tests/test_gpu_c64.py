"""The _c64 entry points of include/jstsp.h (MATLAB's own element type at the boundary): called through ctypes with
float64 numpy arrays / float64 device tensors.  Two checks each: (1) against the float64 oracle on genuinely double
inputs, to the fp32 tolerance of the path; (2) bit-for-bit against the _c32 entry point when the doubles are
float-representable (narrowing and widening are then exact, so any difference is a staging bug)."""
import ctypes as C

import numpy as np
import pytest

import jstsp19_amd
from jstsp19_amd import _lib
from conftest import rel_err

pytestmark = pytest.mark.gpu
HOST, DEVICE = 0, 1


def _f(a):                       # column-major bytes of a (batch, R, C) / (R, C) array: trial index slowest
    a = np.asarray(a)
    if a.ndim == 3:
        return np.ascontiguousarray(np.transpose(a, (0, 2, 1)))
    return np.ascontiguousarray(a.T)


def _unf(buf, shape):            # inverse of _f
    if len(shape) == 3:
        b, R, Cc = shape
        return np.transpose(buf.reshape(b, Cc, R), (0, 2, 1))
    R, Cc = shape
    return buf.reshape(Cc, R).T


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _problem(rng, batch, N, M, Gr, G2, dtype):
    A = (rng.standard_normal((N, Gr)) + 1j * rng.standard_normal((N, Gr))) / np.sqrt(2 * N)
    B = (rng.standard_normal((batch, G2, M)) + 1j * rng.standard_normal((batch, G2, M))) / np.sqrt(2 * G2)
    Om = (rng.random((batch, N, M)) < 0.4).astype(np.float64)
    S0 = np.zeros((batch, Gr, G2), complex)
    for t in range(batch):
        ix = rng.choice(Gr * G2, 6, replace=False)
        S0[t].flat[ix] = rng.standard_normal(6) + 1j * rng.standard_normal(6)
    Y = np.einsum("ng,tgh,thm->tnm", A, S0, B) + 0.05 * (rng.standard_normal((batch, N, M)) + 1j * rng.standard_normal((batch, N, M)))
    subY = Om * Y
    if dtype == np.complex64:     # float-representable doubles
        A, B, subY = (x.astype(np.complex64).astype(complex) for x in (A, B, subY))
    return A, B, Om, subY


def _proposed(lib, ctx, suffix, A, B, Om, subY, Imax, tY, tS, rho, type_, indx=None, want_ce=True):
    batch, N, M = subY.shape
    Gr, G2 = A.shape[1], B.shape[1]
    cdt, rdt = (np.complex128, np.float64) if suffix == "c64" else (np.complex64, np.float32)
    a, b, om, sy = _f(A.astype(cdt)), _f(B.astype(cdt)), _f(Om.astype(rdt)), _f(subY.astype(cdt))
    S = np.empty(batch * Gr * G2, cdt)
    Y = np.empty(batch * N * M, cdt)
    ce = np.empty(batch * 3 * Imax, np.float64) if want_ce else None
    ty, ts, rh = (np.full(batch, v, np.float64) for v in (tY, tS, rho))
    ix = np.ascontiguousarray(indx, np.int32) if indx is not None else None
    fn = getattr(lib, "jstsp_proposed_algorithm_" + suffix)
    _lib.check(fn(ctx.handle, N, M, Gr, G2, batch, _p(sy), _p(om), _p(a), 0, _p(b), G2 * M, Imax, _dp(ty), _dp(ts), _dp(rh),
                  type_, _p(ix), _p(S), _p(Y), _p(ce), HOST), "proposed_" + suffix)
    return (_unf(S, (batch, Gr, G2)), _unf(Y, (batch, N, M)),
            np.transpose(ce.reshape(batch, 3, Imax), (0, 2, 1)) if want_ce else None)


@pytest.mark.parametrize("type_", [0, 1])
def test_proposed_algorithm_c64(type_):
    from oracle import solvers as O
    lib, ctx = jstsp19_amd.load(), jstsp19_amd.Context(0)
    rng = np.random.default_rng(31 + type_)
    batch, N, M, Gr, G2, Imax = 3, 16, 48, 16, 24, 20
    A, B, Om, subY = _problem(rng, batch, N, M, Gr, G2, np.complex128)
    S, Y, ce = _proposed(lib, ctx, "c64", A, B, Om, subY, Imax, 0.02, 0.01, 0.4, type_)
    assert S.dtype == np.complex128 and ce.dtype == np.float64
    for t in range(batch):
        So, Yo, ceo = O.proposed_algorithm(subY[t], Om[t], A, B[t], Imax, 0.02, 0.01, 0.4, "approximate" if type_ == 0 else "std")
        assert rel_err(S[t], So) < 3e-4 and rel_err(Y[t], Yo) < 3e-4
        np.testing.assert_allclose(ce[t][1:], ceo[1:], rtol=3e-3)
    # float-representable inputs: identical to the _c32 entry point, bit for bit
    A, B, Om, subY = _problem(rng, batch, N, M, Gr, G2, np.complex64)
    indx = np.stack([rng.permutation(Gr * G2) + 1 for _ in range(batch)]).astype(np.int32)
    for ix in (None, indx):
        S64, Y64, ce64 = _proposed(lib, ctx, "c64", A, B, Om, subY, Imax, 0.02, 0.01, 0.4, type_, ix)
        S32, Y32, ce32 = _proposed(lib, ctx, "c32", A, B, Om, subY, Imax, 0.02, 0.01, 0.4, type_, ix)
        assert np.array_equal(S64, S32.astype(complex)) and np.array_equal(Y64, Y32.astype(complex))
        assert np.array_equal(ce64, ce32, equal_nan=True)
    # outputs that are not wanted
    S64b, _, _ = _proposed(lib, ctx, "c64", A, B, Om, subY, Imax, 0.02, 0.01, 0.4, type_, indx, want_ce=False)
    assert np.array_equal(S64b, S64)


def test_kernel_level_and_svt_c64():
    from oracle import solvers as O
    lib, ctx = jstsp19_amd.load(), jstsp19_amd.Context(0)
    rng = np.random.default_rng(5)
    batch, N, M, Gr, G2 = 4, 12, 40, 10, 18
    A, B, _, K = _problem(rng, batch, N, M, Gr, G2, np.complex128)
    a, b, k = _f(A), _f(B), _f(K)
    out = np.empty(batch * Gr * G2, complex)
    _lib.check(lib.jstsp_correlate_c64(ctx.handle, N, M, Gr, G2, batch, _p(k), _p(a), 0, _p(b), G2 * M, _p(out), HOST))
    ref = np.einsum("ng,tnm,thm->tgh", A.conj(), K, B.conj())
    assert rel_err(_unf(out, (batch, Gr, G2)), ref) < 2e-6
    S = rng.standard_normal((batch, Gr, G2)) + 1j * rng.standard_normal((batch, Gr, G2))
    out2 = np.empty(batch * N * M, complex)
    _lib.check(lib.jstsp_synthesize_c64(ctx.handle, N, M, Gr, G2, batch, _p(_f(S)), _p(a), 0, _p(b), G2 * M, _p(out2), HOST))
    assert rel_err(_unf(out2, (batch, N, M)), np.einsum("ng,tgh,thm->tnm", A, S, B)) < 2e-6
    tau = np.full(batch, 0.3)
    X = np.empty(batch * N * M, complex)
    _lib.check(lib.jstsp_svt_c64(ctx.handle, N, M, batch, _p(k), _dp(tau), _p(X), HOST))
    X = _unf(X, (batch, N, M))
    for t in range(batch):
        assert rel_err(X[t], O.svt(K[t], 0.3)) < 2e-5
    # NULL / shape errors come back as codes, not crashes
    assert lib.jstsp_svt_c64(ctx.handle, N, M, batch, None, _dp(tau), _p(X), HOST) == -1
    assert lib.jstsp_svt_c64(ctx.handle, 0, M, batch, _p(k), _dp(tau), _p(X), HOST) == -2
    assert lib.jstsp_svt_c64(ctx.handle, N, M, batch, _p(k), _dp(tau), _p(X), 7) == -4


def test_benchmark_algorithms_c64():
    from oracle import solvers as O
    lib, ctx = jstsp19_amd.load(), jstsp19_amd.Context(0)
    rng = np.random.default_rng(9)
    # OMP.m
    meas, size_d, batch, m = 24, 40, 3, 5
    Ad = (rng.standard_normal((meas, size_d)) + 1j * rng.standard_normal((meas, size_d))) / np.sqrt(2 * meas)
    x0 = np.zeros((batch, size_d), complex)
    for t in range(batch):
        x0[t, rng.choice(size_d, m, replace=False)] = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    v = x0 @ Ad.T + 0.01 * rng.standard_normal((batch, meas))
    xh = np.empty(batch * size_d, complex)
    idx = np.empty(batch * m, np.int32)
    tgt = np.empty(batch * meas * m, complex)
    _lib.check(lib.jstsp_omp_c64(ctx.handle, meas, size_d, batch, _p(_f(Ad)), 0, _p(np.ascontiguousarray(v)), m, _p(xh), _p(idx),
                                 _p(tgt), HOST))
    for t in range(batch):
        xo, io, _, To = O.omp_literal(Ad, v[t], m)
        assert np.array_equal(idx.reshape(batch, m)[t], io)
        assert rel_err(xh.reshape(batch, size_d)[t], xo) < 1e-4
        assert rel_err(_unf(tgt, (batch, meas, m))[t], To) < 1e-6
    # mc_svt.m, mc_admm.m, sparse_admm.m
    Mr = Mt = 16
    H = (rng.standard_normal((batch, Mr, 3)) + 1j * rng.standard_normal((batch, Mr, 3))) @ \
        (rng.standard_normal((batch, 3, Mt)) + 1j * rng.standard_normal((batch, 3, Mt)))
    Om = (rng.random((batch, Mr, Mt)) < 0.6).astype(float)
    OH = Om * H
    tau, rho = np.full(batch, 0.5), np.full(batch, 0.3)
    X = np.empty(batch * Mr * Mt, complex)
    _lib.check(lib.jstsp_mc_svt_c64(ctx.handle, Mr, Mt, batch, _p(_f(OH)), _p(_f(Om)), 15, _dp(tau), _dp(rho), _p(X), HOST))
    for t in range(batch):
        assert rel_err(_unf(X, (batch, Mr, Mt))[t], O.mc_svt(OH[t], Om[t], 15, 0.5, 0.3)) < 1e-4
    ce = np.empty(batch * 15)
    _lib.check(lib.jstsp_mc_admm_c64(ctx.handle, Mr, Mt, batch, _p(_f(H)), _p(_f(OH)), _p(_f(Om)), 15, _dp(tau), _dp(rho), _p(X),
                                     _p(ce), HOST))
    for t in range(batch):
        Xo, ceo = O.mc_admm(H[t], OH[t], Om[t], 15, 0.5, 0.3)
        assert rel_err(_unf(X, (batch, Mr, Mt))[t], Xo) < 1e-4
        np.testing.assert_allclose(ce.reshape(batch, 15)[t], np.ravel(ceo), rtol=2e-3)
    F = np.fft.fft(np.eye(Mr)) / np.sqrt(Mr)
    S = np.empty(batch * Mr * Mt, complex)
    _lib.check(lib.jstsp_sparse_admm_c64(ctx.handle, Mr, Mt, Mr, Mt, batch, _p(_f(H)), _p(_f(OH)), _p(_f(F)), _p(_f(F)), 12, _p(S),
                                         _p(ce[:batch * 12]), HOST))
    for t in range(batch):
        So, ceo = O.sparse_admm(H[t], OH[t], F, F, 12)
        assert rel_err(_unf(S, (batch, Mr, Mt))[t], So) < 2e-4
    # vamp.m
    Mv, Nv = 24, 48
    Av = (rng.standard_normal((Mv, Nv)) + 1j * rng.standard_normal((Mv, Nv))) / np.sqrt(2 * Mv)
    xs = np.zeros((batch, Nv), complex)
    for t in range(batch):
        xs[t, rng.choice(Nv, 4, replace=False)] = 3 * (rng.standard_normal(4) + 1j * rng.standard_normal(4))
    yv = xs @ Av.T + 0.05 * (rng.standard_normal((batch, Mv)) + 1j * rng.standard_normal((batch, Mv)))
    from oracle import vamp as V
    xo = np.empty(batch * Nv, complex)
    _lib.check(lib.jstsp_vamp_c64(ctx.handle, Mv, Nv, batch, _p(np.ascontiguousarray(yv)), _p(_f(Av)), 0, 1.0, 4.0, 5, _p(xo), HOST))
    for t in range(batch):
        assert rel_err(xo.reshape(batch, Nv)[t], V.vamp_literal(yv[t], Av, 1.0, 4.0, nit=5)) < 1e-10      # (float64 on the device since round 6)
    # the reference's operating point, nit = 100: the _c64 entry computes in float64 (csrc/vamp64.hip) and follows the literal
    # float64 restatement per trial; the _c32 entry on the same (float-representable) inputs agrees statistically only
    _lib.check(lib.jstsp_vamp_c64(ctx.handle, Mv, Nv, batch, _p(np.ascontiguousarray(yv)), _p(_f(Av)), 0, 1.0, 4.0, 100, _p(xo), HOST))
    for t in range(batch):
        assert rel_err(xo.reshape(batch, Nv)[t], V.vamp_literal(yv[t], Av, 1.0, 4.0, nit=100)) < 1e-8      # (this small system is not chaotic)


def test_c64_device_memory_stays_asynchronous_and_matches_host():
    import torch
    lib, ctx = jstsp19_amd.load(), jstsp19_amd.Context(0)
    rng = np.random.default_rng(77)
    batch, N, M, Gr, G2, Imax = 2, 16, 48, 16, 24, 10
    A, B, Om, subY = _problem(rng, batch, N, M, Gr, G2, np.complex128)
    S_h, Y_h, ce_h = _proposed(lib, ctx, "c64", A, B, Om, subY, Imax, 0.02, 0.01, 0.4, 0)
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(dev)
    _lib.check(lib.jstsp_set_stream(ctx.handle, C.c_void_p(stream.cuda_stream)))
    with torch.cuda.stream(stream):
        t = lambda a: torch.from_numpy(_f(a)).to(dev)
        a, b, om, sy = t(A), t(B), t(Om), t(subY)
        S = torch.empty(batch * Gr * G2, dtype=torch.complex128, device=dev)
        Y = torch.empty(batch * N * M, dtype=torch.complex128, device=dev)
        ce = torch.empty(batch * 3 * Imax, dtype=torch.float64, device=dev)
        ty, ts, rh = (np.full(batch, v) for v in (0.02, 0.01, 0.4))
        _lib.check(lib.jstsp_proposed_algorithm_c64(ctx.handle, N, M, Gr, G2, batch, sy.data_ptr(), om.data_ptr(), a.data_ptr(), 0,
                                                    b.data_ptr(), G2 * M, Imax, _dp(ty), _dp(ts), _dp(rh), 0, None, S.data_ptr(),
                                                    Y.data_ptr(), ce.data_ptr(), DEVICE))
    stream.synchronize()
    _lib.check(lib.jstsp_use_own_stream(ctx.handle))
    assert np.array_equal(_unf(S.cpu().numpy(), (batch, Gr, G2)), S_h)
    assert np.array_equal(_unf(Y.cpu().numpy(), (batch, N, M)), Y_h)
    assert np.array_equal(np.transpose(ce.cpu().numpy().reshape(batch, 3, Imax), (0, 2, 1)), ce_h, equal_nan=True)
