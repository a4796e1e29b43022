"""mex/jstsp_mex.cpp compiled by g++ against the first-party stand-in for mex.h (tests/mex_stub/) and driven through
`mexFunction` from here: argument checks, batch handling (third dimension), output creation (complex matrices, the 1 x m
cell of OMP, convergence_error only for nargout >= 3) and the error route through mexErrMsgIdAndTxt.  This checks the
gateway's own logic - MATLAB is not in the image and the stand-in pins nothing about MATLAB itself.

CPU part: compile + every error that is raised before a GPU is needed (+ a failing jstsp_create when there is no GPU).
GPU part (-m gpu): the calls of the reference's drivers against the float64 oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLE, INT32, CHAR, CELL = 6, 12, 4, 1


@pytest.fixture(scope="module")
def mex(tmp_path_factory):
    from jstsp19_amd import build as B
    lib = B.build()
    out = str(tmp_path_factory.mktemp("mexstub") / "jstsp_mex_stub.so")
    cmd = ["g++", "-O1", "-Wall", "-Wextra", "-Werror", "-std=c++17", "-shared", "-fPIC", "-DMATLAB_MEX_FILE",
           "-I" + os.path.join(ROOT, "tests", "mex_stub"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "mex", "jstsp_mex.cpp"), os.path.join(ROOT, "tests", "mex_stub", "stub.cpp"), "-o", out,
           "-L" + os.path.dirname(lib), "-ljstsp_mi355x", "-Wl,-rpath," + os.path.dirname(lib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import torch  # noqa: F401  (one HIP runtime per process: torch's copy first, as jstsp19_amd._lib does)
    m = C.CDLL(out)
    vp = C.c_void_p
    m.mxCreateNumericArray.restype = vp
    m.mxCreateNumericArray.argtypes = [C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_int]
    m.mxCreateString.restype = vp
    m.mxCreateString.argtypes = [C.c_char_p]
    m.mxCreateDoubleScalar.restype = vp
    m.mxCreateDoubleScalar.argtypes = [C.c_double]
    m.mxGetData.restype = vp
    m.mxGetData.argtypes = [vp]
    m.mxGetCell.restype = vp
    m.mxGetCell.argtypes = [vp, C.c_size_t]
    m.mxGetNumberOfDimensions.restype = C.c_size_t
    m.mxGetNumberOfDimensions.argtypes = [vp]
    m.mxGetDimensions.restype = C.POINTER(C.c_size_t)
    m.mxGetDimensions.argtypes = [vp]
    m.mxIsComplex.argtypes = [vp]
    m.mxGetClassID.argtypes = [vp]
    m.mxGetScalar.restype = C.c_double
    m.mxGetScalar.argtypes = [vp]
    m.stub_call.argtypes = [C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp)]
    m.stub_error_id.restype = C.c_char_p
    m.stub_error_message.restype = C.c_char_p
    yield m
    m.stub_run_at_exit()


class MexError(RuntimeError):
    def __init__(self, ident, msg):
        super().__init__("%s: %s" % (ident, msg))
        self.ident = ident


def to_mx(m, x):
    """python value -> mxArray* (str: char row; scalar: double scalar; ndarray (batch-last MATLAB order): double / complex)"""
    if isinstance(x, str):
        return m.mxCreateString(x.encode())
    if np.isscalar(x):
        return m.mxCreateDoubleScalar(float(x))
    x = np.asarray(x)
    cplx = np.iscomplexobj(x)
    dims = (C.c_size_t * max(x.ndim, 2))(*(list(x.shape) + [1] * (2 - x.ndim)))
    a = m.mxCreateNumericArray(max(x.ndim, 2), dims, DOUBLE, 1 if cplx else 0)
    buf = np.ascontiguousarray(x.astype(np.complex128 if cplx else np.float64).reshape(-1, order="F"))
    C.memmove(m.mxGetData(a), buf.ctypes.data, buf.nbytes)
    return a


def from_mx(m, a):
    nd = m.mxGetNumberOfDimensions(a)
    shape = [m.mxGetDimensions(a)[i] for i in range(nd)]
    cls = m.mxGetClassID(a)
    n = int(np.prod(shape))
    if cls == CELL:
        return [from_mx(m, m.mxGetCell(a, i)) for i in range(n)]
    dt = {DOUBLE: np.complex128 if m.mxIsComplex(a) else np.float64, INT32: np.int32}[cls]
    buf = np.empty(n, dtype=dt)
    C.memmove(buf.ctypes.data, m.mxGetData(a), buf.nbytes)
    return buf.reshape(shape, order="F")


def call(m, nlhs, *args):
    prhs = (C.c_void_p * len(args))(*[to_mx(m, a) for a in args])
    plhs = (C.c_void_p * max(nlhs, 1))()
    if m.stub_call(nlhs, plhs, len(args), prhs):
        raise MexError(m.stub_error_id().decode(), m.stub_error_message().decode())
    return [from_mx(m, plhs[i]) for i in range(max(nlhs, 1))]


def _problem(rng, N=12, M=20, Gr=9, G2=10, batch=1):
    c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    A, B = c(N, Gr) / np.sqrt(N), c(G2, M) / np.sqrt(G2)
    S0 = np.zeros((Gr, G2, batch), complex)
    for t in range(batch):
        S0[:, :, t].flat[rng.choice(Gr * G2, 4, replace=False)] = 3 * c(4)
    Om = (rng.random((N, M, batch)) < 0.5).astype(float)
    subY = Om * (np.stack([A @ S0[:, :, t] @ B for t in range(batch)], axis=2) + 0.05 * c(N, M, batch))
    tY = 1.0 / np.sum(np.abs(subY) ** 2, axis=(0, 1))
    return subY, Om, A, B, tY, S0


def test_gateway_compiles_and_rejects_bad_calls_before_touching_the_gpu(mex):
    rng = np.random.default_rng(1)
    subY, Om, A, B, tY, _ = _problem(rng)
    with pytest.raises(MexError) as e:
        call(mex, 1, 3.0)
    assert e.value.ident == "jstsp:args" and "function name" in str(e.value)
    with pytest.raises(MexError) as e:
        call(mex, 1, "no_such_solver", subY)
    assert e.value.ident == "jstsp:args" and "no_such_solver" in str(e.value)
    with pytest.raises(MexError) as e:                                      # proposed_algorithm.m:1 has nine inputs
        call(mex, 1, "proposed_algorithm", subY[:, :, 0], Om[:, :, 0], A, B)
    assert e.value.ident == "jstsp:args" and "9 to 10" in str(e.value)
    with pytest.raises(MexError) as e:                                      # four outputs asked of a three-output function
        call(mex, 4, "proposed_algorithm", subY[:, :, 0], Om[:, :, 0], A, B, 5, 1.0, 1.0, 0.2, "approximate")
    assert e.value.ident == "jstsp:args" and "output" in str(e.value)
    with pytest.raises(MexError) as e:                                      # size(A,1) ~= size(subY,1)
        call(mex, 1, "proposed_algorithm", subY[:, :, 0], Om[:, :, 0], A[:-1], B, 5, 1.0, 1.0, 0.2, "approximate")
    assert e.value.ident == "jstsp:shape"
    with pytest.raises(MexError) as e:                                      # tau_Y with a wrong number of entries
        call(mex, 1, "proposed_algorithm", subY[:, :, 0], Om[:, :, 0], A, B, 5, np.ones(3), 1.0, 0.2, "approximate")
    assert e.value.ident == "jstsp:args" and "tau_Y" in str(e.value)
    with pytest.raises(MexError) as e:
        call(mex, 1, "svt", subY[:, :, 0])                                  # svt.m:1 has two inputs
    assert e.value.ident == "jstsp:args"
    with pytest.raises(MexError) as e:
        call(mex, 1, "OMP", A, subY[:5, 0, 0], 3, 0)
    assert e.value.ident == "jstsp:shape"
    import torch
    if not torch.cuda.is_available():                                       # no GPU: jstsp_create fails, loudly, by the same route
        with pytest.raises(MexError) as e:
            call(mex, 1, "svt", subY[:, :, 0], 0.1)
        assert e.value.ident == "jstsp:call" and "jstsp_create" in str(e.value)


@pytest.mark.gpu
def test_gateway_proposed_algorithm_outputs_and_batch(mex):
    """[S], [S, Y], [S, Y, convergence_error] of proposed_algorithm.m:1 and the 10-argument _angles form, unbatched and with
    the realisations stacked along a third dimension (per-problem tau_Y, scalar tau_S / rho), against the float64 oracle."""
    from oracle import solvers as O
    rng = np.random.default_rng(2)
    subY, Om, A, B, tY, S0 = _problem(rng, batch=3)
    Imax, tS, rho = 25, 0.02, 0.2
    ref = [O.proposed_algorithm(subY[:, :, t], Om[:, :, t], A, B, Imax, tY[t], tS, rho, "approximate") for t in range(3)]
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    (S,) = call(mex, 1, "proposed_algorithm", subY[:, :, 0], Om[:, :, 0], A, B, Imax, tY[0], tS, rho, "approximate")
    assert S.shape == (9, 10) and rel(S, ref[0][0]) < 2e-4
    S, Y = call(mex, 2, "proposed_algorithm", subY[:, :, 0], Om[:, :, 0], A, B, Imax, tY[0], tS, rho, "approximate")
    assert Y.shape == (12, 20) and rel(Y, ref[0][1]) < 2e-4
    S, Y, ce = call(mex, 3, "proposed_algorithm", subY, Om, A, B, Imax, tY, tS, rho, "approximate")
    assert S.shape == (9, 10, 3) and Y.shape == (12, 20, 3) and ce.shape == (Imax, 3, 3)
    for t in range(3):
        assert rel(S[:, :, t], ref[t][0]) < 2e-4 and rel(Y[:, :, t], ref[t][1]) < 2e-4
        assert np.isinf(ce[0, 2, t])
        np.testing.assert_allclose(ce[1:, :, t], ref[t][2][1:], rtol=2e-3)
    # 'std' (anything but 'approximate', proposed_algorithm.m:45-54) needs full column rank: N >= Gr, M >= G2 holds here
    (Sstd,) = call(mex, 1, "proposed_algorithm", subY[:, :, 1], Om[:, :, 1], A, B, 10, tY[1], tS, rho, "std")
    assert rel(Sstd, O.proposed_algorithm(subY[:, :, 1], Om[:, :, 1], A, B, 10, tY[1], tS, rho, "std")[0]) < 5e-4
    # proposed_algorithm_angles.m:1 - indx_S as MATLAB doubles, per-problem dictionaries B (3-D)
    idx = np.stack([np.argsort(-np.abs(S0[:, :, t]).reshape(-1, order="F"), kind="stable") + 1.0 for t in range(3)], axis=1)
    B3 = np.stack([B, 1.1 * B, 0.9 * B], axis=2)
    (Sa,) = call(mex, 1, "proposed_algorithm", subY, Om, A, B3, Imax, tY, tS, rho, "approximate", idx)
    for t in range(3):
        Sr = O.proposed_algorithm(subY[:, :, t], Om[:, :, t], A, B3[:, :, t], Imax, tY[t], tS, rho, "approximate",
                                  indx_S=idx[:, t].astype(np.int64))[0]
        assert rel(Sa[:, :, t], Sr) < 2e-4
    # a real-valued dictionary arrives as a REAL mxArray
    (Sr_,) = call(mex, 1, "proposed_algorithm", subY[:, :, 0], Om[:, :, 0], A.real, B, 8, tY[0], tS, rho, "approximate")
    assert rel(Sr_, O.proposed_algorithm(subY[:, :, 0], Om[:, :, 0], A.real, B, 8, tY[0], tS, rho, "approximate")[0]) < 2e-4
    # a library error (here: 'std' without full column rank) comes back through mexErrMsgIdAndTxt with the library's message
    with pytest.raises(MexError) as e:
        call(mex, 1, "proposed_algorithm", subY[:5, :, 0], Om[:5, :, 0], A[:5], B, 5, tY[0], tS, rho, "std")
    assert e.value.ident == "jstsp:call" and "full column rank" in str(e.value)


@pytest.mark.gpu
def test_gateway_omp_svt_and_the_baselines(mex):
    from oracle import solvers as O
    rng = np.random.default_rng(3)
    c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    # [x_hat, indexSet, v, targetMatrix] = OMP(A, v, m, snr)   OMP.m:1 - indexSet is a 1 x m cell
    meas, size_d, m_ = 24, 40, 3
    D = np.exp(-2j * np.pi * np.outer(np.arange(meas), np.arange(size_d)) / size_d) / np.sqrt(meas)
    x0 = np.zeros(size_d, complex); x0[[3, 17, 29]] = [2, -1.5j, 1 + 1j]
    v = D @ x0
    x, cell, v_out, T = call(mex, 4, "OMP", D, v, m_, 10.0)
    xo, ido, _, To = O.omp(D, v, m_)
    assert isinstance(cell, list) and len(cell) == m_
    assert [int(ci.reshape(-1)[0]) for ci in cell] == [int(i) for i in ido]
    assert rel(x.reshape(-1), xo) < 2e-4 and rel(T, To) < 1e-5 and np.allclose(v_out.reshape(-1), v)
    # X = svt(Y, tau), also with pages
    Y3 = c(10, 14, 2)
    (X,) = call(mex, 1, "svt", Y3, 1.5)
    for t in range(2):
        assert rel(X[:, :, t], O.svt(Y3[:, :, t], 1.5)) < 2e-5
    # mc_svt / mc_admm / sparse_admm signatures
    Om = (rng.random((10, 14)) < 0.6).astype(float)
    H = c(10, 2) @ c(2, 14)
    (Xm,) = call(mex, 1, "mc_svt", Om * H, Om, 15, 0.2, 0.2)
    assert rel(Xm, O.mc_svt(Om * H, Om, 15, 0.2, 0.2)) < 2e-4
    Xa, cea = call(mex, 2, "mc_admm", H, Om * H, Om, 15, 0.2, 0.2)
    Xo, ceo = O.mc_admm(H, Om * H, Om, 15, 0.2, 0.2)
    assert rel(Xa, Xo) < 2e-4 and cea.shape == (15, 1)
    np.testing.assert_allclose(cea.reshape(-1), ceo, rtol=5e-3)
    n = 8
    Dn = np.exp(-2j * np.pi * np.outer(np.arange(n), np.arange(n)) / n) / np.sqrt(n)
    Hs = Dn @ (np.eye(n)[:, [2]] @ np.eye(n)[[5], :] * (1 + 2j)) @ Dn.conj().T
    Ss, ces = call(mex, 2, "sparse_admm", Hs, Hs + 0.01 * c(n, n), Dn, Dn, 20)
    So, co = O.sparse_admm(Hs, Hs + 0.0, Dn, Dn, 20)
    assert Ss.shape == (n, n) and ces.shape == (20, 1)
    # LS baseline and the capped spectral NMSE of the drivers
    A, B = c(12, 9), c(10, 20)
    Yl = A @ c(9, 10) @ B
    (Sl,) = call(mex, 1, "ls", Yl, A, B)
    assert rel(Sl, np.linalg.pinv(A) @ Yl @ np.linalg.pinv(B)) < 1e-3
    (nm,) = call(mex, 1, "nmse", Sl, 1.3 * Sl)
    assert abs(float(nm.reshape(-1)[0]) - O.nmse_capped(Sl, 1.3 * Sl)) < 1e-5


@pytest.mark.gpu
def test_gateway_vamp_with_the_drivers_dense_512_dictionary(mex):
    """plot_errorVSsnr.m:73-80,100 through the gateway, as the driver writes it: ``Phi = kron((B*B').', A)`` (512 x 512),
    ``y = vec(Y*B')``, ``vamp(y, Phi, 1, numOfnz)``.  The gateway always runs the reference's 100 iterations, where VAMP in this
    configuration is chaotic, so what is checked here is the plumbing - shapes, finiteness, agreement of
    the dense call with the factored ``vamp_kron`` command in estimation quality; tests/test_gpu_baselines.py holds the
    per-iteration comparison with the float64 restatement."""
    from oracle import solvers as O
    rng = np.random.default_rng(12)
    c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    Nr, Gr, G2, Th = 32, 32, 16, 16
    A = c(Nr, Gr) / np.sqrt(Nr)
    B = c(G2, Th) / np.sqrt(Th)
    Z = np.zeros((Gr, G2), complex)
    Z.flat[rng.choice(Gr * G2, 6, replace=False)] = 4 * c(6)
    Y = A @ Z @ B + 0.05 * c(Nr, Th)
    Gb = B @ B.conj().T
    Phi = np.kron(Gb.T, A)                                                 # :79
    y = (Y @ B.conj().T).flatten("F")                                      # :80
    assert Phi.shape == (512, 512)
    (x,) = call(mex, 1, "vamp", y, Phi, 1.0, 100)                          # :100
    assert x.shape in ((512, 1), (512,)) and np.all(np.isfinite(x))
    (Xk,) = call(mex, 1, "vamp_kron", Y @ B.conj().T, A, Gb, 1.0, 100)
    e_d = O.nmse_capped(x.reshape(Gr, G2, order="F"), Z)
    e_k = O.nmse_capped(Xk, Z)
    assert abs(e_d - e_k) < 0.25 * max(e_k, 0.05) + 0.05, (e_d, e_k)
