"""Host-side mirror of the reference's solver functions on top of the C ABI.

Same names, argument order and meaning as the MATLAB functions they replace:

    [S, Y, convergence_error] = proposed_algorithm(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type)
    [S, Y, convergence_error] = proposed_algorithm_angles(subY, Omega, indx_S, A, B, Imax, tau_Y, tau_S, rho, type, greedy_nnz)
    [x_hat, indexSet, v, targetMatrix] = OMP(A, v, m, snr)
    [S, convergence_error] = sparse_admm(Htrue, OH, Dr, Dt, Imax)
    X = svt(Y, tau);  X = mc_svt(OH, Omega, Imax, tau, rho);  [X, ce] = mc_admm(Htrue, OH, Omega, Imax, tau, rho)

Array arguments are numpy arrays (host: the library copies over PCIe) or torch CUDA tensors
(device-resident, asynchronous on torch's current stream).  A leading batch dimension stacks
independent problems: ``subY`` is (N, M) or (batch, N, M); a 2-D ``A``/``B`` with batched
``subY`` means one dictionary shared by the batch.  The C ABI is column-major; numpy inputs
are re-laid-out here, torch inputs must already be column-major per problem
(``colmajor(t)``: stride (R*C, 1, R)) so that nothing is copied on the device.

All compute happens in libjstsp_mi355x.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DEVICE, HOST, JstspError, check

__all__ = ["proposed_algorithm", "proposed_algorithm_angles", "svt", "mc_svt", "mc_admm", "OMP", "omp_kron",
           "sparse_admm", "vamp", "vamp_kron", "sparse_sca_estim", "cawgn_estim_out", "ls_estimate", "pinv", "mmv_omp", "tssr", "rate", "correlate", "synthesize", "gradient_head", "nmse_spectral", "colmajor",
           "empty_colmajor"]


# ----------------------------------------------------------------------------- array plumbing
def _is_torch(x):
    return type(x).__module__.startswith("torch")


def colmajor(t):
    """Return a torch tensor with the same logical shape (..., R, C) whose last two dims are
    stored column-major (stride (..., 1, R)) — the layout the C ABI expects."""
    return t.transpose(-1, -2).contiguous().transpose(-1, -2)


def empty_colmajor(batch, R, C_, dtype, device):
    import torch
    return torch.empty((batch, C_, R), dtype=dtype, device=device).transpose(1, 2)


class _Arg:
    """A matrix argument normalised to (batch, R, C) + pointer in the C ABI's layout."""

    def __init__(self, x, np_dtype, name, allow_none=False):
        self.keep = None
        self.name = name
        if x is None:
            if not allow_none:
                raise ValueError("%s is required" % name)
            self.ptr, self.batch, self.R, self.C, self.torch, self.batched = None, 0, 0, 0, False, False
            return
        self.torch = _is_torch(x)
        self.batched = x.ndim == 3
        if x.ndim not in (2, 3):
            raise ValueError("%s must be 2-D or 3-D (batch first), got shape %s" % (name, tuple(x.shape)))
        if self.torch:
            import torch
            want = {np.complex64: torch.complex64, np.complex128: torch.complex128, np.float32: torch.float32, np.int32: torch.int32}[np_dtype]
            if not x.is_cuda:
                raise ValueError("%s: torch tensors must live on the GPU (numpy arrays use the host path)" % name)
            if x.dtype != want:
                raise ValueError("%s: expected dtype %s, got %s" % (name, want, x.dtype))
            x3 = x if x.ndim == 3 else x.unsqueeze(0)
            b, R, Cc = x3.shape
            ok = x3.stride(1) == 1 and (x3.stride(2) == R or Cc == 1) and (x3.stride(0) == R * Cc or b == 1)
            if not ok:
                raise ValueError("%s: device tensors must be column-major per problem "
                                 "(use jstsp19_amd.colmajor); strides %s" % (name, x3.stride()))
            self.keep = x3
            self.ptr = x3.data_ptr()
            self.device = x3.device
        else:
            x3 = np.asarray(x)
            x3 = x3 if x3.ndim == 3 else x3[None]
            b, R, Cc = x3.shape
            buf = np.ascontiguousarray(np.swapaxes(x3, 1, 2), dtype=np_dtype)     # [t][c][r]
            self.keep = buf
            self.ptr = buf.ctypes.data
        self.batch, self.R, self.C = int(b), int(R), int(Cc)


def _out(kind_torch, batch, R, Cc, np_dtype, device=None):
    """Allocate an output in the C ABI's layout; returns (ptr, finisher) where finisher(squeeze)
    gives the user-facing (batch, R, C) (or (R, C)) array."""
    if kind_torch:
        import torch
        td = {np.complex64: torch.complex64, np.complex128: torch.complex128, np.float32: torch.float32, np.float64: torch.float64,
              np.int32: torch.int32}[np_dtype]
        buf = torch.empty((batch, Cc, R), dtype=td, device=device)
        return buf.data_ptr(), (lambda sq: (buf.transpose(1, 2)[0] if sq else buf.transpose(1, 2)))
    buf = np.empty((batch, Cc, R), dtype=np_dtype)
    return buf.ctypes.data, (lambda sq: (np.swapaxes(buf, 1, 2)[0] if sq else np.swapaxes(buf, 1, 2)))


def _scalars(v, batch, name):
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.size == 1:
        a = np.full(batch, a[0], dtype=np.float64)
    a = np.ascontiguousarray(a)
    if a.size != batch:
        raise ValueError("%s must be a scalar or have one entry per problem (%d), got %d" % (name, batch, a.size))
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _ctx_for(args, ctx):
    tor = [a for a in args if a.ptr is not None and a.torch]
    npy = [a for a in args if a.ptr is not None and not a.torch]
    if tor and npy:
        raise ValueError("mixing numpy (host) and torch CUDA (device) array arguments is not supported")
    if tor:
        dev = tor[0].device.index or 0
        c = ctx or _lib.default_context(dev)
        c.use_torch_stream()
        return c, DEVICE, tor[0].device
    return ctx or _lib.default_context(0), HOST, None


def _shared_stride(arg, rows_cols, batch, name):
    if arg.batched:
        if arg.batch != batch:
            raise ValueError("%s has batch %d, expected %d" % (name, arg.batch, batch))
        return rows_cols
    return 0


# ----------------------------------------------------------------------------- proposed_algorithm
def proposed_algorithm(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type="approximate", *, indx_S=None,
                       want_ce=True, ctx=None):
    """basic_system_functions/proposed_algorithm.m:1 — returns (S, Y, convergence_error).

    ``convergence_error`` is (batch, Imax, 3) float64 (``None`` with ``want_ce=False``, the
    analogue of calling the MATLAB function with fewer than three outputs).
    """
    a_sub = _Arg(subY, np.complex64, "subY")
    a_om = _Arg(Omega, np.float32, "Omega")
    a_A = _Arg(A, np.complex64, "A")
    a_B = _Arg(B, np.complex64, "B")
    a_ix = _Arg(None, np.int32, "indx_S", allow_none=True)
    batch, N, M = a_sub.batch, a_sub.R, a_sub.C
    Gr, G2 = a_A.C, a_B.R
    if (a_om.batch, a_om.R, a_om.C) != (batch, N, M):
        raise ValueError("Omega must have the shape of subY")
    if a_A.R != N or a_B.C != M:
        raise ValueError("size(A,1) must equal size(subY,1) and size(B,2) must equal size(subY,2)")
    if indx_S is not None:
        if _is_torch(indx_S):
            import torch
            ix2 = indx_S.reshape(batch, Gr * G2, 1).to(torch.int32).contiguous()
        else:
            ix2 = np.asarray(indx_S).reshape(batch, Gr * G2, 1).astype(np.int32)
        a_ix = _Arg(ix2, np.int32, "indx_S")
    c, mem, dev = _ctx_for([a_sub, a_om, a_A, a_B, a_ix], ctx)
    sA = _shared_stride(a_A, N * Gr, batch, "A")
    sB = _shared_stride(a_B, G2 * M, batch, "B")
    tY, ptY = _scalars(tau_Y, batch, "tau_Y")
    tS, ptS = _scalars(tau_S, batch, "tau_S")
    rh, prh = _scalars(rho, batch, "rho")
    pS, fS = _out(mem == DEVICE, batch, Gr, G2, np.complex64, dev)
    pY, fY = _out(mem == DEVICE, batch, N, M, np.complex64, dev)
    if want_ce:
        pce, fce = _out(mem == DEVICE, batch, int(Imax), 3, np.float64, dev)
    else:
        pce, fce = None, None
    tcode = _lib.TYPE_APPROXIMATE if type == "approximate" else _lib.TYPE_STD
    rc = c._lib.jstsp_proposed_algorithm_c32(c.handle, N, M, Gr, G2, batch, a_sub.ptr, a_om.ptr, a_A.ptr, sA,
                                             a_B.ptr, sB, int(Imax), ptY, ptS, prh, tcode, a_ix.ptr, pS, pY, pce,
                                             mem)
    check(rc, "jstsp_proposed_algorithm_c32")
    sq = not a_sub.batched
    return fS(sq), fY(sq), (fce(sq) if want_ce else None)


class PendingSolve:
    """Handle of :func:`proposed_algorithm_begin`: the solve is running on the context's stream; :meth:`end` completes it and
    returns ``(S, Y, convergence_error)``.  Keeps every array of the call alive until then."""

    def __init__(self, ctx, handle, keep, result):
        self._ctx, self._h, self._keep, self._result = ctx, handle, keep, result
        self.fallbacks = None

    def end(self):
        if self._h is None:
            raise JstspError("this solve has already been completed")
        n = C.c_int(0)
        h, self._h = self._h, None
        check(self._ctx._lib.jstsp_proposed_algorithm_end(self._ctx.handle, h, C.byref(n)), "jstsp_proposed_algorithm_end")
        self.fallbacks = int(n.value)
        self._keep = None
        return self._result


def proposed_algorithm_begin(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type="approximate", *, indx_S=None, want_ce=True,
                             ctx=None):
    """The two-phase form of :func:`proposed_algorithm` for torch CUDA tensors (include/jstsp.h:
    jstsp_proposed_algorithm_begin_c32 / _end): enqueues the solve and returns a :class:`PendingSolve` without waiting for the
    GPU; ``handle.end()`` returns the outputs.  Between the two the host is free (e.g. to build the next batch)."""
    a_sub = _Arg(subY, np.complex64, "subY")
    a_om = _Arg(Omega, np.float32, "Omega")
    a_A = _Arg(A, np.complex64, "A")
    a_B = _Arg(B, np.complex64, "B")
    a_ix = _Arg(None, np.int32, "indx_S", allow_none=True)
    batch, N, M = a_sub.batch, a_sub.R, a_sub.C
    Gr, G2 = a_A.C, a_B.R
    if (a_om.batch, a_om.R, a_om.C) != (batch, N, M):
        raise ValueError("Omega must have the shape of subY")
    if a_A.R != N or a_B.C != M:
        raise ValueError("size(A,1) must equal size(subY,1) and size(B,2) must equal size(subY,2)")
    if indx_S is not None:
        import torch
        a_ix = _Arg(indx_S.reshape(batch, Gr * G2, 1).to(torch.int32).contiguous(), np.int32, "indx_S")
    c, mem, dev = _ctx_for([a_sub, a_om, a_A, a_B, a_ix], ctx)
    if mem != DEVICE:
        raise ValueError("proposed_algorithm_begin takes torch CUDA tensors (a host call has nothing to overlap: use proposed_algorithm)")
    sA = _shared_stride(a_A, N * Gr, batch, "A")
    sB = _shared_stride(a_B, G2 * M, batch, "B")
    tY, ptY = _scalars(tau_Y, batch, "tau_Y")
    tS, ptS = _scalars(tau_S, batch, "tau_S")
    rh, prh = _scalars(rho, batch, "rho")
    pS, fS = _out(True, batch, Gr, G2, np.complex64, dev)
    pY, fY = _out(True, batch, N, M, np.complex64, dev)
    pce, fce = _out(True, batch, int(Imax), 3, np.float64, dev) if want_ce else (None, None)
    tcode = _lib.TYPE_APPROXIMATE if type == "approximate" else _lib.TYPE_STD
    h = C.c_void_p()
    check(c._lib.jstsp_proposed_algorithm_begin_c32(c.handle, N, M, Gr, G2, batch, a_sub.ptr, a_om.ptr, a_A.ptr, sA, a_B.ptr, sB,
                                                    int(Imax), ptY, ptS, prh, tcode, a_ix.ptr, pS, pY, pce, C.byref(h)),
          "jstsp_proposed_algorithm_begin_c32")
    sq = not a_sub.batched
    return PendingSolve(c, h, (a_sub, a_om, a_A, a_B, a_ix, tY, tS, rh), (fS(sq), fY(sq), (fce(sq) if want_ce else None)))


def proposed_algorithm_angles(subY, Omega, indx_S, A, B, Imax, tau_Y, tau_S, rho, type="approximate",
                              greedy_nnz=None, *, want_ce=True, ctx=None):
    """basic_system_functions/proposed_algorithm_angles.m:1 (``greedy_nnz`` is unused there too).
    ``indx_S``: 1-based column-major linear indices, (Gr*G2,) or (batch, Gr*G2)."""
    return proposed_algorithm(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type, indx_S=indx_S,
                              want_ce=want_ce, ctx=ctx)


# ----------------------------------------------------------------------------- kernel level
def correlate(K, A, B, *, ctx=None):
    """``A' * K * B'`` (Gr x G2) — ``K2'*k`` of proposed_algorithm.m:47 in structured form."""
    a_K, a_A, a_B = _Arg(K, np.complex64, "K"), _Arg(A, np.complex64, "A"), _Arg(B, np.complex64, "B")
    batch, N, M, Gr, G2 = a_K.batch, a_K.R, a_K.C, a_A.C, a_B.R
    if a_A.R != N or a_B.C != M:
        raise ValueError("shape mismatch")
    c, mem, dev = _ctx_for([a_K, a_A, a_B], ctx)
    p, f = _out(mem == DEVICE, batch, Gr, G2, np.complex64, dev)
    check(c._lib.jstsp_correlate_c32(c.handle, N, M, Gr, G2, batch, a_K.ptr, a_A.ptr,
                                     _shared_stride(a_A, N * Gr, batch, "A"), a_B.ptr,
                                     _shared_stride(a_B, G2 * M, batch, "B"), p, mem), "jstsp_correlate_c32")
    return f(not a_K.batched)


def synthesize(S, A, B, *, ctx=None):
    """``A * S * B`` (N x M) — ``K2*s`` of proposed_algorithm.m:38,58."""
    a_S, a_A, a_B = _Arg(S, np.complex64, "S"), _Arg(A, np.complex64, "A"), _Arg(B, np.complex64, "B")
    batch, Gr, G2, N, M = a_S.batch, a_S.R, a_S.C, a_A.R, a_B.C
    if a_A.C != Gr or a_B.R != G2:
        raise ValueError("shape mismatch")
    c, mem, dev = _ctx_for([a_S, a_A, a_B], ctx)
    p, f = _out(mem == DEVICE, batch, N, M, np.complex64, dev)
    check(c._lib.jstsp_synthesize_c32(c.handle, N, M, Gr, G2, batch, a_S.ptr, a_A.ptr,
                                      _shared_stride(a_A, N * Gr, batch, "A"), a_B.ptr,
                                      _shared_stride(a_B, G2 * M, batch, "B"), p, mem), "jstsp_synthesize_c32")
    return f(not a_S.batched)


def gradient_head(Tc, A, GA, RV=None, *, ctx=None):
    """``Res = A'*Tc - RV`` and ``P1 = GA*Res`` (Gr x G2 each) — the 64-term products of the gradient step,
    proposed_algorithm.m:47-48, as the solver forms them (N = Gr = 64, G2 a multiple of 64)."""
    a_T, a_A, a_G = _Arg(Tc, np.complex64, "Tc"), _Arg(A, np.complex64, "A"), _Arg(GA, np.complex64, "GA")
    a_R = _Arg(RV, np.complex64, "RV", allow_none=True)
    batch, N, G2, Gr = a_T.batch, a_T.R, a_T.C, a_A.C
    if a_A.R != N or (a_G.R, a_G.C) != (Gr, Gr) or (RV is not None and (a_R.batch, a_R.R, a_R.C) != (batch, Gr, G2)):
        raise ValueError("shape mismatch")
    c, mem, dev = _ctx_for([a_T, a_A, a_G, a_R], ctx)
    pr, fr = _out(mem == DEVICE, batch, Gr, G2, np.complex64, dev)
    pp, fp = _out(mem == DEVICE, batch, Gr, G2, np.complex64, dev)
    check(c._lib.jstsp_gradient_head_c32(c.handle, N, Gr, G2, batch, a_T.ptr, a_A.ptr, _shared_stride(a_A, N * Gr, batch, "A"),
                                         a_G.ptr, _shared_stride(a_G, Gr * Gr, batch, "GA"), a_R.ptr, pr, pp, mem),
          "jstsp_gradient_head_c32")
    return fr(not a_T.batched), fp(not a_T.batched)


def ls_estimate(Y, A, B, *, ctx=None):
    """``pinv(A)*Y*pinv(B)`` — the LS baseline of plot_errorVSsnr.m:83 (SVD-based float64 pinv of every factor that
    fits the in-LDS kernel; the fp32 Gram inverse with a conditioning check for larger, full-rank factors)."""
    a_Y, a_A, a_B = _Arg(Y, np.complex64, "Y"), _Arg(A, np.complex64, "A"), _Arg(B, np.complex64, "B")
    batch, N, M, Gr, G2 = a_Y.batch, a_Y.R, a_Y.C, a_A.C, a_B.R
    if a_A.R != N or a_B.C != M:
        raise ValueError("shape mismatch")
    c, mem, dev = _ctx_for([a_Y, a_A, a_B], ctx)
    p, f = _out(mem == DEVICE, batch, Gr, G2, np.complex64, dev)
    check(c._lib.jstsp_ls_c32(c.handle, N, M, Gr, G2, batch, a_Y.ptr, a_A.ptr, _shared_stride(a_A, N * Gr, batch, "A"),
                              a_B.ptr, _shared_stride(a_B, G2 * M, batch, "B"), p, mem), "jstsp_ls_c32")
    return f(not a_Y.batched)


def pinv(A, *, ctx=None):
    """MATLAB's ``pinv(A)`` as the drivers call it (plot_errorVSsnr.m:83): SVD-based, float64 on the device,
    singular values below ``max(size(A))*eps(norm(A))`` dropped.  ``A``: (rows, cols) or (batch, rows, cols)."""
    a_A = _Arg(A, np.complex64, "A")
    c, mem, dev = _ctx_for([a_A], ctx)
    p, f = _out(mem == DEVICE, a_A.batch, a_A.C, a_A.R, np.complex64, dev)
    check(c._lib.jstsp_pinv_c32(c.handle, a_A.R, a_A.C, a_A.batch, a_A.ptr, p, mem), "jstsp_pinv_c32")
    return f(not a_A.batched)


def svt(Y, tau, *, ctx=None):
    """benchmark_algorithms/svt.m:1 — singular-value soft threshold."""
    a_Y = _Arg(Y, np.complex64, "Y")
    c, mem, dev = _ctx_for([a_Y], ctx)
    t, pt = _scalars(tau, a_Y.batch, "tau")
    p, f = _out(mem == DEVICE, a_Y.batch, a_Y.R, a_Y.C, np.complex64, dev)
    check(c._lib.jstsp_svt_c32(c.handle, a_Y.R, a_Y.C, a_Y.batch, a_Y.ptr, pt, p, mem), "jstsp_svt_c32")
    return f(not a_Y.batched)


def nmse_spectral(S, Zbar, *, ctx=None):
    """plot_errorVSsnr.m:138-141 — ``min(1, norm(S-Zbar)^2/norm(Zbar)^2)`` with spectral norms."""
    a_S, a_Z = _Arg(S, np.complex64, "S"), _Arg(Zbar, np.complex64, "Zbar")
    if (a_S.batch, a_S.R, a_S.C) != (a_Z.batch, a_Z.R, a_Z.C):
        raise ValueError("S and Zbar must have the same shape")
    c, mem, dev = _ctx_for([a_S, a_Z], ctx)
    if mem == DEVICE:
        import torch
        out = torch.empty(a_S.batch, dtype=torch.float64, device=dev)
        ptr = out.data_ptr()
    else:
        out = np.empty(a_S.batch, dtype=np.float64)
        ptr = out.ctypes.data
    check(c._lib.jstsp_nmse_spectral_c32(c.handle, a_S.R, a_S.C, a_S.batch, a_S.ptr, a_Z.ptr, ptr, mem),
          "jstsp_nmse_spectral_c32")
    return out if a_S.batched else out[0]


def lambda_max_sequence(G, *, ctx=None):
    """``lambda_max`` of a sequence of batches of Hermitian matrices, ``G``: (steps, batch, n, n) numpy complex64 (each matrix
    column-major = its transpose in C order; Hermitian, so the conjugate).  Matrix t of step s is warm-started from matrix t of
    step s - 1 — the kernel behind ``convergence_error(:,1:2)`` (proposed_algorithm.m:67,69) as the ADMM loops drive it.
    Returns (steps, batch) float32."""
    G = np.asarray(G)
    if G.ndim != 4 or G.shape[2] != G.shape[3]:
        raise ValueError("G must be (steps, batch, n, n)")
    steps, batch, n, _ = G.shape
    Gc = np.ascontiguousarray(np.swapaxes(G, 2, 3).astype(np.complex64))        # column-major matrices
    c = ctx if ctx is not None else _lib.default_context(0)
    out = np.empty((steps, batch), dtype=np.float32)
    check(c._lib.jstsp_lambda_max_sequence_c32(c.handle, n, batch, steps, Gc.ctypes.data, out.ctypes.data, HOST),
          "jstsp_lambda_max_sequence_c32")
    return out


def rate(S, Zbar, noise_var, *, ctx=None):
    """plot_rateVSframelength.m:81,113,130,135 — ``log2(real(det(eye(Nr) + 1/Nr*Zbar*Zbar'/(noise_var + nmse))))`` with
    the (uncapped) spectral-norm NMSE of ``S``."""
    a_S, a_Z = _Arg(S, np.complex64, "S"), _Arg(Zbar, np.complex64, "Zbar")
    if (a_S.batch, a_S.R, a_S.C) != (a_Z.batch, a_Z.R, a_Z.C):
        raise ValueError("S and Zbar must have the same shape")
    c, mem, dev = _ctx_for([a_S, a_Z], ctx)
    if mem == DEVICE:
        import torch
        out = torch.empty(a_S.batch, dtype=torch.float64, device=dev)
        ptr = out.data_ptr()
    else:
        out = np.empty(a_S.batch, dtype=np.float64)
        ptr = out.ctypes.data
    check(c._lib.jstsp_rate_c32(c.handle, a_S.R, a_S.C, a_S.batch, a_S.ptr, a_Z.ptr, float(noise_var), ptr, mem),
          "jstsp_rate_c32")
    return out if a_S.batched else out[0]


def mmv_omp(A, Y, K, *, norm="l2", ctx=None):
    """Joint (MMV) OMP — ``spx.pursuit.joint.OrthogonalMatchingPursuit(A, K).solve(Y).Z`` of the drivers
    (plot_errorVSsnr.m:116-117; sparse-plex is un-vendored and unpinned: the published simultaneous OMP, atom score
    ``||A(:,g)'*R||_2`` or ``_1``).  ``A``: (N, Gr) or (batch, N, Gr); ``Y``: (N, S) or (batch, N, S).
    Returns (Z (Gr, S), support (1-based atoms in selection order, 0 beyond the count), count)."""
    a_A, a_Y = _Arg(A, np.complex64, "A"), _Arg(Y, np.complex64, "Y")
    batch, N, S, Gr = a_Y.batch, a_Y.R, a_Y.C, a_A.C
    if a_A.R != N:
        raise ValueError("size(A,1) must equal size(Y,1)")
    c, mem, dev = _ctx_for([a_A, a_Y], ctx)
    pz, fz = _out(mem == DEVICE, batch, Gr, S, np.complex64, dev)
    pi, fi = _out(mem == DEVICE, batch, int(K), 1, np.int32, dev)
    pc, fc = _out(mem == DEVICE, batch, 1, 1, np.int32, dev)
    check(c._lib.jstsp_mmv_omp_c32(c.handle, N, Gr, S, batch, a_A.ptr, _shared_stride(a_A, N * Gr, batch, "A"), a_Y.ptr,
                                   int(K), 1 if norm == "l1" else 2, pz, pi, pc, mem), "jstsp_mmv_omp_c32")
    sq = not a_Y.batched
    return fz(sq), fi(sq)[..., 0], fc(sq)[..., 0, 0]


def tssr(Y_prop, Omega, A, B, Imax, tau, rho, K, *, norm="l2", ctx=None):
    """The drivers' two-stage TSSR baseline (plot_errorVSsnr.m:151,158-162, a commented recipe): matrix completion by
    ``mc_svt`` followed by joint OMP on ``Y_svt*pinv(B)`` with the dictionary ``A``.  Returns (S_tssr, Y_svt, S_svt) with
    ``S_svt = pinv(A)*Y_svt*pinv(B)`` the "SVT-based" estimate of :151-152."""
    Y_svt = mc_svt(Y_prop, Omega, Imax, tau, rho, ctx=ctx)
    PB = pinv(B, ctx=ctx)
    # Y_svt*pinv(B) and pinv(A)*(...): both on the library's synthesis kernel (A*S*B with one factor the identity)
    n = Y_svt.shape[-2]
    if _is_torch(Y_svt):
        import torch
        eye = colmajor(torch.eye(n, dtype=torch.complex64, device=Y_svt.device))
    else:
        eye = np.eye(n, dtype=np.complex64)
    T = synthesize(Y_svt, eye, PB, ctx=ctx)
    S_svt = ls_estimate(Y_svt, A, B, ctx=ctx)             # pinv(A)*Y_svt*pinv(B)  (:151)
    Z, _, _ = mmv_omp(A, T, K, norm=norm, ctx=ctx)
    return Z, Y_svt, S_svt


def mc_svt(OH, Omega, Imax, tau, rho, *, ctx=None):
    """benchmark_algorithms/mc_svt.m:1."""
    a_O, a_om = _Arg(OH, np.complex64, "OH"), _Arg(Omega, np.float32, "Omega")
    c, mem, dev = _ctx_for([a_O, a_om], ctx)
    t, pt = _scalars(tau, a_O.batch, "tau")
    r, pr = _scalars(rho, a_O.batch, "rho")
    p, f = _out(mem == DEVICE, a_O.batch, a_O.R, a_O.C, np.complex64, dev)
    check(c._lib.jstsp_mc_svt_c32(c.handle, a_O.R, a_O.C, a_O.batch, a_O.ptr, a_om.ptr, int(Imax), pt, pr, p,
                                  mem), "jstsp_mc_svt_c32")
    return f(not a_O.batched)


def mc_admm(Htrue, OH, Omega, Imax, tau, rho, *, want_ce=True, ctx=None):
    """benchmark_algorithms/mc_admm.m:1 — returns (X, convergence_error (batch, Imax))."""
    a_H = _Arg(Htrue, np.complex64, "Htrue", allow_none=not want_ce)
    a_O, a_om = _Arg(OH, np.complex64, "OH"), _Arg(Omega, np.float32, "Omega")
    c, mem, dev = _ctx_for([a_H, a_O, a_om], ctx)
    t, pt = _scalars(tau, a_O.batch, "tau")
    r, pr = _scalars(rho, a_O.batch, "rho")
    p, f = _out(mem == DEVICE, a_O.batch, a_O.R, a_O.C, np.complex64, dev)
    if want_ce:
        pce, fce = _out(mem == DEVICE, a_O.batch, int(Imax), 1, np.float64, dev)
    else:
        pce, fce = None, None
    check(c._lib.jstsp_mc_admm_c32(c.handle, a_O.R, a_O.C, a_O.batch, a_H.ptr, a_O.ptr, a_om.ptr, int(Imax), pt,
                                   pr, p, pce, mem), "jstsp_mc_admm_c32")
    sq = not a_O.batched
    ce = None
    if want_ce:
        ce = fce(sq)
        ce = ce[..., 0]
    return f(sq), ce


def sparse_admm(Htrue, OH, Dr, Dt, Imax, *, want_ce=True, ctx=None):
    """benchmark_algorithms/sparse_admm.m:1 — returns (S, convergence_error (batch, Imax))."""
    a_H = _Arg(Htrue, np.complex64, "Htrue", allow_none=not want_ce)
    a_O = _Arg(OH, np.complex64, "OH")
    a_Dr, a_Dt = _Arg(Dr, np.complex64, "Dr"), _Arg(Dt, np.complex64, "Dt")
    if a_Dr.batched or a_Dt.batched:
        raise ValueError("Dr and Dt are shared by the batch (2-D)")
    c, mem, dev = _ctx_for([a_H, a_O, a_Dr, a_Dt], ctx)
    p, f = _out(mem == DEVICE, a_O.batch, a_O.R, a_O.C, np.complex64, dev)
    if want_ce:
        pce, fce = _out(mem == DEVICE, a_O.batch, int(Imax), 1, np.float64, dev)
    else:
        pce, fce = None, None
    check(c._lib.jstsp_sparse_admm_c32(c.handle, a_O.R, a_O.C, a_Dr.C, a_Dt.C, a_O.batch, a_H.ptr, a_O.ptr,
                                       a_Dr.ptr, a_Dt.ptr, int(Imax), p, pce, mem), "jstsp_sparse_admm_c32")
    sq = not a_O.batched
    return f(sq), (fce(sq)[..., 0] if want_ce else None)


def OMP(A, v, m, snr=None, *, want_target=True, ctx=None):
    """benchmark_algorithms/OMP.m:1 — returns (x_hat, indexSet, v, targetMatrix).

    ``A``: (measures, size_d) or (batch, measures, size_d); ``v``: (measures,) or (batch, measures).
    ``indexSet`` is an int32 array of 1-based atom indices (the reference's 1 x m cell).
    ``snr`` is accepted and ignored, as in the reference."""
    a_A = _Arg(A, np.complex64, "A")
    tor = _is_torch(v)
    single = v.ndim == 1
    v3 = (v.reshape(1, -1, 1) if single else v.reshape(v.shape[0], -1, 1))
    if tor:
        v3 = colmajor(v3)
    a_v = _Arg(v3, np.complex64, "v")
    batch, meas, size_d = a_v.batch, a_A.R, a_A.C
    if a_v.R != meas:
        raise ValueError("length(v) must equal size(A,1)")
    c, mem, dev = _ctx_for([a_A, a_v], ctx)
    px, fx = _out(mem == DEVICE, batch, size_d, 1, np.complex64, dev)
    pi, fi = _out(mem == DEVICE, batch, int(m), 1, np.int32, dev)
    if want_target:
        pt, ft = _out(mem == DEVICE, batch, meas, int(m), np.complex64, dev)
    else:
        pt, ft = None, None
    check(c._lib.jstsp_omp_c32(c.handle, meas, size_d, batch, a_A.ptr, _shared_stride(a_A, meas * size_d, batch, "A"),
                               a_v.ptr, int(m), px, pi, pt, mem), "jstsp_omp_c32")
    x = fx(single)[..., 0]
    idx = fi(single)[..., 0]
    return x, idx, v, (ft(single) if want_target else None)


def omp_kron(Af, Bf, y, m, *, ctx=None):
    """OMP.m on the Kronecker dictionary ``kron(Bf.', Af)`` given by its factors (never formed;
    plot_errorVSdelays.m:77 builds the dictionary this way).  ``y``: (N*M,) or (batch, N*M) in
    column-major vec order.  Returns (x_hat (Gr*G2), indexSet (1-based))."""
    a_A, a_B = _Arg(Af, np.complex64, "Af"), _Arg(Bf, np.complex64, "Bf")
    tor = _is_torch(y)
    single = y.ndim == 1
    y3 = (y.reshape(1, -1, 1) if single else y.reshape(y.shape[0], -1, 1))
    if tor:
        y3 = colmajor(y3)
    a_y = _Arg(y3, np.complex64, "y")
    batch, N, Gr, G2, M = a_y.batch, a_A.R, a_A.C, a_B.R, a_B.C
    if a_y.R != N * M:
        raise ValueError("length(y) must be N*M")
    c, mem, dev = _ctx_for([a_A, a_B, a_y], ctx)
    px, fx = _out(mem == DEVICE, batch, Gr * G2, 1, np.complex64, dev)
    pi, fi = _out(mem == DEVICE, batch, int(m), 1, np.int32, dev)
    check(c._lib.jstsp_omp_kron_c32(c.handle, N, M, Gr, G2, batch, a_A.ptr, _shared_stride(a_A, N * Gr, batch, "Af"),
                                    a_B.ptr, _shared_stride(a_B, G2 * M, batch, "Bf"), a_y.ptr, int(m), px, pi, mem),
          "jstsp_omp_kron_c32")
    return fx(single)[..., 0], fi(single)[..., 0]


def _is_c128(x):
    return str(getattr(x, "dtype", "")) in ("complex128", "torch.complex128")


def vamp(y, A, sigma, L, *, nit=100, ctx=None):
    """benchmark_algorithms/vamp.m:1 — ``x = vamp(y, A, sigma, L)`` (dense dictionary, min(M, N) <= 2048:
    the drivers' 512 x 512 ``kron((B*B').', A)`` included).
    ``y``: (M,) or (batch, M).  ``nit`` = 100 is what the reference always runs.

    complex128 inputs (``y`` and ``A``) take ``jstsp_vamp_c64``: float64 storage and arithmetic on the device, which
    reproduces the reference's output at nit = 100 per trial (csrc/vamp64.hip); complex64 inputs the fp32-storage path."""
    f64 = _is_c128(A) and _is_c128(y)
    cdt = np.complex128 if f64 else np.complex64
    a_A = _Arg(A, cdt, "A")
    tor = _is_torch(y)
    single = y.ndim == 1
    y3 = (y.reshape(1, -1, 1) if single else y.reshape(y.shape[0], -1, 1))
    if tor:
        y3 = colmajor(y3)
    a_y = _Arg(y3, cdt, "y")
    batch, M, N = a_y.batch, a_A.R, a_A.C
    if a_y.R != M:
        raise ValueError("length(y) must equal size(A,1)")
    c, mem, dev = _ctx_for([a_A, a_y], ctx)
    px, fx = _out(mem == DEVICE, batch, N, 1, cdt, dev)
    fn = c._lib.jstsp_vamp_c64 if f64 else c._lib.jstsp_vamp_c32
    check(fn(c.handle, M, N, batch, a_y.ptr, a_A.ptr, _shared_stride(a_A, M * N, batch, "A"),
             float(sigma), float(L), int(nit), px, mem), "jstsp_vamp_c64" if f64 else "jstsp_vamp_c32")
    return fx(single)[..., 0]


def vamp_kron(Y, Af, Gb, sigma, L, *, nit=100, ctx=None):
    """``vamp(vec(Y), kron(Gb.', Af), sigma, L)`` without forming the dictionary — the call of
    plot_errorVSsnr.m:79-80,100 with ``Gb = B*B'``, ``Y = Y_hbf*B'``.  Returns X (Gr x G2), x = vec(X).
    complex128 inputs (all three) take the float64 path ``jstsp_vamp_kron_c64`` (see ``vamp``)."""
    f64 = _is_c128(Y) and _is_c128(Af) and _is_c128(Gb)
    cdt = np.complex128 if f64 else np.complex64
    a_Y, a_A, a_G = _Arg(Y, cdt, "Y"), _Arg(Af, cdt, "Af"), _Arg(Gb, cdt, "Gb")
    batch, Na, G2, Gr = a_Y.batch, a_Y.R, a_Y.C, a_A.C
    if a_A.R != Na or (a_G.R, a_G.C) != (G2, G2):
        raise ValueError("shape mismatch")
    c, mem, dev = _ctx_for([a_Y, a_A, a_G], ctx)
    px, fx = _out(mem == DEVICE, batch, Gr, G2, cdt, dev)
    fn = c._lib.jstsp_vamp_kron_c64 if f64 else c._lib.jstsp_vamp_kron_c32
    check(fn(c.handle, Na, Gr, G2, batch, a_Y.ptr, a_A.ptr, _shared_stride(a_A, Na * Gr, batch, "Af"), a_G.ptr,
             _shared_stride(a_G, G2 * G2, batch, "Gb"), float(sigma), float(L), int(nit), px, mem),
          "jstsp_vamp_kron_c64" if f64 else "jstsp_vamp_kron_c32")
    return fx(not a_Y.batched)


def sparse_sca_estim(rhat, rvar, var0, p1, *, ctx=None):
    """``[xhat, xvar] = SparseScaEstim(CAwgnEstimIn(0, var0), p1).estim(rhat, rvar)`` on real coordinates with the complex
    log-likelihood branch (MPbased_solvers/main/SparseScaEstim.m:76-166, CAwgnEstimIn.m:94-102,181-184) - the denoiser of every
    VAMP iteration (vamp.m:23-25), stand-alone.  ``rhat``: 1-D float64 numpy array (host path); returns two float64 arrays."""
    r = np.ascontiguousarray(np.asarray(rhat, dtype=np.float64).reshape(-1))
    c = ctx or _lib.default_context(0)
    xh, xv = np.empty_like(r), np.empty_like(r)
    check(c._lib.jstsp_sparse_sca_estim_f64(c.handle, r.size, r.ctypes.data, float(rvar), float(var0), float(p1), xh.ctypes.data, xv.ctypes.data,
                                            HOST), "jstsp_sparse_sca_estim_f64")
    return xh.reshape(np.shape(rhat)), xv.reshape(np.shape(rhat))


def cawgn_estim_out(y, phat, pvar, wvar, *, ctx=None):
    """``[zhat, zvar] = CAwgnEstimOut(y, wvar).estim(phat, pvar)`` with scale = 1 (MPbased_solvers/main/CAwgnEstimOut.m:97-108) on real
    coordinates - the likelihood of every VAMP iteration (vamp.m:30), stand-alone.  Returns (zhat array, zvar scalar)."""
    yy = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    pp = np.ascontiguousarray(np.asarray(phat, dtype=np.float64).reshape(-1))
    if yy.size != pp.size:
        raise ValueError("y and phat must have the same number of entries")
    c = ctx or _lib.default_context(0)
    zh = np.empty_like(yy)
    zv = C.c_double(0.0)
    check(c._lib.jstsp_cawgn_estim_out_f64(c.handle, yy.size, yy.ctypes.data, pp.ctypes.data, float(pvar), float(wvar), zh.ctypes.data,
                                           C.byref(zv), HOST), "jstsp_cawgn_estim_out_f64")
    return zh.reshape(np.shape(y)), float(zv.value)
