"""Float64 numpy restatement of the VAMP baseline (TEST INFRASTRUCTURE).

    x = vamp(y, A, sigma, L)                           benchmark_algorithms/vamp.m:1-55
      -> VampGlmEst(EstimIn, EstimOut, B, vampOpt)      MPbased_solvers/VAMP/VampGlmEst.m:324-549
         denoiser  SparseScaEstim(CAwgnEstimIn(0, nx/L), L/nx)   main/SparseScaEstim.m:76-166,
                                                                  main/CAwgnEstimIn.m:94-102,181-184
         likelihood CAwgnEstimOut(b, sigma, map=0)                main/CAwgnEstimOut.m:97-108
         options   VampGlmOpt                                     VAMP/VampGlmOpt.m:5-27

Reference quirks reproduced:
 * the complex system is real-stacked (vamp.m:3-4) but ``r1init = eps*1i`` (:45) is a complex
   SCALAR: it broadcasts at iteration 1 and makes every later ``rhat`` complex, so
   ``SparseScaEstim`` takes its *complex* log-likelihood branch (:100-103) on real-stacked data;
 * the tolerance stop is commented out (VampGlmEst.m:505-507): always ``nitMax = 100`` iterations;
 * ``r2`` uses the UNclipped ``gam2x`` (:369-379); ``alf`` has ``- eps`` (:401);
 * ``rvar`` is floored at eps after ``loglike1`` was computed (SparseScaEstim.m:96).

``vamp_literal`` follows the files line by line with a dense real-stacked matrix and its full
SVD.  ``vamp_kron`` runs the same recurrences for ``Phi = kron(Gb.', Af)`` (the dictionary the
drivers pass, plot_errorVSsnr.m:79) in complex arithmetic through the factors' decompositions,
never forming Phi.
"""
from __future__ import annotations

import numpy as np

EPS = np.finfo(np.float64).eps
GAM_MIN, GAM_MAX = 1e-8, 1e14            # VampGlmOpt.m:7-8


def _bg_denoise(rhat, rvar, var0, p1):
    """SparseScaEstim.estim (:76-166) around CAwgnEstimIn (mean0 = 0, var0), x0 = 0.

    ``rhat`` is complex (the eps*1i contamination), so the complex branch of loglike0 is taken.
    """
    rvar = np.asarray(rvar, dtype=np.float64)
    loglike1 = -(np.log(np.pi) + np.log(var0 + rvar) + np.abs(rhat) ** 2 / (var0 + rvar))   # CAwgnEstimIn.m:181-184
    rvar = np.maximum(rvar, EPS)                                                            # SparseScaEstim.m:96
    loglike0 = -(np.log(np.pi) + np.log(rvar) + np.abs(rhat) ** 2 / rvar)                   # :100-103 (complex branch)
    exparg = loglike0 - loglike1 + np.log(1 - p1) - np.log(p1)                              # :107
    exparg = np.maximum(np.minimum(exparg, 500), -500)                                      # :108-109
    py1 = 1.0 / (1.0 + np.exp(exparg))                                                      # :110
    py0 = 1 - py1
    gain = var0 / (var0 + rvar)                                                             # CAwgnEstimIn.m:100-102
    xhat1 = gain * rhat
    xvar1 = gain * rvar
    xhat = py1 * xhat1                                                                      # :160 (x0 = 0)
    xvar = py1 * (np.abs(xhat1) ** 2 - np.abs(xhat) ** 2) + py1 * xvar1 + py0 * (0 - np.abs(xhat) ** 2)   # :163-165
    return xhat, xvar


def _awgn_like(phat, pvar, y, wvar):
    """CAwgnEstimOut.estim (:97-108) with scale = 1."""
    gain = pvar / (pvar + wvar)
    zhat = gain * (y - phat) + phat
    zvar = wvar * gain
    return zhat, zvar


def _clip(g):
    return np.minimum(np.maximum(g, GAM_MIN), GAM_MAX)


def vamp_literal(y, A, sigma, L, nit=100, damp=0.85, trace=None):
    """benchmark_algorithms/vamp.m:1-55, line by line (dense real-stacked matrix + full SVD)."""
    A = np.asarray(A, dtype=np.complex128)
    y = np.asarray(y, dtype=np.complex128).reshape(-1)
    B = np.block([[A.real, -A.imag], [A.imag, A.real]])             # vamp.m:3
    b = np.concatenate([y.real, y.imag])                             # :4
    nx = B.shape[1]                                                  # :6
    MM = B.shape[0]                                                  # :7
    beta = L / nx                                                    # :23
    var0 = np.ones(nx) / beta                                        # :18,:24  (xvar0 = ones)
    wvar = sigma                                                     # :20
    U, s, _ = np.linalg.svd(B)                                       # :32
    d = np.concatenate([s ** 2, np.zeros(MM - s.size)])              # :34
    M, N = MM, nx
    if M > N:
        # vamp.m sets opt.U and opt.d but never opt.V, so for M > N VampGlmEst recomputes BOTH from the
        # eigen-decomposition of A'*A (VampGlmEst.m:196-218: `isempty(opt.V)` -> `[V,D] = eig(AhA); d = diag(D)`)
        d, V = np.linalg.eigh(B.T @ B)
    dl = M / N                                                       # VampGlmEst.m:257
    r1 = EPS * 1j                                                    # vamp.m:45 (complex scalar, broadcasts)
    p1 = np.zeros(M)                                                 # VampGlmEst.m:331
    gam1x = 1e-8                                                     # VampGlmOpt.m:25
    gam1z = 1e-8                                                     # :27
    x1 = z2 = gam2z = None
    for i in range(1, nit + 1):                                      # :347
        if i > 1:
            x1old, z2old, gam2zold, gam1xold = x1, z2, gam2z, gam1x  # :354-359
        x1, xvar1 = _bg_denoise(r1 * np.ones(N), np.ones(N) / gam1x, var0, beta)   # :361
        eta1x = 1.0 / np.mean(xvar1)                                 # :362
        if i > 1:
            x1 = damp * x1 + (1 - damp) * x1old                      # :363-365
        gam2x = eta1x - gam1x                                        # :366
        r2 = (x1 * eta1x - r1 * gam1x) / gam2x                       # :367 (unclipped gam2x)
        gam2x = _clip(gam2x)                                         # :376
        z1, zvar1 = _awgn_like(p1, np.ones(M) / gam1z, b, wvar)      # :378
        eta1z = 1.0 / np.mean(zvar1)                                 # :379
        gam2z = eta1z - gam1z                                        # :380
        p2 = (z1 * eta1z - p1 * gam1z) / gam2z                       # :381
        gam2z = _clip(gam2z)                                         # :390
        if i > 1:
            gam2z = damp * gam2z + (1 - damp) * gam2zold             # :391-393
        q = 1.0 / (d + gam2x / gam2z)                                # :397
        alf = (1 / N) * (d @ q) - EPS                                # :398
        if M <= N:                                                   # :399-403
            Ar2 = B @ r2
            t = (U.T @ (p2 - Ar2)) * q
            x2 = r2 + B.T @ (U @ t)
            z2 = Ar2 + U @ (d * t)
        else:                                                        # :407-411 (not reached by any driver of the reference)
            t = V.T @ (r2 * (gam2x / gam2z) + B.T @ p2)
            x2 = V @ (t * q)
            z2 = B @ x2
        if i > 1:
            z2 = damp * z2 + (1 - damp) * z2old                      # :411-413
        r1 = (x2 - r2 * (1 - alf)) / alf                             # :464
        p1 = (dl * z2 - p2 * alf) / (dl - alf)                       # :465
        gam1x = _clip(gam2x * alf / (1 - alf))                       # :469,:479
        gam1z = _clip(gam2z * (dl - alf) / alf)                      # :480,:489
        if i > 1:
            gam1x = damp * gam1x + (1 - damp) * gam1xold             # :490-492
        if trace is not None:
            trace.append(dict(alf=alf, gam1x=gam1x, gam1z=gam1z, gam2x=gam2x, gam2z=gam2z, eta1x=eta1x))
    return x1[: N // 2] + 1j * x1[N // 2:]                           # vamp.m:54


def vamp_dense(y, A, sigma, L, nit=100, damp=0.85):
    """Structured form for a dense dictionary: the same recurrences in complex arithmetic with the
    complex SVD A = U diag(s) V^H.  The real-stacked B has each singular value twice and the
    realification of U as left vectors, so U^T(.) .* q on 2M reals equals U^H(.) .* q on M complex
    numbers; only the denoiser / likelihood act per real coordinate.  The O(eps) imaginary parts the
    reference carries are dropped (real arithmetic + complex-branch formulas)."""
    A = np.asarray(A, dtype=np.complex128)
    y = np.asarray(y, dtype=np.complex128).reshape(-1)
    Mc, Nc = A.shape
    if Mc > Nc:                                                      # VampGlmEst.m:196-218,407-411: eig(A'A) = V diag(d) V'
        d, V = np.linalg.eigh(A.conj().T @ A)
        return _vamp_complex(y, lambda x: A @ x, lambda z: A.conj().T @ z, lambda t: V @ t, lambda x: V.conj().T @ x,
                             d, Mc, Nc, sigma, L, nit, damp)
    U, s, _ = np.linalg.svd(A, full_matrices=True)
    d = np.concatenate([s ** 2, np.zeros(Mc - s.size)])              # per complex row (each counted twice below)
    return _vamp_complex(y, lambda x: A @ x, lambda z: A.conj().T @ z, lambda z: U @ z, lambda z: U.conj().T @ z,
                         d, Mc, Nc, sigma, L, nit, damp)


def vamp_kron(Y, Af, Gb, sigma, L, nit=100, damp=0.85):
    """VAMP for ``Phi = kron(Gb.', Af)`` and ``y = vec(Y)`` (plot_errorVSsnr.m:79-80,100:
    ``Phi = kron((B*B').', A); y = vec(Y_hbf*B')``), never forming Phi.

    Phi vec(X) = vec(Af X Gb).  With Af = Ua Sa Va^H and the Hermitian Gb = Ub Lb Ub^H:
    Phi = (conj(Ub) (x) Ua) (Lb (x) Sa) (conj(Ub) (x) Va)^H, so the left singular vectors act as
    U^H vec(P) = vec(Ua^H P conj(Ub)... ) — written below with matrices.
    Returns the estimate as a Gr x G2 matrix.
    """
    Af = np.asarray(Af, dtype=np.complex128)
    Gb = np.asarray(Gb, dtype=np.complex128)
    Y = np.asarray(Y, dtype=np.complex128)
    Na, Gr = Af.shape
    G2 = Gb.shape[0]
    lb, Ub = np.linalg.eigh((Gb + Gb.conj().T) / 2)                 # Gb = Ub diag(lb) Ub^H
    if Na > Gr:
        # M > N (VampGlmEst.m:196-218,407-411): Phi'Phi = (conj(Ub) (x) Va) (lb^2 (x) la) (conj(Ub) (x) Va)^H with
        # Af'Af = Va diag(la) Va^H;  V vec(T) = vec(Va T Ub^H),  V^H vec(X) = vec(Va^H X Ub)
        la, Va = np.linalg.eigh(Af.conj().T @ Af)
        Dt = np.outer(la, lb ** 2)                                   # Gr x G2
        fAt = lambda Xv: (Af @ Xv.reshape(Gr, G2, order="F") @ Gb).reshape(-1, order="F")
        fAht = lambda Zv: (Af.conj().T @ Zv.reshape(Na, G2, order="F") @ Gb.conj().T).reshape(-1, order="F")
        fV = lambda Tv: (Va @ Tv.reshape(Gr, G2, order="F") @ Ub.conj().T).reshape(-1, order="F")
        fVh = lambda Xv: (Va.conj().T @ Xv.reshape(Gr, G2, order="F") @ Ub).reshape(-1, order="F")
        x = _vamp_complex(Y.reshape(-1, order="F"), fAt, fAht, fV, fVh, Dt.reshape(-1, order="F"), Na * G2, Gr * G2, sigma,
                          L, nit, damp)
        return x.reshape(Gr, G2, order="F")
    Ua, sa, _ = np.linalg.svd(Af, full_matrices=True)               # Na x Na
    sa_full = np.concatenate([sa, np.zeros(Na - sa.size)])
    # singular values of Phi: sa_i * |lb_j|; Phi's left vectors: Ua(:,i) (x) conj(Ub(:,j)) * sign(lb_j)
    sgn = np.where(lb < 0, -1.0, 1.0)
    D = np.outer(sa_full ** 2, lb ** 2)                             # d for the pair (i, j), Na x G2

    def fA(Xv):                                                      # Phi x
        return (Af @ Xv.reshape(Gr, G2, order="F") @ Gb).reshape(-1, order="F")

    def fAh(Zv):                                                     # Phi^H z
        return (Af.conj().T @ Zv.reshape(Na, G2, order="F") @ Gb.conj().T).reshape(-1, order="F")

    def fU(Tv):                                                      # U t,  U = (conj(Ub) sgn) (x) Ua
        T = Tv.reshape(Na, G2, order="F")
        return (Ua @ T @ (Ub.conj() * sgn).T).reshape(-1, order="F")

    def fUh(Zv):
        Z = Zv.reshape(Na, G2, order="F")
        return (Ua.conj().T @ Z @ (Ub * sgn)).reshape(-1, order="F")

    x = _vamp_complex(Y.reshape(-1, order="F"), fA, fAh, fU, fUh, D.reshape(-1, order="F"), Na * G2, Gr * G2, sigma,
                      L, nit, damp)
    return x.reshape(Gr, G2, order="F")


def _vamp_complex(y, fA, fAh, fU, fUh, d, Mc, Nc, sigma, L, nit, damp):
    """VampGlmEst.m:347-511 on the complexified system (2*Mc real rows, 2*Nc real unknowns)."""
    N, M = 2 * Nc, 2 * Mc
    beta = L / N                                                     # vamp.m:23 with nx = 2*Nc
    var0 = 1.0 / beta                                                # :24
    dl = M / N
    r1 = np.zeros(Nc, complex)                                       # |eps*1i|^2 ~ 5e-32: numerically zero
    p1 = np.zeros(Mc, complex)
    gam1x = gam1z = 1e-8
    dsum_w = 2.0                                                     # every complex singular value counts twice
    x1 = z2 = gam2z = None

    def den(rc, rvar):   # per REAL coordinate, complex-branch formulas
        xr, vr = _bg_denoise(rc.real + 0j, rvar * np.ones(Nc), var0, beta)
        xi, vi = _bg_denoise(rc.imag + 0j, rvar * np.ones(Nc), var0, beta)
        return xr.real + 1j * xi.real, np.concatenate([vr, vi])

    for i in range(1, nit + 1):
        if i > 1:
            x1old, z2old, gam2zold, gam1xold = x1, z2, gam2z, gam1x
        x1, xvar1 = den(r1, 1.0 / gam1x)
        eta1x = 1.0 / np.mean(xvar1)
        if i > 1:
            x1 = damp * x1 + (1 - damp) * x1old
        gam2x = eta1x - gam1x
        r2 = (x1 * eta1x - r1 * gam1x) / gam2x
        gam2x = _clip(gam2x)
        pvar = 1.0 / gam1z
        gain = pvar / (pvar + sigma)
        z1 = gain * (y - p1) + p1
        eta1z = 1.0 / (sigma * gain)
        gam2z = eta1z - gam1z
        p2 = (z1 * eta1z - p1 * gam1z) / gam2z
        gam2z = _clip(gam2z)
        if i > 1:
            gam2z = damp * gam2z + (1 - damp) * gam2zold
        q = 1.0 / (d + gam2x / gam2z)
        alf = (1 / N) * dsum_w * (d @ q) - EPS
        if Mc <= Nc:                                                 # VampGlmEst.m:402-406 (fU, fUh: U, U')
            Ar2 = fA(r2)
            t = fUh(p2 - Ar2) * q
            x2 = r2 + fAh(fU(t))
            z2 = Ar2 + fU(d * t)
        else:                                                        # :407-411 (fU, fUh: V, V'; d: eig(A'A))
            t = fUh(r2 * (gam2x / gam2z) + fAh(p2))
            x2 = fU(t * q)
            z2 = fA(x2)
        if i > 1:
            z2 = damp * z2 + (1 - damp) * z2old
        r1 = (x2 - r2 * (1 - alf)) / alf
        p1 = (dl * z2 - p2 * alf) / (dl - alf)
        gam1x = _clip(gam2x * alf / (1 - alf))
        gam1z = _clip(gam2z * (dl - alf) / alf)
        if i > 1:
            gam1x = damp * gam1x + (1 - damp) * gam1xold
    return x1
