"""Float64 numpy restatement of the reference's solver functions (TEST INFRASTRUCTURE).

Two forms of every solver:

* ``*_literal``  — follows the ``.m`` file line by line: dense ``kron`` matrices, ``lu``,
  full ``svd``, the reference's operation order.  Only feasible at small shapes.
* structured     — the same arithmetic through the exact identities of SURVEY.md §0.5
  (``K2*vec(S) = vec(A*S*B)``, ``K1 = diag(vec(Omega))`` ...).  Feasible at BASELINE sizes;
  this is the form the HIP path is compared against and the timed CPU baseline.

All citations are ``path:line`` relative to /root/reference.  MATLAB conventions that the
code relies on: column-major ``vec`` (basic_system_functions/vec.m:1-2), ``norm(X)`` of a
matrix = spectral norm, ``'`` = conjugate transpose, ``.'`` = transpose, ``sign(0) = 0``.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

__all__ = [
    "vec", "unvec", "svt", "soft_threshold_complex",
    "proposed_algorithm_literal", "proposed_algorithm",
    "proposed_algorithm_angles_literal", "proposed_algorithm_angles",
    "omp_literal", "omp", "sparse_admm_literal", "sparse_admm",
    "mc_svt", "mc_admm_literal", "mc_admm", "spectral_norm", "nmse_capped",
    "mmv_omp", "tssr", "rate",
]


# --------------------------------------------------------------------------- helpers
def vec(X):
    """basic_system_functions/vec.m:1-2 — ``X(:)`` column-major flatten."""
    return np.asarray(X).reshape(-1, order="F")


def unvec(x, rows, cols):
    """MATLAB ``reshape(x, rows, cols)`` (column-major)."""
    return np.asarray(x).reshape(rows, cols, order="F")


def spectral_norm(X):
    """MATLAB ``norm(X)``: 2-norm of a vector, largest singular value of a matrix."""
    X = np.asarray(X)
    if X.ndim == 1 or 1 in X.shape:
        return float(np.linalg.norm(X.ravel()))
    return float(np.linalg.norm(X, 2))


def _div(a, b):
    """IEEE division with MATLAB's x/0 -> Inf, 0/0 -> NaN (no Python exception)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.float64(a) / np.float64(b)


def nmse_capped(S, Zbar):
    """plot_errorVSsnr.m:138-141 — spectral-norm NMSE clipped to 1."""
    e = _div(spectral_norm(S - Zbar) ** 2, spectral_norm(Zbar) ** 2)
    return 1.0 if e > 1 else float(e)


def soft_threshold_complex(v, t):
    """proposed_algorithm.m:56 / sparse_admm.m:22 — separable real/imag soft threshold."""
    return (np.maximum(np.abs(v.real) - t, 0) * np.sign(v.real)
            + 1j * np.maximum(np.abs(v.imag) - t, 0) * np.sign(v.imag))


# --------------------------------------------------------------------------- svt
def svt(Y, tau):
    """benchmark_algorithms/svt.m:1-15.

    Full SVD, ``softThres = max(0, lambda - tau) .* lambda ./ abs(lambda)`` (:7), and the
    guard ``if(~isnan(softThres))`` (:8): any exactly-zero singular value makes 0/0 = NaN and
    the whole output becomes zeros (:12).
    """
    Y = np.asarray(Y, dtype=np.complex128)
    Mr, Mt = Y.shape
    U, lam, Vh = np.linalg.svd(Y, full_matrices=False)          # :5-6
    with np.errstate(divide="ignore", invalid="ignore"):
        soft = np.maximum(0.0, lam - tau) * lam / np.abs(lam)    # :7
    if not np.any(np.isnan(soft)):                               # :8
        return (U * soft) @ Vh                                   # :9-10
    return np.zeros((Mr, Mt), dtype=np.complex128)               # :12


# --------------------------------------------------------------------------- proposed_algorithm
def _dense_K1(Omega):
    """proposed_algorithm.m:14-19 — K1 = sum_i kron(diag(Omega(i,:))', Eii)."""
    N, M = Omega.shape
    K1 = np.zeros((N * M, N * M))
    for i in range(N):
        Eii = np.zeros((N, N))
        Eii[i, i] = 1
        K1 = K1 + np.kron(np.diag(Omega[i, :]).conj().T, Eii)
    return K1


def _lu_ls_setup(K2):
    """proposed_algorithm.m:29 — ``[L,U] = lu(K2)`` (L row-permuted lower trapezoidal)."""
    PL, U = sla.lu(K2, permute_l=True)
    return PL, U


def _lu_ls_solve(PL, U, k):
    """proposed_algorithm.m:53 — ``v = U\\(L\\k)``; rectangular ``L\\`` is a QR least squares."""
    y = np.linalg.lstsq(PL, k, rcond=None)[0]
    return sla.solve_triangular(U, y, lower=False)


def proposed_algorithm_literal(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type_,
                               indx_S=None, snapshots=None):
    """basic_system_functions/proposed_algorithm.m:1-73, line by line (dense Kronecker).

    With ``indx_S`` (1-based linear indices, a permutation of 1..Gr*Gt) it is
    proposed_algorithm_angles.m:1-85 (``greedy_nnz`` is unused there).
    """
    subY = np.asarray(subY, dtype=np.complex128)
    Omega = np.asarray(Omega, dtype=np.float64)
    A = np.asarray(A, dtype=np.complex128)
    B = np.asarray(B, dtype=np.complex128)
    N, M = subY.shape                                   # :3
    Gr = A.shape[1]                                     # :4
    Gt = B.shape[0]                                     # :5
    ce = np.zeros((Imax, 3))                            # :6
    X = np.zeros((N, M), complex)                       # :8
    V1 = np.zeros((N, M), complex)                      # :9
    V2 = np.zeros((N, M), complex)                      # :10
    C = np.zeros((N, M), complex)                       # :11
    s = np.zeros(Gr * Gt, complex)                      # :12
    K1 = _dense_K1(Omega)                               # :14-19
    iK1 = 1.0 / np.diag(K1 + 2 * rho * np.eye(N * M))   # :20 (diagonal of the sparse diag)
    K2 = np.kron(B.T, A)                                # :22
    if type_ == "approximate":                          # :24
        R = K2.conj().T @ K2                            # :25
        v = np.zeros(R.shape[1], complex)               # :26-27
    else:
        PL, U = _lu_ls_setup(K2)                        # :29
        v = None
    Omega_S = np.zeros((Gr, Gt))                        # angles :32
    S = np.zeros((Gr, Gt), complex)
    Y = np.zeros((N, M), complex)
    for i in range(1, Imax + 1):                        # :32
        if indx_S is not None:
            cnt = min(10 + 5 * i, Gt * Gr)              # angles :36
            lin = np.asarray(indx_S[:cnt], dtype=np.int64) - 1
            Omega_S[np.unravel_index(lin, (Gr, Gt), order="F")] = 1
            K3 = _dense_K1(Omega_S)                     # angles :37-43 (same construction)
        Y = svt(X - 1 / rho * V1, tau_Y / rho)          # :35
        b = vec(V1) + rho * vec(Y) + vec(subY) + vec(V2) + rho * vec(C) + rho * (K2 @ s)   # :38
        x = iK1 * b                                     # :39
        X = unvec(x, N, M)                              # :40
        k = vec(X) - 1 / rho * vec(V2) - vec(C)         # :43
        if type_ == "approximate":
            res = K2.conj().T @ k - R @ v               # :47
            alpha = _cdiv(np.vdot(res, res), np.vdot(res, R @ res))   # :48
            prev_v = v                                  # :49
            v = v + alpha * res                         # :50
            ce[i - 1, 2] = _div(np.linalg.norm(prev_v - v) ** 2, np.linalg.norm(prev_v) ** 2)  # :51
        else:
            v = _lu_ls_solve(PL, U, k)                  # :53
        s = soft_threshold_complex(v, tau_S / rho)      # :56
        if indx_S is not None:
            s = K3 @ s                                  # angles :68
        S = unvec(s, Gr, Gt)                            # :57
        Xs = A @ S @ B                                  # :58
        C = rho / (rho + 1) * (X - Xs - V2 / rho)       # :61
        V1 = V1 + rho * (Y - X)                         # :64
        V2 = V2 + rho * (C - X + Xs)                    # :65
        nX = spectral_norm(X) ** 2
        ce[i - 1, 0] = _div(spectral_norm(V1) ** 2, nX)   # :67
        ce[i - 1, 1] = _div(spectral_norm(V2) ** 2, nX)   # :69
        if snapshots is not None:
            snapshots.append(dict(X=X.copy(), V1=V1.copy(), V2=V2.copy(), C=C.copy(),
                                  S=S.copy(), Y=Y.copy(), v=np.array(v).copy()))
    return S, Y, ce


def _cdiv(a, b):
    """Complex IEEE division without Python exceptions (0/0 -> nan)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.complex128(a) / np.complex128(b)


def proposed_algorithm_angles_literal(subY, Omega, indx_S, A, B, Imax, tau_Y, tau_S, rho,
                                      type_, greedy_nnz=None, snapshots=None):
    """basic_system_functions/proposed_algorithm_angles.m:1-85 (``greedy_nnz`` unused)."""
    return proposed_algorithm_literal(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type_,
                                      indx_S=indx_S, snapshots=snapshots)


def proposed_algorithm(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type_, indx_S=None,
                       snapshots=None, want_ce=True):
    """Structured restatement of proposed_algorithm.m:1-73 / proposed_algorithm_angles.m:1-85.

    Identities: ``K2*s = vec(A*S*B)``; ``K2'*k = vec(A'*K*B')``;
    ``R*v = vec((A'A)*V*(BB'))``; ``iK1*b = b ./ (Omega + 2 rho)``;
    ``K3*s = Omega_S .* S``; 'std' branch ``U\\(L\\k)`` = LS solution
    ``vec(pinv(A)*K*pinv(B))`` when K2 has full column rank.
    """
    subY = np.asarray(subY, dtype=np.complex128)
    Omega = np.asarray(Omega, dtype=np.float64)
    A = np.asarray(A, dtype=np.complex128)
    B = np.asarray(B, dtype=np.complex128)
    N, M = subY.shape
    Gr = A.shape[1]
    Gt = B.shape[0]
    ce = np.zeros((Imax, 3))
    X = np.zeros((N, M), complex)
    V1 = np.zeros((N, M), complex)
    V2 = np.zeros((N, M), complex)
    C = np.zeros((N, M), complex)
    S = np.zeros((Gr, Gt), complex)
    Xs = np.zeros((N, M), complex)                      # A*S*B of the previous iteration (:38)
    inv_d = 1.0 / (Omega + 2 * rho)                     # :14-20
    Ah = A.conj().T
    Bh = B.conj().T
    if type_ == "approximate":
        GA = Ah @ A                                     # R = K2'*K2 = (B B')^T (x) (A'A)  (:25)
        GB = B @ Bh
        V = np.zeros((Gr, Gt), complex)
    else:
        pA = np.linalg.pinv(A)
        pB = np.linalg.pinv(B)
    Omega_S = np.zeros((Gr, Gt))
    Y = np.zeros((N, M), complex)
    for i in range(1, Imax + 1):
        if indx_S is not None:
            cnt = min(10 + 5 * i, Gt * Gr)
            lin = np.asarray(indx_S[:cnt], dtype=np.int64) - 1
            Omega_S[np.unravel_index(lin, (Gr, Gt), order="F")] = 1
        Y = svt(X - V1 / rho, tau_Y / rho)              # :35
        X = (V1 + rho * Y + subY + V2 + rho * C + rho * Xs) * inv_d   # :38-40
        K = X - V2 / rho - C                            # :43
        if type_ == "approximate":
            Res = Ah @ K @ Bh - GA @ V @ GB             # :47
            RRes = GA @ Res @ GB
            alpha = _cdiv(np.vdot(Res, Res), np.vdot(Res, RRes))     # :48
            Vp = V
            V = V + alpha * Res                         # :50
            ce[i - 1, 2] = _div(np.linalg.norm(Vp - V) ** 2, np.linalg.norm(Vp) ** 2)   # :51
        else:
            V = pA @ K @ pB                             # :53
        S = soft_threshold_complex(V, tau_S / rho)      # :56
        if indx_S is not None:
            S = Omega_S * S                             # angles :68
        Xs = A @ S @ B                                  # :58
        C = rho / (rho + 1) * (X - Xs - V2 / rho)       # :61
        V1 = V1 + rho * (Y - X)                         # :64
        V2 = V2 + rho * (C - X + Xs)                    # :65
        if want_ce:
            nX = spectral_norm(X) ** 2
            ce[i - 1, 0] = _div(spectral_norm(V1) ** 2, nX)          # :67
            ce[i - 1, 1] = _div(spectral_norm(V2) ** 2, nX)          # :69
        if snapshots is not None:
            snapshots.append(dict(X=X.copy(), V1=V1.copy(), V2=V2.copy(), C=C.copy(),
                                  S=S.copy(), Y=Y.copy(), v=vec(V).copy()))
    return S, Y, ce


def proposed_algorithm_angles(subY, Omega, indx_S, A, B, Imax, tau_Y, tau_S, rho, type_,
                              greedy_nnz=None, snapshots=None, want_ce=True):
    """Structured proposed_algorithm_angles.m:1-85."""
    return proposed_algorithm(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type_,
                              indx_S=indx_S, snapshots=snapshots, want_ce=want_ce)


# --------------------------------------------------------------------------- OMP
def omp_literal(A, v, m, snr=None):
    """benchmark_algorithms/OMP.m:1-32, line by line (``snr`` unused, exactly m iterations).

    Returns ``x_hat`` (size_d,), ``indexSet`` (1-based ints, length m), ``v``,
    ``targetMatrix`` (measures x m).
    """
    A = np.asarray(A, dtype=np.complex128)
    v = np.asarray(v, dtype=np.complex128).reshape(-1)
    measures, size_d = A.shape                           # :9
    r = v.copy()                                         # :10
    target = np.zeros((measures, 0), complex)            # :12
    index_set = []                                       # :13
    x = np.zeros(0, complex)
    t = 1
    while t <= m:                                        # :16
        corr = np.abs(A.conj().T @ r)                    # :17
        idx = int(np.argmax(corr))                       # first index on ties, like MATLAB max
        index_set.append(idx + 1)
        target = np.concatenate([target, A[:, idx:idx + 1]], axis=1)   # :18
        x = np.linalg.pinv(target) @ v                   # :19
        a = target @ x                                   # :20
        r = v - a                                        # :21
        t += 1
    x_hat = np.zeros(size_d, complex)                    # :27
    for i, idx in enumerate(index_set):                  # :29-32 (later duplicates overwrite)
        x_hat[idx - 1] = x[i]
    return x_hat, np.array(index_set, dtype=np.int64), v, target


def omp(A, v, m, snr=None):
    """Structured OMP.m: same selections, LS by ``lstsq`` on the selected columns.

    ``pinv(T)*v`` (OMP.m:19) is the minimum-norm least-squares solution; ``lstsq`` returns the
    same vector (also for a rank-deficient T, i.e. a re-selected atom).
    """
    A = np.asarray(A, dtype=np.complex128)
    v = np.asarray(v, dtype=np.complex128).reshape(-1)
    measures, size_d = A.shape
    r = v.copy()
    idxs = []
    x = np.zeros(0, complex)
    Ah = A.conj().T
    for _ in range(m):
        idx = int(np.argmax(np.abs(Ah @ r)))
        idxs.append(idx)
        T = A[:, idxs]
        x = np.linalg.lstsq(T, v, rcond=None)[0]
        r = v - T @ x
    x_hat = np.zeros(size_d, complex)
    for i, idx in enumerate(idxs):
        x_hat[idx] = x[i]
    return x_hat, np.array(idxs, dtype=np.int64) + 1, v, A[:, idxs]


def omp_kron(Af, Bf, y, m):
    """OMP.m on the Kronecker dictionary ``Phi = kron(Bf.', Af)`` without forming Phi.

    ``Phi'*r = vec(Af' * R * conj(Bf))`` with ``R = unvec(r)``; column ``idx`` of Phi is
    ``vec(Af(:,g) * Bf(h,:))`` with ``idx = g + Gr*h`` (0-based).  Used at BASELINE config 1
    (plot_errorVSdelays.m:77 builds the dictionary this way).
    """
    Af = np.asarray(Af, dtype=np.complex128)
    Bf = np.asarray(Bf, dtype=np.complex128)
    N, Gr = Af.shape
    G2, M = Bf.shape
    y = np.asarray(y, dtype=np.complex128).reshape(-1)
    r = y.copy()
    idxs, cols = [], []
    x = np.zeros(0, complex)
    for _ in range(m):
        Rm = unvec(r, N, M)
        corr = np.abs(vec(Af.conj().T @ Rm @ Bf.conj().T))
        idx = int(np.argmax(corr))
        g, h = idx % Gr, idx // Gr
        idxs.append(idx)
        cols.append(vec(np.outer(Af[:, g], Bf[h, :])))
        T = np.stack(cols, axis=1)
        x = np.linalg.lstsq(T, y, rcond=None)[0]
        r = y - T @ x
    x_hat = np.zeros(Gr * G2, complex)
    for i, idx in enumerate(idxs):
        x_hat[idx] = x[i]
    return x_hat, np.array(idxs, dtype=np.int64) + 1, y, np.stack(cols, axis=1)


# --------------------------------------------------------------------------- sparse_admm
def sparse_admm_literal(Htrue, OH, Dr, Dt, Imax):
    """benchmark_algorithms/sparse_admm.m:1-36, line by line (dense kron + dense solve)."""
    Htrue = np.asarray(Htrue, dtype=np.complex128)
    OH = np.asarray(OH, dtype=np.complex128)
    Dr = np.asarray(Dr, dtype=np.complex128)
    Dt = np.asarray(Dt, dtype=np.complex128)
    Mr, Mt = OH.shape                                   # :3
    Gr = Dr.shape[1]
    Gt = Dt.shape[1]
    ce = np.zeros(Imax)                                 # :6
    Z = np.zeros((Mr, Mt), complex)                     # :8
    R = np.zeros((Gr, Gt), complex)                     # :9
    rho = 0.01                                          # :12
    tau_s = 0.0001                                      # :13
    A = np.kron(Dt.conj(), Dr)                          # :15
    Bm = A.conj().T @ A - rho * np.eye(Mr * Mt)         # :16
    S = np.zeros((Mr, Mt), complex)
    for i in range(Imax):                               # :18
        v = vec(R + Z / rho)                            # :21
        s = soft_threshold_complex(v, tau_s / rho)      # :22
        S = unvec(s, Mr, Mt)                            # :23
        r = np.linalg.solve(Bm, vec(Z) - rho * s + A.conj().T @ vec(OH))   # :26
        R = unvec(r, Mr, Mt)                            # :27
        Z = Z + rho * (R - S)                           # :30
        ce[i] = _div(spectral_norm(Dr @ S @ Dt.conj().T - Htrue) ** 2,
                     spectral_norm(Htrue) ** 2)         # :32
    return S, ce


def sparse_admm(Htrue, OH, Dr, Dt, Imax, want_ce=True):
    """Structured sparse_admm.m: ``A'A = (Dt^T conj(Dt)) (x) (Dr' Dr)``.

    ``A'*vec(OH) = vec(Dr' * OH * Dt)``; the dense solve ``B \\ rhs`` (:26) with
    ``B = Gt_ (x) Gr_ - rho I`` is done in the eigenbases of the two Hermitian factor Grams:
    ``Gr_ = Ur diag(lr) Ur'``, ``Gt_ = Ut diag(lt) Ut'`` =>
    ``R = Ur * ((Ur' * RHS * conj(Ut)) ./ (lr lt^T - rho)) * Ut^T``.
    """
    Htrue = np.asarray(Htrue, dtype=np.complex128)
    OH = np.asarray(OH, dtype=np.complex128)
    Dr = np.asarray(Dr, dtype=np.complex128)
    Dt = np.asarray(Dt, dtype=np.complex128)
    Mr, Mt = OH.shape
    Gr = Dr.shape[1]
    Gt = Dt.shape[1]
    ce = np.zeros(Imax)
    Z = np.zeros((Mr, Mt), complex)
    R = np.zeros((Gr, Gt), complex)
    rho = 0.01
    tau_s = 0.0001
    Gr_ = Dr.conj().T @ Dr                               # Gr x Gr
    Gt_ = Dt.T @ Dt.conj()                               # (conj(Dt))' * conj(Dt), Gt x Gt
    lr, Ur = np.linalg.eigh(Gr_)
    lt, Ut = np.linalg.eigh(Gt_)
    den = np.outer(lr, lt) - rho
    AhOH = Dr.conj().T @ OH @ Dt
    S = np.zeros((Mr, Mt), complex)
    for i in range(Imax):
        V = R + Z / rho
        S = soft_threshold_complex(V, tau_s / rho)
        RHS = Z - rho * S + AhOH
        # (Gt_ (x) Gr_ - rho I) vec(R) = vec(RHS)  <=>  Gr_ R Gt_^T - rho R = RHS
        T = Ur.conj().T @ RHS @ Ut.conj()
        R = Ur @ (T / den) @ Ut.T
        Z = Z + rho * (R - S)
        if want_ce:
            ce[i] = _div(spectral_norm(Dr @ S @ Dt.conj().T - Htrue) ** 2,
                         spectral_norm(Htrue) ** 2)
    return S, ce


# --------------------------------------------------------------------------- matrix completion
def mc_svt(OH, Omega, Imax, tau, rho):
    """benchmark_algorithms/mc_svt.m:1-12."""
    OH = np.asarray(OH, dtype=np.complex128)
    Omega = np.asarray(Omega, dtype=np.float64)
    Y = np.zeros_like(OH)                               # :5
    X = np.zeros_like(OH)
    for _ in range(Imax):                               # :7
        X = svt(Y, tau / rho)                           # :8
        Y = Y + rho * (OH - Omega * X)                  # :9
    return X


def mc_admm_literal(Htrue, OH, Omega, Imax, tau, rho):
    """benchmark_algorithms/mc_admm.m:1-34, line by line (dense diag + dense solve)."""
    Htrue = np.asarray(Htrue, dtype=np.complex128)
    OH = np.asarray(OH, dtype=np.complex128)
    Omega = np.asarray(Omega, dtype=np.float64)
    Mr, Mt = OH.shape                                   # :3
    ce = np.zeros(Imax)                                 # :4
    X = np.zeros((Mr, Mt), complex)                     # :6
    Y = np.zeros((Mr, Mt), complex)                     # :7
    Z = np.zeros((Mr, Mt), complex)                     # :8
    Am = _dense_K1(Omega) + rho * np.eye(Mr * Mt)       # :11-17
    for i in range(Imax):                               # :20
        X = svt(Y - 1 / rho * Z, tau / rho)             # :22
        y = np.linalg.solve(Am, vec(OH) + vec(Z) + rho * vec(X))   # :24
        Y = unvec(y, Mr, Mt)                            # :25
        Z = Z + rho * (X - Y)                           # :26
        ce[i] = _div(spectral_norm(X - Htrue) ** 2, spectral_norm(Htrue) ** 2)   # :28
    return X, ce


def mc_admm(Htrue, OH, Omega, Imax, tau, rho, want_ce=True):
    """Structured mc_admm.m: ``A\\b`` with ``A = diag(vec(Omega)) + rho I`` is ``b ./ (Omega+rho)``."""
    Htrue = np.asarray(Htrue, dtype=np.complex128)
    OH = np.asarray(OH, dtype=np.complex128)
    Omega = np.asarray(Omega, dtype=np.float64)
    ce = np.zeros(Imax)
    X = np.zeros_like(OH)
    Y = np.zeros_like(OH)
    Z = np.zeros_like(OH)
    inv_d = 1.0 / (Omega + rho)
    for i in range(Imax):
        X = svt(Y - Z / rho, tau / rho)
        Y = (OH + Z + rho * X) * inv_d
        Z = Z + rho * (X - Y)
        if want_ce:
            ce[i] = _div(spectral_norm(X - Htrue) ** 2, spectral_norm(Htrue) ** 2)
    return X, ce


# --------------------------------------------------------------------------- joint (MMV) OMP, TSSR, rate
def mmv_omp(A, Y, K, norm="l2"):
    """Joint (simultaneous) OMP — what the drivers call through the UN-VENDORED, UNPINNED sparse-plex:
    ``spx.pursuit.joint.OrthogonalMatchingPursuit(A, K).solve(Y).Z`` (plot_errorVSsnr.m:116-117, :158-162).
    PARITY UNPINNED: sparse-plex is not in /root/reference and no version is recorded (README.md:9); this restates
    the published algorithm (Tropp-Gilbert-Strauss 2006 with the l1 row score, Chen-Huo 2006 with l2): one support
    for all columns, atom = argmax_g ||A(:,g)'*R||_p (first index on ties), least squares on the support; stops after
    K atoms, when min(N, Gr) atoms are in, when the new atom depends on the support, or when
    ||R||_F <= 1e-6 ||Y||_F.  Returns (Z (Gr x S), support (1-based, selection order))."""
    A = np.asarray(A, dtype=np.complex128)
    Y = np.asarray(Y, dtype=np.complex128)
    N, Gr = A.shape
    S = Y.shape[1]
    R = Y.copy()
    Z = np.zeros((Gr, S), complex)
    support = []
    y2 = np.linalg.norm(Y) ** 2
    for _ in range(min(K, N, Gr)):
        C = A.conj().T @ R
        score = np.sum(np.abs(C), axis=1) if norm == "l1" else np.sum(np.abs(C) ** 2, axis=1)
        score[support] = -1.0
        g = int(np.argmax(score))                         # first index on ties
        sub = A[:, support + [g]]
        if np.linalg.matrix_rank(sub, tol=1e-5 * np.linalg.norm(A[:, g])) < sub.shape[1]:
            break
        support.append(g)
        coef = np.linalg.lstsq(sub, Y, rcond=None)[0]
        R = Y - sub @ coef
        Z[:] = 0
        Z[support, :] = coef
        if np.linalg.norm(R) ** 2 <= 1e-12 * y2:
            break
    return Z, np.array(support, dtype=np.int64) + 1


def tssr(Y_prop, Omega, A, B, Imax, tau, rho, K, norm="l2"):
    """plot_errorVSsnr.m:151,158-162 (commented recipe): ``Y_svt = mc_svt(...)``; joint OMP of ``Y_svt*pinv(B)`` on ``A``."""
    Y_svt = mc_svt(Y_prop, Omega, Imax, tau, rho)
    T = Y_svt @ np.linalg.pinv(np.asarray(B, dtype=np.complex128))
    Z, _ = mmv_omp(A, T, K, norm)
    return Z, Y_svt, np.linalg.pinv(np.asarray(A, dtype=np.complex128)) @ T       # S_tssr, Y_svt, S_svt (:152)


def rate(S, Zbar, noise_var):
    """plot_rateVSframelength.m:81,113,130,135 —
    ``log2(real(det(eye(Nr) + 1/(Nr)*Zbar*Zbar'*1/(square_noise_variance+norm(Zbar-S)^2/norm(Zbar)^2))))``."""
    S = np.asarray(S, dtype=np.complex128)
    Zbar = np.asarray(Zbar, dtype=np.complex128)
    Nr = Zbar.shape[0]
    e = _div(spectral_norm(Zbar - S) ** 2, spectral_norm(Zbar) ** 2)
    M = np.eye(Nr) + (1.0 / Nr) * (Zbar @ Zbar.conj().T) / (noise_var + e)
    sign, logdet = np.linalg.slogdet(M)
    return float(logdet / np.log(2.0))
