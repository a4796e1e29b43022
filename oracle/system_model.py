"""Float64 numpy restatement of the reference's input builders (TEST INFRASTRUCTURE).

Every random draw is taken from an explicit ``draws`` object so that the product-side
generator (``jstsp19_amd.system_model``) can be fed *the same* numbers and compared
bit-for-bit in structure / to rounding in value.  MATLAB's own RNG streams cannot be
reproduced (the reference never seeds them).

Citations are ``path:line`` relative to /root/reference.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "matlab_round", "toeplitz_matlab", "qam4_alphabet", "create_beamformer",
    "wideband_mmwave_channel", "proposed_hbf", "hbf", "training_inputs_errorVSsnr",
    "rho_from_eigs", "wideband_hybBF_comm_system_training", "training_inputs_errorVSsnr_approx",
    "draw_trial_approx",
]


def matlab_round(x):
    """MATLAB ``round``: half away from zero (numpy rounds half to even)."""
    return float(np.sign(x) * np.floor(np.abs(x) + 0.5))


def toeplitz_matlab(c):
    """MATLAB ``toeplitz(r)`` of one (complex) vector: ``r`` is the first ROW, ``conj(r)`` the
    first column, the diagonal is ``r(1)`` => T(i,j) = r(j-i) for j>=i, conj(r(i-j)) for i>j
    (Hermitian off the diagonal) [MATLAB-sem]."""
    c = np.asarray(c).reshape(-1)
    n = c.size
    i = np.arange(n)[:, None]
    j = np.arange(n)[None, :]
    d = j - i
    return np.where(d >= 0, c[np.abs(d)], np.conj(c[np.abs(d)]))


def qam4_alphabet():
    """basic_system_functions/qam4mod.m:7 — the 4-QAM alphabet order used by randsrc."""
    s = 1 / np.sqrt(2)
    return np.array([(1 + 1j) * s, (-1 + 1j) * s, (1 - 1j) * s, (-1 - 1j) * s])


def create_beamformer(N, kind, rand_idx=None):
    """basic_system_functions/createBeamformer.m:1-35 (deterministic kinds + 'rand*' with
    explicit draws)."""
    n = np.arange(N)[:, None]
    if kind == "fft":                                   # :5  fft(eye(N)) = DFT matrix
        return np.exp(-2j * np.pi * n * np.arange(N)[None, :] / N) / np.sqrt(N)
    if kind == "ps":                                    # :12-13
        return np.exp(-1j * n * 2 * np.pi * np.arange(N)[None, :] / N) / np.sqrt(N)
    if kind == "ZC":                                    # :15-16
        return np.exp(-1j * 11 * n * np.pi * np.arange(1, N + 1)[None, :] / N) / np.sqrt(N)
    if kind in ("quantized_4", "quantized"):            # :17-32
        nq = 4 if kind == "quantized_4" else 6
        a = np.arange(2 ** nq)
        K = int(np.ceil(N / a.size))
        a = np.tile(a, K)[:N]
        omega = 2 * np.pi / 2 ** nq * a
        return np.exp(-1j * n * omega[None, :]) / np.sqrt(N)
    if kind == "rand":                                  # :6-7, rand_idx in 0..3, N x N
        return np.array([1, -1, 1j, -1j])[np.asarray(rand_idx)] / np.sqrt(N)
    if kind == "rand_ps":                               # :8-10, rand_idx in 1..32, length N
        return np.exp(-1j * n * 2 * np.pi * np.asarray(rand_idx)[None, :] / 32) / np.sqrt(N)
    raise ValueError(kind)


def _steer(phi, M):
    """wideband_mmwave_channel.m:42-52 — ULA response, half-wavelength spacing:
    wavenumber*spacing = pi, phase = pi*sin(0 - phi)*(0:M-1)', exp(-1j*phase), un-normalised."""
    ghz = 90
    wavelength = 30 / ghz
    spacing = 0.5 * wavelength
    wavenumber = 2 * np.pi / wavelength
    phase = wavenumber * spacing * np.sin(0 - phi) * np.arange(M)
    return np.exp(-1j * phase)


def _laplacian(u):
    """wideband_mmwave_channel.m:56-62 — beta*(exp(-sqrt(2)/50*pi) - cosh(u)), u ~ U(0,1)."""
    sigma_phi = 50
    beta = 1 / (1 - np.exp(-np.sqrt(2) * np.pi / sigma_phi))
    return beta * (np.exp(-np.sqrt(2) / sigma_phi * np.pi) - np.cosh(u))


def wideband_mmwave_channel(L, Mr, Mt, clusters, rays, Gr, Gt, gains, u_r, u_t):
    """basic_system_functions/wideband_mmwave_channel.m:1-39.

    ``gains`` complex (L, clusters*rays) = 1/sqrt(2)(randn + j randn) (:19), ``u_r``/``u_t``
    uniform draws (L, clusters*rays) for the AoA/AoD samplers (:20,:22), in the draw order
    (l, tap, ray).  Reproduces the reference's quirks: ``Ar(:,index)`` linear-indexes page 1
    (:24), so taps l>1 reuse tap 1's steering vectors, and ``H(:,:,l) += Hl`` sits inside the
    cluster loop (:29) so it adds the *cumulative* Hl once per cluster.
    """
    Np = clusters * rays
    H = np.zeros((Mr, Mt, L), complex)                  # :4
    Z = np.zeros((Gr, Gt, L), complex)                  # :6
    Ar = np.zeros((Mr, Np, L), complex)                 # :7
    At = np.zeros((Mt, Np, L), complex)                 # :8
    Dr = np.exp(-1j * np.arange(Mr)[:, None] * 2 * np.pi * np.arange(Gr)[None, :] / Gr) / np.sqrt(Mr)   # :9
    Dt = np.exp(-1j * np.arange(Mt)[:, None] * 2 * np.pi * np.arange(Gt)[None, :] / Gt) / np.sqrt(Mt)   # :10
    for l in range(L):                                  # :12
        Hl = np.zeros((Mr, Mt), complex)                # :14
        index = 0                                       # :16
        for _tap in range(clusters):                    # :17
            for _ray in range(rays):                    # :18
                coeff = gains[l, index]                 # :19
                Ar[:, index, l] = _steer(_laplacian(u_r[l, index]), Mr)   # :20-21
                At[:, index, l] = _steer(_laplacian(u_t[l, index]), Mt)   # :22
                # :24 — Ar(:,index) / At(:,index) address page 1 of the 3-D arrays
                Hl = Hl + coeff * np.outer(Ar[:, index, 0], At[:, index, 0].conj())
                index += 1                              # :27
            H[:, :, l] = H[:, :, l] + Hl                # :29 (cumulative Hl, once per cluster)
        H[:, :, l] = H[:, :, l] / np.sqrt(Np)           # :33
        Z[:, :, l] = Dr.conj().T @ H[:, :, l] @ Dt      # :35
    Zbar = Z.reshape(Gr, L * Gt, order="F")             # :38
    return H, Zbar, Ar, At, Dr, Dt


def _psi_bar(Psi_rows, Nt, T, L):
    """proposed_hbf.m:15-18 — ``Psi_bar(k,:,l) = Psi_i(l,:,k)``: row l of toeplitz(s_k).
    ``Psi_rows`` is (L, T, Nt): rows 1..L of each antenna's Toeplitz matrix."""
    Psi_bar = np.zeros((Nt, T, L), complex)
    for l in range(L):
        for k in range(Nt):
            Psi_bar[k, :, l] = Psi_rows[l, :, k]
    return Psi_bar


def toeplitz_rows(s, L):
    """Rows 1..L of ``toeplitz(s)`` without building the T x T matrix (only those rows are
    ever read, proposed_hbf.m:17)."""
    s = np.asarray(s).reshape(-1)
    T = s.size
    out = np.zeros((L, T), complex)
    for l in range(L):
        j = np.arange(T)
        d = j - l
        out[l, :] = np.where(d >= 0, s[np.abs(d)], np.conj(s[np.abs(d)]))
    return out


def proposed_hbf(H, Nn, Psi_rows, T, Lr_e, Lr, W, omega_rows):
    """basic_system_functions/proposed_hbf.m:1-44.

    ``omega_rows`` (T, Lr) int: for column t the 0-based rows ``randperm(Lr_e)(1:Lr)`` (:36-40).
    """
    _, Nt, L = H.shape                                  # :4
    Psi_bar = _psi_bar(Psi_rows, Nt, T, L)              # :8,:15-18
    W_e = W[:, :Lr_e]                                   # :11
    Y = np.zeros(Nn.shape, complex)                     # :14
    for l in range(L):
        Y = Y + H[:, :, l] @ Psi_bar[:, :, l]           # :19
    R = Y + Nn                                          # :22
    Omega = np.zeros((Lr_e, T))                         # :36
    for t in range(T):
        Omega[omega_rows[t], t] = 1                     # :37-41
    Y_prop = Omega * (W_e.conj().T @ R)                 # :42
    return Y_prop, W_e, Psi_bar, Omega, Y


def hbf(H, Nn, Psi_rows, T, Lr, W):
    """basic_system_functions/hbf.m:1-26."""
    _, Nt, L = H.shape
    Psi_bar = _psi_bar(Psi_rows, Nt, T, L)
    Y = np.zeros(Nn.shape, complex)
    for l in range(L):
        Y = Y + H[:, :, l] @ Psi_bar[:, :, l]           # :17
    R = Y + Nn                                          # :20
    W_c = W[:, :Lr]                                     # :23
    return W_c.conj().T @ R, W_c, Psi_bar, Y            # :24


def rho_from_eigs(Y, which="min6"):
    """plot_errorVSsnr.m:129-130 — ``eigs(Y'*Y)`` returns the 6 largest eigenvalues, so
    ``min(eigvalues)`` is sigma_6(Y)^2 (siblings use ``max`` = sigma_1^2,
    plot_errorVSdelays.m:128)."""
    s = np.linalg.svd(Y, compute_uv=False)
    fro2 = np.linalg.norm(Y, "fro") ** 2
    if which == "min6":
        # eigenvalues of the M x M matrix Y'*Y: the squared singular values padded with zeros
        M = Y.shape[1]
        ev = np.concatenate([s ** 2, np.zeros(max(0, M - s.size))])
        lam = ev[min(5, M - 1)]
    else:
        lam = s[0] ** 2
    return float(np.sqrt(lam * (1 / fro2)))


def training_inputs_errorVSsnr(params, draws):
    """plot_errorVSsnr.m:57-136 for one trial — the solver inputs of the proposed scheme.

    ``params``: dict(Nt, Nr, Mr_e, Gr, Gt, clusters, rays, L, Mr, T, noise_var) and, for the sibling drivers,
    optionally ``beamformer`` ('ZC' | 'fft' | 'ps'), ``rho_rule`` ('min' | 'max', plot_errorVSdelays.m:128),
    ``rho_scale`` (plot_errorVSzy.m:65), ``T_prop`` (plot_errorVSadmmiters.m:21: the frame itself).
    ``draws``: dict(gains, u_r, u_t, noise (Nr x T_prop complex standard normal * 1 — scaled
    here by sqrt(var/2) as :60 does with two real normals), qam_idx (Nt x T_prop ints 0..3),
    omega_rows (T_prop x Mr ints)).
    Returns dict(subY, Omega, A, B, tau_Y, tau_Z, rho, Zbar, H, indx_S).
    """
    p = params
    Nt, Nr, L, T = p["Nt"], p["Nr"], p["L"], p["T"]
    T_prop = p.get("T_prop") or T * Nt                  # :23
    H, Zbar, _, _, Dr, Dt = wideband_mmwave_channel(
        L, Nr, Nt, p["clusters"], p["rays"], p["Gr"], p["Gt"],
        draws["gains"], draws["u_r"], draws["u_t"])     # :57
    Nn = np.sqrt(p["noise_var"] / 2) * draws["noise"]   # :60 (noise = randn + 1j*randn)
    alphabet = qam4_alphabet()
    Psi_rows = np.zeros((L, T_prop, Nt), complex)
    for k in range(Nt):                                 # :63-67
        s = alphabet[draws["qam_idx"][k]]
        Psi_rows[:, :, k] = toeplitz_rows(s, L)
    W = create_beamformer(Nr, p.get("beamformer", "ZC"))   # :124
    Y_prop, W_tilde, Psi_bar, Omega, _ = proposed_hbf(
        H, Nn, Psi_rows, T_prop, p["Mr_e"], p["Mr"], W, draws["omega_rows"])   # :125
    tau_Y = 1 / np.linalg.norm(Y_prop, "fro") ** 2      # :127
    tau_Z = 1 / np.linalg.norm(Zbar, "fro") ** 2 / 2    # :128
    rho = p.get("rho_scale", 1.0) * rho_from_eigs(Y_prop, "max" if p.get("rho_rule") == "max" else "min6")   # :129-130
    A = W_tilde.conj().T @ Dr                           # :132
    Gt = p["Gt"]
    B = np.zeros((L * Gt, T_prop), complex)             # :133
    for l in range(L):
        B[l * Gt:(l + 1) * Gt, :] = Dt.conj().T @ Psi_bar[:, :, l]   # :135
    # :143 — sort(abs(vec(Zbar)),'descend'); MATLAB's sort is stable
    absz = np.abs(Zbar.reshape(-1, order="F"))
    indx_S = np.argsort(-absz, kind="stable") + 1
    return dict(subY=Y_prop, Omega=Omega, A=A, B=B, tau_Y=float(tau_Y), tau_Z=float(tau_Z),
                rho=rho, Zbar=Zbar, H=H, indx_S=indx_S)


def draw_trial(rng, params):
    """Draw one trial's random numbers in the reference's order (numpy Generator)."""
    p = params
    Np = p["clusters"] * p["rays"]
    L, Nr, Nt = p["L"], p["Nr"], p["Nt"]
    T_prop = p.get("T_prop") or p["T"] * Nt
    gains = np.zeros((L, Np), complex)
    u_r = np.zeros((L, Np))
    u_t = np.zeros((L, Np))
    for l in range(L):
        for i in range(Np):                             # wideband_mmwave_channel.m:19-22
            gains[l, i] = (rng.standard_normal() + 1j * rng.standard_normal()) / np.sqrt(2)
            u_r[l, i] = rng.random()
            u_t[l, i] = rng.random()
    noise = rng.standard_normal((Nr, T_prop)) + 1j * rng.standard_normal((Nr, T_prop))
    qam_idx = rng.integers(0, 4, size=(Nt, T_prop))
    omega_rows = np.stack([rng.permutation(p["Mr_e"])[:p["Mr"]] for _ in range(T_prop)])
    return dict(gains=gains, u_r=u_r, u_t=u_t, noise=noise, qam_idx=qam_idx,
                omega_rows=omega_rows)


def wideband_hybBF_comm_system_training(H, T, snr, subSamplingRatio, noise, pilots, omega_rows):
    """basic_system_functions/wideband_hybBF_comm_system_training.m:1-58 — the builder of the
    Alg.1-vs-Alg.2 driver (plot_errorVSsnr_approx.m:46): Gaussian Hermitian-Toeplitz pilots, the
    unitary DFT combiner over all Nr outputs, ``Lr = round(ratio*Nr)`` of them kept per column.

    Explicit draws: ``noise`` (Nr x T, randn + 1j*randn), ``pilots`` (Nt x T, randn + 1j*randn —
    scaled by 1/sqrt(2) here as :20 does), ``omega_rows`` (T x Lr ints, 0-based: randperm(Nr)(1:Lr)).
    Returns (Y_proposed_hbf, Y_conventional_hbf, W_tilde, Psi_bar, Omega, Lr).
    """
    Nr, Nt, L = H.shape                                 # :4
    Lr = int(matlab_round(subSamplingRatio * Nr))       # :5
    n = np.arange(Nr)
    W_tilde = np.exp(-2j * np.pi * np.outer(n, n) / Nr) / np.sqrt(Nr)   # :10  fft(eye(Nr))/sqrt(Nr)
    Nn = np.sqrt(snr / 2) * noise                       # :16
    Psi_bar = np.zeros((Nt, T, L), complex)             # :9
    for k in range(Nt):                                 # :19-22
        s = pilots[k] / np.sqrt(2)                      # :20
        Psi_bar[k] = toeplitz_rows(s, L).T              # :21,:28  Psi_bar(k,:,l) = Psi_i(l,:,k)
    R = np.zeros((Nr, T), complex)                      # :25
    for l in range(L):
        R = R + H[:, :, l] @ Psi_bar[:, :, l]           # :30
    R = R + Nn                                          # :33
    Omega = np.zeros((Nr, T))                           # :48
    for t in range(T):
        Omega[omega_rows[t, :Lr], t] = 1                # :49-53
    Y_conv = W_tilde.conj().T @ R                       # :57
    return Omega * Y_conv, Y_conv, W_tilde, Psi_bar, Omega, Lr   # :54


def training_inputs_errorVSsnr_approx(params, draws):
    """plot_errorVSsnr_approx.m:45-60 for one trial.

    ``params``: dict(Nt, Nr, L, T, clusters, rays, ratio, noise_var) (Gr = Nr, Gt = Nt, :9-10).
    ``draws``: dict(gains, u_r, u_t, noise (Nr x T), pilots (Nt x T), omega_rows (T x Lr)).
    Returns dict(subY, Omega, A, B, tau_X, tau_S, rho, Zbar, H, Lr).
    """
    p = params
    Nt, Nr, L, T = p["Nt"], p["Nr"], p["L"], p["T"]
    H, Zbar, _, _, Dr, Dt = wideband_mmwave_channel(
        L, Nr, Nt, p["clusters"], p["rays"], Nr, Nt,
        draws["gains"], draws["u_r"], draws["u_t"])     # :45
    Y_prop, _, W_tilde, Psi_bar, Omega, Lr = wideband_hybBF_comm_system_training(
        H, T, p["noise_var"], p["ratio"], draws["noise"], draws["pilots"], draws["omega_rows"])   # :46
    tau_X = 1 / np.linalg.norm(Y_prop, "fro") ** 2      # :50
    tau_S = tau_X / 2                                   # :51
    sv = np.linalg.svd(Y_prop, compute_uv=False)
    ev = np.concatenate([sv ** 2, np.zeros(max(0, T - sv.size))])
    rho = float(np.sqrt(ev[min(5, T - 1)] * (tau_X + tau_S) / 2))   # :52-53  eigs() -> 6 largest
    A = W_tilde.conj().T @ Dr                           # :54
    B = np.zeros((L * Nt, T), complex)                  # :55
    for l in range(L):
        B[l * Nt:(l + 1) * Nt, :] = Dt.conj().T @ Psi_bar[:, :, l]   # :57
    return dict(subY=Y_prop, Omega=Omega, A=A, B=B, tau_X=float(tau_X), tau_S=float(tau_S), rho=rho,
                Zbar=Zbar, H=H, Lr=Lr)


def draw_trial_approx(rng, params):
    """Draw one trial's random numbers of plot_errorVSsnr_approx.m:45-46 (numpy Generator)."""
    p = params
    Np = p["clusters"] * p["rays"]
    L, Nr, Nt, T = p["L"], p["Nr"], p["Nt"], p["T"]
    Lr = int(matlab_round(p["ratio"] * Nr))
    gains = (rng.standard_normal((L, Np)) + 1j * rng.standard_normal((L, Np))) / np.sqrt(2)
    u_r = rng.random((L, Np))
    u_t = rng.random((L, Np))
    noise = rng.standard_normal((Nr, T)) + 1j * rng.standard_normal((Nr, T))
    pilots = rng.standard_normal((Nt, T)) + 1j * rng.standard_normal((Nt, T))
    omega_rows = np.stack([rng.permutation(Nr)[:Lr] for _ in range(T)])
    return dict(gains=gains, u_r=u_r, u_t=u_t, noise=noise, pilots=pilots, omega_rows=omega_rows)
