// Batched Hermitian eigen-decomposition of small Gram matrices (n <= 128) by parallel-order
// two-sided Jacobi, one workgroup per matrix, matrix (and eigenvectors, n <= 64) in LDS.
//
// This is the `svd` of benchmark_algorithms/svt.m:5 in Gram form: for Z (N x M, N <= M)
// G = Z Z^H = U diag(sigma^2) U^H, and
//      svt(Z, tau) = U max(0, Sigma - tau) V^H = Z - Q Z,   Q = U diag(min(1, tau/sigma)) U^H
// (svt.m:7-10), so only the N x N factor is ever decomposed (the reference's full `svd`
// builds an M x M Vy).  The same kernel returns lambda_max for the spectral norms of
// proposed_algorithm.m:67,69 (`norm(V1)^2/norm(X)^2` = lambda_max ratio).
//
// Guard of svt.m:8-12: the reference zeroes the output when any singular value is exactly 0
// (0/0 = NaN in :7).  Reproduced for the one input whose zero is exact in every LAPACK, the
// all-zero matrix (the svt argument of iteration 1): all-zero Gram => Q = I, i.e. Z - Q Z = 0.
// For rank-deficient non-zero inputs LAPACK's zeros are only sometimes exact (a zero LAST row
// gives 0.0, a zero first or middle row 4e-16 with numpy's gesdd — DESIGN.md §5); such singular
// values are treated as <= tau (component removed), which is what the oracle computes then.
#include "common.h"

namespace jstsp {

// Round-robin (circle method) pairing: in round s (0 <= s < n-1) pair 0 is (n-1, s) and
// pair k >= 1 is ((s+k) mod (n-1), (s-k) mod (n-1)); every unordered pair meets once per sweep.
__device__ __forceinline__ void rr_pair(int n, int s, int k, int &p, int &q)
{
    int a, b;
    if (k == 0) { a = n - 1; b = s; }
    else {
        a = (s + k) % (n - 1);
        b = (s - k + (n - 1)) % (n - 1);
    }
    p = min(a, b);
    q = max(a, b);
}

template <bool WITH_V, bool V_IN_LDS>
__global__ __launch_bounds__(256) void jacobi_kernel(int mode, int n, int batch,
                                                      const float2 *Gpart, long long sGt, int nsplit,
                                                      long long sGs, const TrialParams *prm,
                                                      const float *tau, float2 *Q, float *lam_out,
                                                      float2 *Vglob)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const int ne = (n + 1) & ~1;             // even working size (odd n padded with a zero row/col)
    const int ld = ne + 1;                   // padded pitch: row access is bank-conflict free
    const int h = ne / 2;
    float2 *G = reinterpret_cast<float2 *>(smem_raw);            // [ne][ld] column-major
    float2 *V = G + (size_t)ne * ld;                             // [ne][ld] when V_IN_LDS
    float *rot = reinterpret_cast<float *>((WITH_V && V_IN_LDS) ? (V + (size_t)ne * ld) : V);  // [h][4]
    float *red = rot + 4 * h;                                    // [8]
    float2 *Vg = (WITH_V && !V_IN_LDS) ? Vglob + (size_t)t * ne * ne : nullptr;

    // ---- load G = sum of split-K partials, init V = I ------------------------------------
    for (int e = tid; e < ne * ne; e += 256) {
        const int i = e % ne, j = e / ne;
        float2 g = make_float2(0.f, 0.f);
        if (i < n && j < n) {
            const float2 *src = Gpart + (long long)t * sGt + i + (long long)n * j;
            for (int s = 0; s < nsplit; ++s) {
                const float2 v = src[(long long)s * sGs];
                g.x += v.x; g.y += v.y;
            }
        }
        G[i + ld * j] = g;
        if (WITH_V) {
            const float2 id = make_float2(i == j ? 1.f : 0.f, 0.f);
            if (V_IN_LDS) V[i + ld * j] = id; else Vg[i + (size_t)ne * j] = id;
        }
    }
    __syncthreads();
    // Hermitian-symmetrise (the MFMA Gram is Hermitian only up to rounding).
    for (int e = tid; e < ne * ne; e += 256) {
        const int i = e % ne, j = e / ne;
        if (i < j) {
            const float2 u = G[i + ld * j], l = G[j + ld * i];
            const float2 a = make_float2(0.5f * (u.x + l.x), 0.5f * (u.y - l.y));
            G[i + ld * j] = a;
            G[j + ld * i] = make_float2(a.x, -a.y);
        } else if (i == j) {
            G[i + ld * i].y = 0.f;
        }
    }
    __syncthreads();

    // Guard of svt.m:7-12 (see jacobi2_kernel in eig2.hip): all-zero Gram = all-zero svt argument => output 0, i.e. Q = I.
    if (WITH_V && mode == EIG_SVT_Q) {
        if (tid == 0) red[2] = 0.f;
        __syncthreads();
        for (int i = tid; i < n; i += 256)
            if (G[i + ld * i].x != 0.f) red[2] = 1.f;
        __syncthreads();
        if (red[2] == 0.f) {
            for (int e = tid; e < n * n; e += 256)
                Q[(size_t)t * n * n + e] = make_float2((e % n == e / n) ? 1.f : 0.f, 0.f);
            return;
        }
    }

    // dmax = largest diagonal entry: the absolute scale of the convergence test
    {
        float m = 0.f;
        for (int i = tid; i < ne; i += 256) m = fmaxf(m, fabsf(G[i + ld * i].x));
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = m;
        __syncthreads();
        if (tid == 0) red[1] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        __syncthreads();
    }
    const float dmax = red[1];

    const int MAX_SWEEPS = 14;
    for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
        if (tid == 0) red[0] = 0.f;
        __syncthreads();
        float worst = 0.f;
        for (int s = 0; s < ne - 1; ++s) {
            // -- phase 1: rotation angles of the h disjoint pairs of this round
            if (tid < h) {
                int p, q;
                rr_pair(ne, s, tid, p, q);
                const float a = G[p + ld * p].x, dd = G[q + ld * q].x;
                const float2 bq = G[p + ld * q];
                const float ab = sqrtf(bq.x * bq.x + bq.y * bq.y);
                float c = 1.f, wx = 0.f, wy = 0.f;
                const float scale = sqrtf(fabsf(a) * fabsf(dd));
                if (ab > 0.f && ab > 1e-8f * scale) {
                    // relative to sqrt(a d), floored at 1e-3 dmax so that noise-level
                    // off-diagonals between negligible eigenvalues do not block convergence
                    worst = fmaxf(worst, ab / fmaxf(scale, 1e-3f * dmax));
                    const float zeta = (dd - a) / (2.f * ab);
                    const float tt = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                    c = 1.f / sqrtf(1.f + tt * tt);
                    const float sn = tt * c;
                    wx = sn * bq.x / ab;      // w = s * e^{i phi}
                    wy = sn * bq.y / ab;
                }
                rot[4 * tid + 0] = c; rot[4 * tid + 1] = wx; rot[4 * tid + 2] = wy;
                rot[4 * tid + 3] = __int_as_float(p | (q << 16));
            }
            __syncthreads();
            // -- phase 2: column update  G <- G J,  V <- V J
            for (int e = tid; e < h * ne; e += 256) {
                const int k = e / ne, r = e % ne;
                const float c = rot[4 * k], wx = rot[4 * k + 1], wy = rot[4 * k + 2];
                if (wx == 0.f && wy == 0.f) continue;
                const int pq = __float_as_int(rot[4 * k + 3]);
                const int p = pq & 0xffff, q = pq >> 16;
                {
                    const float2 xp = G[r + ld * p], xq = G[r + ld * q];
                    // new_p = c xp - conj(w) xq ; new_q = w xp + c xq
                    G[r + ld * p] = make_float2(c * xp.x - (wx * xq.x + wy * xq.y),
                                                c * xp.y - (wx * xq.y - wy * xq.x));
                    G[r + ld * q] = make_float2((wx * xp.x - wy * xp.y) + c * xq.x,
                                                (wx * xp.y + wy * xp.x) + c * xq.y);
                }
                if (WITH_V) {
                    float2 *vp = V_IN_LDS ? &V[r + ld * p] : &Vg[r + (size_t)ne * p];
                    float2 *vq = V_IN_LDS ? &V[r + ld * q] : &Vg[r + (size_t)ne * q];
                    const float2 xp = *vp, xq = *vq;
                    *vp = make_float2(c * xp.x - (wx * xq.x + wy * xq.y),
                                      c * xp.y - (wx * xq.y - wy * xq.x));
                    *vq = make_float2((wx * xp.x - wy * xp.y) + c * xq.x,
                                      (wx * xp.y + wy * xp.x) + c * xq.y);
                }
            }
            __syncthreads();
            // -- phase 3: row update  G <- J^H G
            for (int e = tid; e < h * ne; e += 256) {
                const int k = e / ne, cc = e % ne;
                const float c = rot[4 * k], wx = rot[4 * k + 1], wy = rot[4 * k + 2];
                if (wx == 0.f && wy == 0.f) continue;
                const int pq = __float_as_int(rot[4 * k + 3]);
                const int p = pq & 0xffff, q = pq >> 16;
                const float2 yp = G[p + ld * cc], yq = G[q + ld * cc];
                // new_p = c yp - w yq ; new_q = conj(w) yp + c yq
                float2 np_ = make_float2(c * yp.x - (wx * yq.x - wy * yq.y),
                                         c * yp.y - (wx * yq.y + wy * yq.x));
                float2 nq_ = make_float2((wx * yp.x + wy * yp.y) + c * yq.x,
                                         (wx * yp.y - wy * yp.x) + c * yq.y);
                if (cc == p) { np_.y = 0.f; nq_ = make_float2(0.f, 0.f); }   // exact zero at (q,p)
                if (cc == q) { nq_.y = 0.f; np_ = make_float2(0.f, 0.f); }   // and at (p,q)
                G[p + ld * cc] = np_;
                G[q + ld * cc] = nq_;
            }
            __syncthreads();
        }
        // convergence: largest relative off-diagonal seen in this sweep
        if (tid < h) atomicMax(reinterpret_cast<int *>(&red[0]), __float_as_int(worst));
        __syncthreads();
        const float w = red[0];
        __syncthreads();
        if (w < 1e-4f) break;      // quadratic convergence: the sweep that started below 1e-4 ends near 1e-8 (see eig2.hip)
    }

    if (mode == EIG_LMAX) {
        float m = -1e30f;
        for (int i = tid; i < n; i += 256) m = fmaxf(m, G[i + ld * i].x);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = m;
        __syncthreads();
        if (tid == 0) lam_out[t] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        return;
    }
    if (WITH_V && mode == EIG_VECS) {
        // eigenvectors (columns) -> Q, eigenvalues -> lam_out[t*n + i]
        float2 *Qt = Q + (size_t)t * n * n;
        for (int e = tid; e < n * n; e += 256) {
            const int r = e % n, c = e / n;
            Qt[e] = V_IN_LDS ? V[r + ld * c] : Vg[r + (size_t)ne * c];
        }
        for (int i = tid; i < n; i += 256) lam_out[(size_t)t * n + i] = G[i + ld * i].x;
        return;
    }
    if (WITH_V) {
        // q_i = min(1, tau / sigma_i), sigma_i = sqrt(max(lambda_i, 0)); store in rot[]
        const float tv = tau ? tau[t] : prm[t].tauY_rho;
        __syncthreads();
        for (int i = tid; i < ne; i += 256) {
            const float lam = fmaxf(G[i + ld * i].x, 0.f);
            const float sig = sqrtf(lam);
            float qv = (sig > 0.f) ? fminf(1.f, tv / sig) : 1.f;
            rot[i] = qv;
        }
        __syncthreads();
        // Q[r][c] = sum_i q_i V[r][i] conj(V[c][i])
        float2 *Qt = Q + (size_t)t * n * n;
        for (int e = tid; e < n * n; e += 256) {
            const int r = e % n, c = e / n;
            float sx = 0.f, sy = 0.f;
            for (int i = 0; i < ne; ++i) {
                const float2 vr = V_IN_LDS ? V[r + ld * i] : Vg[r + (size_t)ne * i];
                const float2 vc = V_IN_LDS ? V[c + ld * i] : Vg[c + (size_t)ne * i];
                const float qi = rot[i];
                sx += qi * (vr.x * vc.x + vr.y * vc.y);
                sy += qi * (vr.y * vc.x - vr.x * vc.y);
            }
            Qt[r + (size_t)n * c] = make_float2(sx, sy);
        }
    }
}

// Eigenvectors fit in LDS beside G up to ne*(ne+1)*16 + small <= 160 KiB (n <= 98).
bool eig_needs_global_v(int n)
{
    const int ne = (n + 1) & ~1;
    const size_t mat = (size_t)ne * (ne + 1) * sizeof(float2);
    const size_t extra = (size_t)(4 * (ne / 2) + 8 + ne) * sizeof(float);
    return 2 * mat + extra > 160 * 1024;
}

int launch_eig(jstsp_ctx *ctx, int mode, int n, int batch, const float2 *Gpart, long long sGt,
               int nsplit, long long sGs, const TrialParams *prm, const float *tau, float2 *Q,
               float *lam_out, float2 *Vg)
{
    if (n > 128) return launch_eig_large(ctx, mode, n, batch, Gpart, sGt, nsplit, sGs, prm, tau, Q, lam_out);
    JSTSP_REQUIRE(n >= 1, JSTSP_E_UNSUPPORTED, "eig: matrix order %d", n);
    const int ne = (n + 1) & ~1;
    const int ld = ne + 1;
    const size_t mat = (size_t)ne * ld * sizeof(float2);
    const size_t extra = (size_t)(4 * (ne / 2) + 8 + ne) * sizeof(float);
    if (mode == EIG_LMAX) {
        const size_t sh = mat + extra;
        JSTSP_HIP(hipFuncSetAttribute((const void *)jacobi_kernel<false, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL((jacobi_kernel<false, false>), dim3(batch), dim3(256), sh, ctx->stream,
                           mode, n, batch, Gpart, sGt, nsplit, sGs, prm, tau, Q, lam_out,
                           (float2 *)nullptr);
    } else if (2 * mat + extra <= 160 * 1024) {
        const size_t sh = 2 * mat + extra;
        JSTSP_HIP(hipFuncSetAttribute((const void *)jacobi_kernel<true, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL((jacobi_kernel<true, true>), dim3(batch), dim3(256), sh, ctx->stream, mode,
                           n, batch, Gpart, sGt, nsplit, sGs, prm, tau, Q, lam_out, (float2 *)nullptr);
    } else {
        // n > ~96: eigenvectors do not fit beside G in 160 KiB of LDS; keep them in HBM/L2.
        const size_t sh = mat + extra;
        JSTSP_REQUIRE(Vg, JSTSP_E_NOMEM, "eig: n = %d needs a global eigenvector workspace", n);
        JSTSP_HIP(hipFuncSetAttribute((const void *)jacobi_kernel<true, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL((jacobi_kernel<true, false>), dim3(batch), dim3(256), sh, ctx->stream, mode,
                           n, batch, Gpart, sGt, nsplit, sGs, prm, tau, Q, lam_out, Vg);
    }
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
