// jstsp_omp_c32 / jstsp_omp_kron_c32 — benchmark_algorithms/OMP.m:1-32, batched over problems.
//
// Per iteration (OMP.m:16-24):
//   [~, idx] = max(abs(A'*r))          -> correlation on the MFMA GEMM (dense dictionary) or as
//                                         A^H R B^H (Kronecker dictionary kron(Bf.', Af), never formed),
//                                         then a wave64-shuffle argmax with first-index tie-break
//   targetMatrix = [targetMatrix, A(:,idx)];  x = pinv(targetMatrix)*v;  r = v - targetMatrix*x
//                                      -> the least squares is carried incrementally: the selected
//                                         atoms are orthonormalised (Gram-Schmidt with re-orthogonalisation,
//                                         fp64 dot products) so that r = v - Q Q^H v costs one pass.
// The reference never excludes chosen atoms (:18).  A re-selected atom makes targetMatrix rank
// deficient; pinv then splits the coefficient equally among the copies and x_hat keeps the
// last copy (:29-32) — reproduced through per-atom multiplicities.
#include "solver_common.h"
#include <algorithm>

namespace jstsp {

struct OmpState {
    float2 *Qb;        // [batch][meas][m]   orthonormal basis of the selected atoms
    float2 *Rm;        // [batch][m][m]      upper-triangular R (column-major), T_unique = Q R
    float2 *z;         // [batch][m]         Q^H v
    float2 *r;         // [batch][meas]      residual
    float2 *w;         // [batch][meas]      scratch column
    int *uniq;         // [batch][m]         0-based atom index of basis vector j
    int *mult;         // [batch][m]         multiplicity of basis vector j
    int *nu;           // [batch]            number of basis vectors
    int *sel;          // [batch][m]         0-based selected atom per iteration
};

__device__ __forceinline__ double2 block_sum2(double2 v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) { v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = v.x; sh[2 * (threadIdx.x >> 6) + 1] = v.y; }
    __syncthreads();
    return make_double2(sh[0] + sh[2] + sh[4] + sh[6], sh[1] + sh[3] + sh[5] + sh[7]);
}

// One workgroup per problem: argmax |corr|, then append the atom.
// Dense dictionary: atom = A[:, idx] (A + t*strideA, meas x size_d).
// Kronecker (Bf != nullptr): atom[i + N*j] = Af[i, g] * Bf[h, j], idx = g + Gr*h.
__global__ __launch_bounds__(256) void omp_step_kernel(int meas, int size_d, int m, int it, const float2 *corr,
                                                       const float2 *A, long long strideA, const float2 *Bf,
                                                       long long strideB, int N, int Gr, int G2, OmpState s)
{
    __shared__ double sh[8];
    __shared__ float shv[4];
    __shared__ int shi[4];
    __shared__ int s_idx, s_dup;
    const int t = blockIdx.x, tid = threadIdx.x;
    // ---- argmax of |corr| with first-index tie-break (MATLAB max) -------------------------------
    const float2 *c = corr + (long long)t * size_d;
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int i = tid; i < size_d; i += 256) {
        const float2 v = c[i];
        float a = sqrtf(v.x * v.x + v.y * v.y);
        if (a != a) a = -1.f;                       // NaN never wins unless everything is NaN
        if (a > best) { best = a; bi = i; }         // strided scan keeps the smallest index per thread
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((tid & 63) == 0) { shv[tid >> 6] = best; shi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k)
            if (shv[k] > best || (shv[k] == best && shi[k] < bi)) { best = shv[k]; bi = shi[k]; }
        if (bi == 0x7fffffff) bi = 0;
        s_idx = bi;
        s.sel[(long long)t * m + it] = bi;
        int dup = -1;
        const int nu = s.nu[t];
        for (int j = 0; j < nu; ++j)
            if (s.uniq[(long long)t * m + j] == bi) { dup = j; break; }
        s_dup = dup;
        if (dup >= 0) s.mult[(long long)t * m + dup] += 1;
    }
    __syncthreads();
    if (s_dup >= 0) return;                         // re-selected atom: span (and residual) unchanged
    const int idx = s_idx;
    const int u = s.nu[t];
    float2 *w = s.w + (long long)t * meas;
    float2 *Q = s.Qb + (long long)t * meas * m;
    float2 *Rc = s.Rm + (long long)t * m * m + (long long)u * m;       // column u of R
    float2 *r = s.r + (long long)t * meas;
    // ---- load the atom ----------------------------------------------------------------------------
    double nrm0 = 0;
    if (Bf) {
        const int g = idx % Gr, h = idx / Gr;
        const float2 *a = A + (long long)t * strideA + (long long)N * g;            // Af(:, g)
        const float2 *b = Bf + (long long)t * strideB + h;                          // Bf(h, :) stride G2
        for (int e = tid; e < meas; e += 256) {
            const float2 x = a[e % N], y = b[(long long)G2 * (e / N)];
            const float2 v = make_float2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
            w[e] = v;
            nrm0 += (double)v.x * v.x + (double)v.y * v.y;
        }
    } else {
        const float2 *a = A + (long long)t * strideA + (long long)meas * idx;
        for (int e = tid; e < meas; e += 256) {
            const float2 v = a[e];
            w[e] = v;
            nrm0 += (double)v.x * v.x + (double)v.y * v.y;
        }
    }
    nrm0 = block_sum2(make_double2(nrm0, 0), sh).x;
    for (int j = tid; j < m; j += 256) Rc[j] = make_float2(0.f, 0.f);
    __syncthreads();
    // ---- Gram-Schmidt against the basis, twice (re-orthogonalisation) ------------------------------
    for (int pass = 0; pass < 2; ++pass) {
        for (int j = 0; j < u; ++j) {
            const float2 *q = Q + (long long)meas * j;
            double2 d = make_double2(0, 0);
            for (int e = tid; e < meas; e += 256) {
                const float2 qq = q[e], ww = w[e];
                d.x += (double)qq.x * ww.x + (double)qq.y * ww.y;       // conj(q) * w
                d.y += (double)qq.x * ww.y - (double)qq.y * ww.x;
            }
            d = block_sum2(d, sh);
            const float hx = (float)d.x, hy = (float)d.y;
            for (int e = tid; e < meas; e += 256) {
                const float2 qq = q[e];
                float2 ww = w[e];
                ww.x -= hx * qq.x - hy * qq.y;
                ww.y -= hx * qq.y + hy * qq.x;
                w[e] = ww;
            }
            if (tid == 0) { Rc[j].x += hx; Rc[j].y += hy; }
            __syncthreads();
        }
    }
    double n2 = 0;
    for (int e = tid; e < meas; e += 256) { const float2 ww = w[e]; n2 += (double)ww.x * ww.x + (double)ww.y * ww.y; }
    n2 = block_sum2(make_double2(n2, 0), sh).x;
    if (!(n2 > 1e-12 * nrm0)) return;              // numerically dependent on the chosen atoms: adds nothing
    const float inv = (float)(1.0 / sqrt(n2));
    // ---- q_u = w/|w|, z_u = q_u^H v = q_u^H r (r is orthogonal to the old basis), r -= z_u q_u --------
    float2 *qu = Q + (long long)meas * u;
    double2 d = make_double2(0, 0);
    for (int e = tid; e < meas; e += 256) {
        const float2 ww = w[e], rr = r[e];
        const float2 qv = make_float2(ww.x * inv, ww.y * inv);
        qu[e] = qv;
        d.x += (double)qv.x * rr.x + (double)qv.y * rr.y;
        d.y += (double)qv.x * rr.y - (double)qv.y * rr.x;
    }
    d = block_sum2(d, sh);
    const float zx = (float)d.x, zy = (float)d.y;
    for (int e = tid; e < meas; e += 256) {
        const float2 qv = qu[e];
        float2 rr = r[e];
        rr.x -= zx * qv.x - zy * qv.y;
        rr.y -= zx * qv.y + zy * qv.x;
        r[e] = rr;
    }
    if (tid == 0) {
        Rc[u] = make_float2((float)sqrt(n2), 0.f);
        s.z[(long long)t * m + u] = make_float2(zx, zy);
        s.uniq[(long long)t * m + u] = idx;
        s.mult[(long long)t * m + u] = 1;
        s.nu[t] = u + 1;
    }
}

// x_unique = R^{-1} z (back substitution), x_hat(idx) = x_unique / multiplicity, indexSet (1-based),
// targetMatrix = the selected columns in selection order (OMP.m:18, 27-32).
__global__ __launch_bounds__(256) void omp_finish_kernel(int meas, int size_d, int m, const float2 *A,
                                                         long long strideA, const float2 *Bf, long long strideB,
                                                         int N, int Gr, int G2, OmpState s, float2 *x_hat,
                                                         int32_t *index_out, float2 *target_out)
{
    extern __shared__ float2 xs[];                 // [m]
    const int t = blockIdx.x, tid = threadIdx.x;
    const int u = s.nu[t];
    const float2 *R = s.Rm + (long long)t * m * m;
    if (tid == 0) {
        for (int i = u - 1; i >= 0; --i) {
            double ax = s.z[(long long)t * m + i].x, ay = s.z[(long long)t * m + i].y;
            for (int j = i + 1; j < u; ++j) {
                const float2 rij = R[i + (long long)m * j];
                ax -= (double)rij.x * xs[j].x - (double)rij.y * xs[j].y;
                ay -= (double)rij.x * xs[j].y + (double)rij.y * xs[j].x;
            }
            const double rii = R[i + (long long)m * i].x;
            xs[i] = make_float2((float)(ax / rii), (float)(ay / rii));
        }
    }
    for (int i = tid; i < size_d; i += 256) x_hat[(long long)t * size_d + i] = make_float2(0.f, 0.f);
    __syncthreads();
    if (tid == 0)
        for (int j = 0; j < u; ++j) {
            const float mu = (float)s.mult[(long long)t * m + j];
            x_hat[(long long)t * size_d + s.uniq[(long long)t * m + j]] = make_float2(xs[j].x / mu, xs[j].y / mu);
        }
    for (int it = tid; it < m; it += 256) index_out[(long long)t * m + it] = s.sel[(long long)t * m + it] + 1;
    if (target_out) {
        for (int it = 0; it < m; ++it) {
            const int idx = s.sel[(long long)t * m + it];
            float2 *o = target_out + ((long long)t * m + it) * meas;
            if (Bf) {
                const int g = idx % Gr, h = idx / Gr;
                const float2 *a = A + (long long)t * strideA + (long long)N * g;
                const float2 *b = Bf + (long long)t * strideB + h;
                for (int e = tid; e < meas; e += 256) {
                    const float2 x = a[e % N], y = b[(long long)G2 * (e / N)];
                    o[e] = make_float2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
                }
            } else {
                const float2 *a = A + (long long)t * strideA + (long long)meas * idx;
                for (int e = tid; e < meas; e += 256) o[e] = a[e];
            }
        }
    }
}

static int omp_alloc(Arena &a, OmpState &s, int meas, int m, int batch)
{
    s.Qb = a.get<float2>((size_t)batch * meas * m);
    s.Rm = a.get<float2>((size_t)batch * m * m);
    s.z = a.get<float2>((size_t)batch * m);
    s.r = a.get<float2>((size_t)batch * meas);
    s.w = a.get<float2>((size_t)batch * meas);
    s.uniq = a.get<int>((size_t)batch * m);
    s.mult = a.get<int>((size_t)batch * m);
    s.nu = a.get<int>(batch);
    s.sel = a.get<int>((size_t)batch * m);
    JSTSP_REQUIRE(s.Qb && s.Rm && s.z && s.r && s.w && s.uniq && s.mult && s.nu && s.sel, JSTSP_E_NOMEM,
                  "OMP: workspace exhausted");
    return 0;
}
static size_t omp_bytes(int meas, int m, int batch)
{
    return rnd256((size_t)batch * meas * m * sizeof(float2)) + rnd256((size_t)batch * m * m * sizeof(float2)) +
           rnd256((size_t)batch * m * sizeof(float2)) + 2 * rnd256((size_t)batch * meas * sizeof(float2)) +
           3 * rnd256((size_t)batch * m * sizeof(int)) + rnd256(batch * sizeof(int));
}

}  // namespace jstsp

using namespace jstsp;

extern "C" {

int jstsp_omp_c32(jstsp_ctx *ctx, int meas, int size_d, int batch, const jstsp_c32 *A_, long long strideA,
                  const jstsp_c32 *v_, int m, jstsp_c32 *x_hat, int32_t *index_out, jstsp_c32 *target_out,
                  int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(A_ && v_ && x_hat && index_out, JSTSP_E_NULL, "OMP: NULL array argument");
    JSTSP_REQUIRE(meas > 0 && size_d > 0 && batch > 0 && m > 0, JSTSP_E_SHAPE, "OMP: bad shape");
    JSTSP_REQUIRE(m <= 1024, JSTSP_E_UNSUPPORTED, "OMP: m = %d > 1024", m);
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_REQUIRE(strideA == 0 || strideA >= (long long)meas * size_d, JSTSP_E_SHAPE, "strideA too small");
    JSTSP_HIP(hipSetDevice(ctx->device));
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)meas * size_d : (size_t)meas * size_d;
    size_t need = omp_bytes(meas, m, batch) + rnd256((size_t)batch * size_d * sizeof(float2)) * 2 +
                  rnd256((size_t)batch * m * sizeof(int32_t)) + rnd256((size_t)batch * meas * m * sizeof(float2));
    if (memspace == JSTSP_HOST) need += rnd256(szA * sizeof(float2)) + rnd256((size_t)batch * meas * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *A, *v;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), szA, memspace, &A));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(v_), (size_t)batch * meas, memspace, &v));
    OmpState s;
    JSTSP_TRY(omp_alloc(ctx->arena, s, meas, m, batch));
    float2 *corr = ctx->arena.get<float2>((size_t)batch * size_d);
    float2 *xh = ctx->arena.get<float2>((size_t)batch * size_d);
    int32_t *io = ctx->arena.get<int32_t>((size_t)batch * m);
    float2 *to = target_out ? ctx->arena.get<float2>((size_t)batch * meas * m) : nullptr;
    JSTSP_REQUIRE(corr && xh && io && (!target_out || to), JSTSP_E_NOMEM, "OMP: workspace exhausted");
    hipStream_t st = ctx->stream;
    JSTSP_HIP(hipMemcpyAsync(s.r, v, (size_t)batch * meas * sizeof(float2), hipMemcpyDeviceToDevice, st));   // r = v (:10)
    JSTSP_HIP(hipMemsetAsync(s.nu, 0, batch * sizeof(int), st));
    JSTSP_HIP(hipMemsetAsync(s.Rm, 0, (size_t)batch * m * m * sizeof(float2), st));
    for (int it = 0; it < m; ++it) {                                                                           // :16
        // A'*r (:17).  Shared dictionary: one GEMM with the residuals of all problems as columns.
        if (strideA == 0)
            JSTSP_TRY(gemm(ctx, 'C', 'N', size_d, batch, meas, 1, Mat{A, 0, meas}, Mat{s.r, 0, meas}, corr, 0,
                           size_d));
        else
            JSTSP_TRY(gemm(ctx, 'C', 'N', size_d, 1, meas, batch, Mat{A, strideA, meas},
                           Mat{s.r, (long long)meas, meas}, corr, (long long)size_d, size_d));
        hipLaunchKernelGGL(omp_step_kernel, dim3(batch), dim3(256), 0, st, meas, size_d, m, it, corr, A, strideA,
                           (const float2 *)nullptr, 0ll, 0, 0, 0, s);
    }
    hipLaunchKernelGGL(omp_finish_kernel, dim3(batch), dim3(256), m * sizeof(float2), st, meas, size_d, m, A,
                       strideA, (const float2 *)nullptr, 0ll, 0, 0, 0, s, xh, io, to);
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(x_hat), xh, (size_t)batch * size_d, memspace));
    JSTSP_TRY(stage_out(ctx, index_out, io, (size_t)batch * m, memspace));
    if (target_out) JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(target_out), to, (size_t)batch * meas * m, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(st));
    return 0;
}

int jstsp_omp_kron_c32(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *Af_,
                       long long strideA, const jstsp_c32 *Bf_, long long strideB, const jstsp_c32 *y_, int m,
                       jstsp_c32 *x_hat, int32_t *index_out, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(Af_ && Bf_ && y_ && x_hat && index_out, JSTSP_E_NULL, "omp_kron: NULL array argument");
    JSTSP_REQUIRE(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0 && m > 0, JSTSP_E_SHAPE, "omp_kron: bad shape");
    JSTSP_REQUIRE(m <= 1024, JSTSP_E_UNSUPPORTED, "omp_kron: m = %d > 1024", m);
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_HIP(hipSetDevice(ctx->device));
    const int meas = N * M, size_d = Gr * G2;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t szB = strideB ? (size_t)strideB * (batch - 1) + (size_t)G2 * M : (size_t)G2 * M;
    const size_t ng = (size_t)N * G2;
    size_t need = omp_bytes(meas, m, batch) + 2 * rnd256((size_t)batch * size_d * sizeof(float2)) +
                  rnd256((size_t)batch * ng * sizeof(float2)) + rnd256((size_t)batch * m * sizeof(int32_t));
    if (memspace == JSTSP_HOST)
        need += rnd256(szA * sizeof(float2)) + rnd256(szB * sizeof(float2)) + rnd256((size_t)batch * meas * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *Af, *Bf, *y;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Af_), szA, memspace, &Af));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Bf_), szB, memspace, &Bf));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(y_), (size_t)batch * meas, memspace, &y));
    OmpState s;
    JSTSP_TRY(omp_alloc(ctx->arena, s, meas, m, batch));
    float2 *corr = ctx->arena.get<float2>((size_t)batch * size_d);
    float2 *xh = ctx->arena.get<float2>((size_t)batch * size_d);
    float2 *Tc = ctx->arena.get<float2>((size_t)batch * ng);
    int32_t *io = ctx->arena.get<int32_t>((size_t)batch * m);
    JSTSP_REQUIRE(corr && xh && Tc && io, JSTSP_E_NOMEM, "omp_kron: workspace exhausted");
    hipStream_t st = ctx->stream;
    JSTSP_HIP(hipMemcpyAsync(s.r, y, (size_t)batch * meas * sizeof(float2), hipMemcpyDeviceToDevice, st));
    JSTSP_HIP(hipMemsetAsync(s.nu, 0, batch * sizeof(int), st));
    JSTSP_HIP(hipMemsetAsync(s.Rm, 0, (size_t)batch * m * m * sizeof(float2), st));
    for (int it = 0; it < m; ++it) {
        // Phi'*r = vec(Af^H R Bf^H) with R = reshape(r, N, M): the correlation kernel of the hot path
        JSTSP_TRY(gemm(ctx, 'N', 'C', N, G2, M, batch, Mat{s.r, (long long)meas, N}, Mat{Bf, strideB, G2}, Tc,
                       (long long)ng, N, 1.f, nullptr, 0, 0, 0.f, GEMM_CORRELATE));
        JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, N, batch, Mat{Af, strideA, N}, Mat{Tc, (long long)ng, N}, corr,
                       (long long)size_d, Gr));
        hipLaunchKernelGGL(omp_step_kernel, dim3(batch), dim3(256), 0, st, meas, size_d, m, it, corr, Af, strideA,
                           Bf, strideB, N, Gr, G2, s);
    }
    hipLaunchKernelGGL(omp_finish_kernel, dim3(batch), dim3(256), m * sizeof(float2), st, meas, size_d, m, Af,
                       strideA, Bf, strideB, N, Gr, G2, s, xh, io, (float2 *)nullptr);
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(x_hat), xh, (size_t)batch * size_d, memspace));
    JSTSP_TRY(stage_out(ctx, index_out, io, (size_t)batch * m, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
