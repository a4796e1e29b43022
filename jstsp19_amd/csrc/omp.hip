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

__device__ __forceinline__ double2 block_sum2_4w(double2 v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) { v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = v.x; sh[2 * (threadIdx.x >> 6) + 1] = v.y; }
    __syncthreads();
    return make_double2(sh[0] + sh[2] + sh[4] + sh[6], sh[1] + sh[3] + sh[5] + sh[7]);
}

// MANY problems (more than 64 per call): one workgroup of four waves per problem, modified Gram-Schmidt with block-wide
// reductions - at a batch that fills the chip what counts is instructions per problem, not the latency of one (batch 1024 at
// BASELINE configs[0]: 14.8 ms per call against 21.4 for the wave-parallel form below, which is 1.9x faster for ONE problem).
// One workgroup per problem: argmax |corr|, then append the atom.
// Dense dictionary: atom = A[:, idx] (A + t*strideA, meas x size_d).
// Kronecker (Bf != nullptr): atom[i + N*j] = Af[i, g] * Bf[h, j], idx = g + Gr*h.
__global__ __launch_bounds__(256) void omp_step_mgs_kernel(int meas, int size_d, int m, int it, const float2 *corr,
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
    nrm0 = block_sum2_4w(make_double2(nrm0, 0), sh).x;
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
            d = block_sum2_4w(d, sh);
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
    n2 = block_sum2_4w(make_double2(n2, 0), sh).x;
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
    d = block_sum2_4w(d, sh);
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

__device__ __forceinline__ double2 wave_sum2(double2 v)
{
    for (int o = 32; o > 0; o >>= 1) { v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); }
    return v;
}
template <int NT> __device__ __forceinline__ double2 block_sum2(double2 v, double *sh)
{
    constexpr int NW = NT / 64;
    v = wave_sum2(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = v.x; sh[2 * (threadIdx.x >> 6) + 1] = v.y; }
    __syncthreads();
    double2 r = make_double2(0, 0);
#pragma unroll
    for (int k = 0; k < NW; ++k) { r.x += sh[2 * k]; r.y += sh[2 * k + 1]; }
    return r;
}

// One workgroup of NT threads per problem: argmax |corr|, then append the atom.
// Dense dictionary: atom = A[:, idx] (A + t*strideA, meas x size_d).
// Kronecker (Bf != nullptr): atom[i + N*j] = Af[i, g] * Bf[h, j], idx = g + Gr*h.
// Round 5: the orthogonalisation is classical Gram-Schmidt done twice (CGS2) with the u inner products of a pass spread over
// the NT / 64 waves (wave-local float64 reductions, no barrier per inner product) and ONE update w -= Q d per pass - rounds
// 1-4 ran modified Gram-Schmidt, 2 u block-wide reductions with two barriers each, which was 45 of the 58 us an OMP iteration
// took at BASELINE configs[0] (a kernel boundary is 1.5 us: the launches were never the cost).  NT = 1024 for few problems.
template <int NT>
__global__ __launch_bounds__(NT) void omp_step_kernel(int meas, int size_d, int m, int it, const float2 *corr,
                                                      const float2 *A, long long strideA, const float2 *Bf,
                                                      long long strideB, int N, int Gr, int G2, OmpState s)
{
    constexpr int NW = NT / 64;
    __shared__ double sh[2 * NW];
    __shared__ float shv[NW];
    __shared__ int shi[NW];
    __shared__ int s_idx, s_dup;
    __shared__ float2 dsh[1024];                    // inner products of a pass (m <= 1024)
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- argmax of |corr| with first-index tie-break (MATLAB max) -------------------------------
    const float2 *c = corr + (long long)t * size_d;
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int i = tid; i < size_d; i += NT) {
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
    if (lane == 0) { shv[wave] = best; shi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < NW; ++k)
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
        for (int e = tid; e < meas; e += NT) {
            const float2 x = a[e % N], y = b[(long long)G2 * (e / N)];
            const float2 v = make_float2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
            w[e] = v;
            nrm0 += (double)v.x * v.x + (double)v.y * v.y;
        }
    } else {
        const float2 *a = A + (long long)t * strideA + (long long)meas * idx;
        for (int e = tid; e < meas; e += NT) {
            const float2 v = a[e];
            w[e] = v;
            nrm0 += (double)v.x * v.x + (double)v.y * v.y;
        }
    }
    nrm0 = block_sum2<NT>(make_double2(nrm0, 0), sh).x;
    for (int j = tid; j < m; j += NT) Rc[j] = make_float2(0.f, 0.f);
    __syncthreads();
    // ---- classical Gram-Schmidt against the basis, twice -----------------------------------------------
    for (int pass = 0; pass < 2; ++pass) {
        for (int j = wave; j < u; j += NW) {            // d_j = q_j^H w, one wave per j
            const float2 *q = Q + (long long)meas * j;
            double2 d = make_double2(0, 0);
            for (int e = lane; e < meas; e += 64) {
                const float2 qq = q[e], ww = w[e];
                d.x += (double)qq.x * ww.x + (double)qq.y * ww.y;       // conj(q) * w
                d.y += (double)qq.x * ww.y - (double)qq.y * ww.x;
            }
            d = wave_sum2(d);
            if (lane == 0) dsh[j] = make_float2((float)d.x, (float)d.y);
        }
        __syncthreads();
        for (int e = tid; e < meas; e += NT) {
            float2 ww = w[e];
            for (int j = 0; j < u; ++j) {
                const float2 qq = Q[(long long)meas * j + e], h = dsh[j];
                ww.x -= h.x * qq.x - h.y * qq.y;
                ww.y -= h.x * qq.y + h.y * qq.x;
            }
            w[e] = ww;
        }
        for (int j = tid; j < u; j += NT) { Rc[j].x += dsh[j].x; Rc[j].y += dsh[j].y; }
        __syncthreads();
    }
    double n2 = 0;
    for (int e = tid; e < meas; e += NT) { const float2 ww = w[e]; n2 += (double)ww.x * ww.x + (double)ww.y * ww.y; }
    n2 = block_sum2<NT>(make_double2(n2, 0), sh).x;
    if (!(n2 > 1e-12 * nrm0)) return;              // numerically dependent on the chosen atoms: adds nothing
    const float inv = (float)(1.0 / sqrt(n2));
    // ---- q_u = w/|w|, z_u = q_u^H v = q_u^H r (r is orthogonal to the old basis), r -= z_u q_u --------
    float2 *qu = Q + (long long)meas * u;
    double2 d = make_double2(0, 0);
    for (int e = tid; e < meas; e += NT) {
        const float2 ww = w[e], rr = r[e];
        const float2 qv = make_float2(ww.x * inv, ww.y * inv);
        qu[e] = qv;
        d.x += (double)qv.x * rr.x + (double)qv.y * rr.y;
        d.y += (double)qv.x * rr.y - (double)qv.y * rr.x;
    }
    d = block_sum2<NT>(d, sh);
    const float zx = (float)d.x, zy = (float)d.y;
    for (int e = tid; e < meas; e += NT) {
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

// The same step with the working vector out of global memory (meas <= 1024 EPT, EPT = 1 or 2): thread tid owns elements
// e = tid + 1024 k of the candidate w and of the residual r in registers, a copy of w sits in LDS for the wave-parallel inner
// products, the re-selection test is one compare per thread instead of a serial scan by thread 0 (u dependent global loads of
// ~0.7 us each), the loads of a basis column are issued eight at a time, and column u of R is written once.  Same arithmetic in
// the same order as omp_step_kernel<1024>: the same atoms, x_hat equal to an ulp (the compiler contracts the updates differently).
template <int EPT>
__global__ __launch_bounds__(1024) void omp_step_reg_kernel(int meas, int size_d, int m, int it, const float2 *corr,
                                                            const float2 *A, long long strideA, const float2 *Bf,
                                                            long long strideB, int N, int Gr, int G2, OmpState s, int qc)
{
    constexpr int NT = 1024, NW = 16;
    extern __shared__ float2 qsh[];                 // the first qc basis columns (one workgroup reads Q through one CU's 64 B per
                                                    // clock: four sweeps over 24 columns were 5 us of an iteration)
    __shared__ double sh[2 * NW];
    __shared__ float shv[NW];
    __shared__ int shi[NW];
    __shared__ int s_dup;
    __shared__ float2 dsh[2][1024];                 // inner products of the two passes (m <= 1024)
    __shared__ float2 wsh[NT * EPT];
    __shared__ double shn[NW];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = s.nu[t];
    float2 *r = s.r + (long long)t * meas;
    float2 rv[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) { const int e = tid + NT * k; rv[k] = e < meas ? r[e] : make_float2(0.f, 0.f); }   // used last
    const float2 *Q = s.Qb + (long long)t * meas * m;
    const int uc = min(u, qc);
    // ---- argmax of |corr| with first-index tie-break (MATLAB max) -------------------------------
    // Load order matters (loads return in order, a wait on one waits for all before it): the correlations first, then the first
    // PRE basis columns into registers - independent of the atom chosen below, they arrive under the argmax and the atom fetch
    // and go to LDS after that (issued as a fill loop up here they cost 2 us per 8 columns before the argmax could start).
    const float2 *c = corr + (long long)t * size_d;
    const float2 c0 = tid < size_d ? c[tid] : make_float2(0.f, 0.f);
    constexpr int PRE = 16 / EPT;
    float2 qpre[EPT][PRE];
    if (uc > 0) {
#pragma unroll
        for (int k = 0; k < EPT; ++k)
#pragma unroll
            for (int i = 0; i < PRE; ++i) qpre[k][i] = Q[(long long)meas * min(i, uc - 1) + min(tid + NT * k, meas - 1)];
    }
    float best = -1.f;
    int bi = 0x7fffffff;
    if (tid < size_d) {
        float a = sqrtf(c0.x * c0.x + c0.y * c0.y);
        if (a != a) a = -1.f;
        best = a; bi = tid;
    }
    for (int i = tid + NT; i < size_d; i += NT) {
        const float2 v = c[i];
        float a = sqrtf(v.x * v.x + v.y * v.y);
        if (a != a) a = -1.f;
        if (a > best) { best = a; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { shv[wave] = best; shi[wave] = bi; }
    if (tid == 0) s_dup = 0x7fffffff;
    __syncthreads();
    best = shv[0]; bi = shi[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) {
        const float ob = shv[k];
        const int oi = shi[k];
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 0;
    const int idx = bi;
    for (int j = tid; j < u; j += NT)
        if (s.uniq[(long long)t * m + j] == idx) atomicMin(&s_dup, j);
    // ---- load the atom (whether or not it turns out to be a re-selection: the loads overlap the test) ------------------
    float2 wv[EPT];
    double nrm0 = 0;
    if (Bf) {
        const int g = idx % Gr, h = idx / Gr;
        const float2 *a = A + (long long)t * strideA + (long long)N * g;            // Af(:, g)
        const float2 *b = Bf + (long long)t * strideB + h;                          // Bf(h, :) stride G2
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + NT * k;
            float2 v = make_float2(0.f, 0.f);
            if (e < meas) {
                const float2 x = a[e % N], y = b[(long long)G2 * (e / N)];
                v = make_float2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
            }
            wv[k] = v;
        }
    } else {
        const float2 *a = A + (long long)t * strideA + (long long)meas * idx;
#pragma unroll
        for (int k = 0; k < EPT; ++k) { const int e = tid + NT * k; wv[k] = e < meas ? a[e] : make_float2(0.f, 0.f); }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        wsh[tid + NT * k] = wv[k];
        nrm0 += (double)wv[k].x * wv[k].x + (double)wv[k].y * wv[k].y;
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + NT * k;
        if (e < meas) {
#pragma unroll
            for (int i = 0; i < PRE; ++i)
                if (i < uc) qsh[i * meas + e] = qpre[k][i];
            for (int j0 = PRE; j0 < uc; j0 += 8) {          // (more columns than registers: short problems, meas < 1024)
                float2 qq[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) qq[i] = Q[(long long)meas * min(j0 + i, uc - 1) + e];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (j0 + i < uc) qsh[(j0 + i) * meas + e] = qq[i];
            }
        }
    }
    nrm0 = wave_sum2(make_double2(nrm0, 0)).x;
    if (lane == 0) shn[wave] = nrm0;                // summed when it is needed (the dependence test below)
    __syncthreads();                                // publishes wsh, s_dup, shn
    const int dup = s_dup;
    if (dup != 0x7fffffff) {                        // re-selected atom: span (and residual) unchanged
        if (tid == 0) { s.sel[(long long)t * m + it] = idx; s.mult[(long long)t * m + dup] += 1; }
        return;
    }
    // ---- classical Gram-Schmidt against the basis, twice -----------------------------------------------
    for (int pass = 0; pass < 2; ++pass) {
        for (int j = wave; j < u; j += NW) {            // d_j = q_j^H w, one wave per j
            const float2 *q = Q + (long long)meas * j;
            double2 d = make_double2(0, 0);
            if (j < uc) {
                const float2 *ql = qsh + j * meas;
                for (int e = lane; e < meas; e += 64) {
                    const float2 qq = ql[e], ww = wsh[e];
                    d.x += (double)qq.x * ww.x + (double)qq.y * ww.y;
                    d.y += (double)qq.x * ww.y - (double)qq.y * ww.x;
                }
            } else
            for (int e0 = lane; e0 < meas; e0 += 64 * 8) {              // eight loads in flight (the plain loop waited for each)
                float2 qq[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) qq[i] = q[min(e0 + 64 * i, meas - 1)];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + 64 * i;
                    if (e < meas) {
                        const float2 ww = wsh[e];
                        d.x += (double)qq[i].x * ww.x + (double)qq[i].y * ww.y;       // conj(q) * w
                        d.y += (double)qq[i].x * ww.y - (double)qq[i].y * ww.x;
                    }
                }
            }
            d = wave_sum2(d);
            if (lane == 0) dsh[pass][j] = make_float2((float)d.x, (float)d.y);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + NT * k;
            if (e < meas) {
                float2 ww = wv[k];
                for (int j = 0; j < uc; ++j) {
                    const float2 qq = qsh[j * meas + e], h = dsh[pass][j];
                    ww.x -= h.x * qq.x - h.y * qq.y;
                    ww.y -= h.x * qq.y + h.y * qq.x;
                }
                for (int j0 = uc; j0 < u; j0 += 8) {
                    float2 qq[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) qq[i] = Q[(long long)meas * min(j0 + i, u - 1) + e];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (j0 + i < u) {
                            const float2 h = dsh[pass][j0 + i];
                            ww.x -= h.x * qq[i].x - h.y * qq[i].y;
                            ww.y -= h.x * qq[i].y + h.y * qq[i].x;
                        }
                }
                wv[k] = ww;
                wsh[e] = ww;
            }
        }
        __syncthreads();
    }
    double n2 = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) n2 += (double)wv[k].x * wv[k].x + (double)wv[k].y * wv[k].y;
    n2 = block_sum2<NT>(make_double2(n2, 0), sh).x;
    if (tid == 0) s.sel[(long long)t * m + it] = idx;
    nrm0 = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) nrm0 += shn[k];
    if (!(n2 > 1e-12 * nrm0)) return;              // numerically dependent on the chosen atoms: adds nothing
    const float inv = (float)(1.0 / sqrt(n2));
    // ---- q_u = w/|w|, z_u = q_u^H v = q_u^H r (r is orthogonal to the old basis), r -= z_u q_u --------
    float2 *qu = s.Qb + (long long)t * meas * m + (long long)meas * u;
    double2 d = make_double2(0, 0);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + NT * k;
        const float2 qv = make_float2(wv[k].x * inv, wv[k].y * inv);
        wv[k] = qv;
        if (e < meas) qu[e] = qv;
        d.x += (double)qv.x * rv[k].x + (double)qv.y * rv[k].y;
        d.y += (double)qv.x * rv[k].y - (double)qv.y * rv[k].x;
    }
    d = block_sum2<NT>(d, sh);
    const float zx = (float)d.x, zy = (float)d.y;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + NT * k;
        float2 rr = rv[k];
        rr.x -= zx * wv[k].x - zy * wv[k].y;
        rr.y -= zx * wv[k].y + zy * wv[k].x;
        if (e < meas) r[e] = rr;
    }
    float2 *Rc = s.Rm + (long long)t * m * m + (long long)u * m;       // column u of R (zero so far)
    for (int j = tid; j < u; j += NT) {
        float2 a = make_float2(0.f, 0.f);
        a.x += dsh[0][j].x; a.y += dsh[0][j].y;
        a.x += dsh[1][j].x; a.y += dsh[1][j].y;
        Rc[j] = a;
    }
    if (tid == 0) {
        Rc[u] = make_float2((float)sqrt(n2), 0.f);
        s.z[(long long)t * m + u] = make_float2(zx, zy);
        s.uniq[(long long)t * m + u] = idx;
        s.mult[(long long)t * m + u] = 1;
        s.nu[t] = u + 1;
    }
}

// JSTSP_OMP_REG=0: the step through global memory (omp_step_kernel<1024>) also where the register form applies
static bool omp_reg_step()
{
    const char *e = xp_getenv("JSTSP_OMP_REG");        // (read at every call)
    return !e || atoi(e) != 0;
}

// basis columns omp_step_reg_kernel keeps in LDS: what 160 KiB leave beside its static arrays (inner products 16 KiB, w 8 EPT KiB)
// (-1: the opt-in to that much dynamic LDS was refused - the caller then takes omp_step_kernel<1024>, which needs none)
static int omp_reg_qcols(int meas, int m)
{
    if (hipFuncSetAttribute((const void *)omp_step_reg_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 138240) != hipSuccess ||
        hipFuncSetAttribute((const void *)omp_step_reg_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 130048) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    const int budget = meas <= 1024 ? 138240 : 130048;
    const int qc = budget / (meas * (int)sizeof(float2));
    return qc < m ? qc : m;
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

// ---- OMP on the Kronecker dictionary entirely in the coefficient domain ("Batch-OMP") --------------------------
// With Phi = kron(Bf.', Af) the Gram of the dictionary factorises,
//   <atom(a', b'), atom(a, b)> = G_A[a', a] * G_B[b, b'],   G_A = Af^H Af,  G_B = Bf Bf^H,
// so the correlation of the residual never needs the measurement space again after c0 = Phi^H v:
//   Phi^H r = c0 - G(:, U) x_U,   x_U = argmin |v - Phi_U x| = (G_UU)^-1 c0_U  (Cholesky, grown by one row per atom).
// One workgroup per problem runs ALL m iterations (OMP.m:16-24): argmax with first-index tie-break, duplicate
// check (multiplicities, :18 never excludes an atom), Cholesky append in fp64, two triangular solves, and the
// refresh of c (size_d x |U| complex MACs).  No per-iteration launches, no meas-sized traffic.
template <int NT>
__global__ __launch_bounds__(NT) void omp_gram_kernel(int size_d, int m, int Gr, int G2, const float2 *c0_, float2 *cw_,
                                                      const float2 *GA_, long long sGA, const float2 *GB_, long long sGB,
                                                      float2 *x_hat, int32_t *index_out)
{
    extern __shared__ double lds[];
    // LDS: L (m*m double2, row-major lower), xu (m double2), cu (m double2), work (m double2), ia/ib/mult (3*m int)
    double2 *L = reinterpret_cast<double2 *>(lds);
    double2 *xu = L + (size_t)m * m, *cu = xu + m, *wk = cu + m;
    int *ia = reinterpret_cast<int *>(wk + m), *ib = ia + m, *mult = ib + m;
    __shared__ float shv[NT / 64];
    __shared__ int shi[NT / 64];
    __shared__ int s_nu, s_new;
    const int t = blockIdx.x, tid = threadIdx.x;
    const float2 *c0 = c0_ + (long long)t * size_d;
    float2 *cw = cw_ + (long long)t * size_d;
    const float2 *GA = GA_ + (long long)t * sGA, *GB = GB_ + (long long)t * sGB;
    if (tid == 0) s_nu = 0;
    for (int i = tid; i < size_d; i += NT) cw[i] = c0[i];
    __syncthreads();
    for (int it = 0; it < m; ++it) {
        // ---- argmax |c| with first-index tie-break (MATLAB max)                                      OMP.m:17
        float best = -1.f;
        int bi = 0x7fffffff;
        for (int i = tid; i < size_d; i += NT) {
            const float2 v = cw[i];
            float a = sqrtf(v.x * v.x + v.y * v.y);
            if (a != a) a = -1.f;
            if (a > best) { best = a; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((tid & 63) == 0) { shv[tid >> 6] = best; shi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < NT / 64; ++k)
                if (shv[k] > best || (shv[k] == best && shi[k] < bi)) { best = shv[k]; bi = shi[k]; }
            if (bi == 0x7fffffff) bi = 0;
            index_out[(long long)t * m + it] = bi + 1;                     // 1-based indexSet
            const int a = bi % Gr, b = bi / Gr;
            int dup = -1;
            const int nu = s_nu;
            for (int j = 0; j < nu; ++j)
                if (ia[j] == a && ib[j] == b) { dup = j; break; }
            s_new = 0;
            if (dup >= 0) {
                mult[dup] += 1;                                            // re-selected atom: span unchanged
            } else {
                // ---- append the atom to the Cholesky factor of G_UU (fp64)
                const float2 gaa = GA[a + (long long)Gr * a], gbb = GB[b + (long long)G2 * b];
                const double gnn = (double)gaa.x * gbb.x;
                double w2 = 0;
                for (int j = 0; j < nu; ++j) {
                    // G[U_j, new] = G_A[ia_j, a] * G_B[b, ib_j]
                    const float2 p = GA[ia[j] + (long long)Gr * a], q = GB[b + (long long)G2 * ib[j]];
                    double gx = (double)p.x * q.x - (double)p.y * q.y, gy = (double)p.x * q.y + (double)p.y * q.x;
                    for (int k = 0; k < j; ++k) {                          // forward substitution L w = g
                        const double2 l = L[(size_t)j * m + k], wv = wk[k];
                        gx -= l.x * wv.x - l.y * wv.y;
                        gy -= l.x * wv.y + l.y * wv.x;
                    }
                    const double d = L[(size_t)j * m + j].x;
                    wk[j] = make_double2(gx / d, gy / d);
                    w2 += wk[j].x * wk[j].x + wk[j].y * wk[j].y;
                }
                const double d2 = gnn - w2;
                if (d2 > 1e-12 * gnn) {                                    // else: numerically in the span, adds nothing
                    for (int j = 0; j < nu; ++j) L[(size_t)nu * m + j] = make_double2(wk[j].x, -wk[j].y);   // row = w^H
                    L[(size_t)nu * m + nu] = make_double2(sqrt(d2), 0.0);
                    ia[nu] = a; ib[nu] = b; mult[nu] = 1;
                    const float2 cv = c0[bi];
                    cu[nu] = make_double2(cv.x, cv.y);
                    s_nu = nu + 1;
                    s_new = 1;
                    // ---- x_U = (L L^H)^-1 c0_U                                                     OMP.m:20
                    const int n = nu + 1;
                    for (int i = 0; i < n; ++i) {                          // L y = c0_U
                        double yx = cu[i].x, yy = cu[i].y;
                        for (int k = 0; k < i; ++k) {
                            const double2 l = L[(size_t)i * m + k], yv = wk[k];
                            yx -= l.x * yv.x - l.y * yv.y;
                            yy -= l.x * yv.y + l.y * yv.x;
                        }
                        const double d = L[(size_t)i * m + i].x;
                        wk[i] = make_double2(yx / d, yy / d);
                    }
                    for (int i = n - 1; i >= 0; --i) {                     // L^H x = y
                        double xx = wk[i].x, xy = wk[i].y;
                        for (int k = i + 1; k < n; ++k) {
                            const double2 l = L[(size_t)k * m + i], xv = xu[k];     // conj(L[k][i])
                            xx -= l.x * xv.x + l.y * xv.y;
                            xy -= l.x * xv.y - l.y * xv.x;
                        }
                        const double d = L[(size_t)i * m + i].x;
                        xu[i] = make_double2(xx / d, xy / d);
                    }
                }
            }
        }
        __syncthreads();
        if (!s_new) continue;                                             // nothing changed: same c, same argmax next time
        // ---- c = c0 - sum_u x_u G_A[:, ia_u] (x) G_B[ib_u, :]                                       (= Phi^H r, :21-22)
        const int nu = s_nu;
        for (int i = tid; i < size_d; i += NT) {
            const int a = i % Gr, b = i / Gr;
            const float2 cv = c0[i];
            double cx = cv.x, cy = cv.y;
            for (int u = 0; u < nu; ++u) {
                const float2 p = GA[a + (long long)Gr * ia[u]], q = GB[ib[u] + (long long)G2 * b];
                const double gx = (double)p.x * q.x - (double)p.y * q.y, gy = (double)p.x * q.y + (double)p.y * q.x;
                cx -= gx * xu[u].x - gy * xu[u].y;
                cy -= gx * xu[u].y + gy * xu[u].x;
            }
            cw[i] = make_float2((float)cx, (float)cy);
        }
        __syncthreads();
    }
    // ---- x_hat(indexSet) = x  (a re-selected atom: pinv splits the coefficient equally, the last copy stays, :29-32)
    for (int i = tid; i < size_d; i += NT) x_hat[(long long)t * size_d + i] = make_float2(0.f, 0.f);
    __syncthreads();
    if (tid == 0)
        for (int j = 0; j < s_nu; ++j) {
            const double mu = (double)mult[j];
            x_hat[(long long)t * size_d + ia[j] + (long long)Gr * ib[j]] =
                make_float2((float)(xu[j].x / mu), (float)(xu[j].y / mu));
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

// ---- corr = A' * r (OMP.m:17) for FEW right-hand sides per dictionary: one wave per atom, lanes stride over the measurements
//      (coalesced along the column), fp32 partial sums per lane, float64 wave-shuffle reduction.  A dictionary of its own per
//      problem, or a shared one with a handful of problems, is a matrix-vector product: HBM-bound on 8 * meas * size_d bytes per
//      iteration, where the MFMA GEMM with a one-column operand ran at 16 GB/s (0.51 ms per iteration
//      at BASELINE configs[0]: profiles/r04_cfg1_omp_kernel_stats.csv, before).
__global__ __launch_bounds__(256) void omp_corr_gemv_kernel(int meas, int size_d, const float2 *A, long long strideA,
                                                            const float2 *r, float2 *corr)
{
    const int t = blockIdx.y, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= size_d) return;
    const float2 *a = A + (long long)t * strideA + (long long)j * meas, *rv = r + (long long)t * meas;
    float sr = 0.f, si = 0.f;
    double dr = 0.0, di = 0.0;
    int cnt = 0;
    for (int i = lane; i < meas; i += 64) {
        const float2 x = a[i], y = rv[i];
        sr = fmaf(x.x, y.x, fmaf(x.y, y.y, sr));            // conj(a) r
        si = fmaf(x.x, y.y, fmaf(-x.y, y.x, si));
        if (++cnt == 32) { dr += sr; di += si; sr = 0.f; si = 0.f; cnt = 0; }      // fp32 chains of at most 32 terms
    }
    dr += sr; di += si;
    for (int o = 32; o > 0; o >>= 1) { dr += __shfl_xor(dr, o); di += __shfl_xor(di, o); }
    if (lane == 0) corr[(long long)t * size_d + j] = make_float2((float)dr, (float)di);
}


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
    JSTSP_ENTER(ctx);
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
    bool reg_step = batch <= 64 && meas <= 2048 && omp_reg_step();
    int qc = reg_step ? omp_reg_qcols(meas, m) : 0;
    if (qc < 0) { reg_step = false; qc = 0; }
    for (int it = 0; it < m; ++it) {                                                             // :16
        // A'*r (:17).  Few right-hand sides per dictionary: the matrix-vector kernel above.  Shared dictionary and many problems:
        // one GEMM with the residuals of all problems as columns.
        if (strideA != 0 || batch <= 16)
            hipLaunchKernelGGL(omp_corr_gemv_kernel, dim3((size_d + 3) / 4, batch), dim3(256), 0, st, meas, size_d, A, strideA, s.r,
                               corr);
        else if (strideA == 0)
            JSTSP_TRY(gemm(ctx, 'C', 'N', size_d, batch, meas, 1, Mat{A, 0, meas}, Mat{s.r, 0, meas}, corr, 0,
                           size_d));
        else
            JSTSP_TRY(gemm(ctx, 'C', 'N', size_d, 1, meas, batch, Mat{A, strideA, meas},
                           Mat{s.r, (long long)meas, meas}, corr, (long long)size_d, size_d));
        if (reg_step && meas <= 1024)        // few problems: sixteen waves per problem, w and r in registers
            hipLaunchKernelGGL(omp_step_reg_kernel<1>, dim3(batch), dim3(1024), (size_t)qc * meas * sizeof(float2), st, meas, size_d, m, it, corr, A, strideA,
                               (const float2 *)nullptr, 0ll, 0, 0, 0, s, qc);
        else if (reg_step)
            hipLaunchKernelGGL(omp_step_reg_kernel<2>, dim3(batch), dim3(1024), (size_t)qc * meas * sizeof(float2), st, meas, size_d, m, it, corr, A, strideA,
                               (const float2 *)nullptr, 0ll, 0, 0, 0, s, qc);
        else if (batch <= 64)
            hipLaunchKernelGGL(omp_step_kernel<1024>, dim3(batch), dim3(1024), 0, st, meas, size_d, m, it, corr, A, strideA,
                               (const float2 *)nullptr, 0ll, 0, 0, 0, s);
        else
            hipLaunchKernelGGL(omp_step_mgs_kernel, dim3(batch), dim3(256), 0, st, meas, size_d, m, it, corr, A, strideA,
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
    JSTSP_ENTER(ctx);
    {
        // Coefficient-domain OMP (omp_gram_kernel): one correlation, the two factor Grams, one kernel for all m
        // iterations.  JSTSP_OMP_GRAM=0 keeps the measurement-space Gram-Schmidt below (also used when the
        // Cholesky factor does not fit in LDS).
        const size_t lds = (size_t)m * m * 16 + 3 * (size_t)m * 16 + 3 * (size_t)m * 4;
        if (tune().omp_gram != 0 && lds <= 150 * 1024) {
            const int size_d = Gr * G2, nA = strideA ? batch : 1, nB = strideB ? batch : 1;
            const size_t nm = (size_t)N * M, ng = (size_t)N * G2, g = (size_t)size_d;
            const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
            const size_t szB = strideB ? (size_t)strideB * (batch - 1) + (size_t)G2 * M : (size_t)G2 * M;
            const bool h2 = use_hgemm(N, G2, M);
            size_t need = 3 * rnd256(batch * g * sizeof(float2)) + rnd256(batch * ng * sizeof(float2)) +
                          rnd256((size_t)nA * Gr * Gr * sizeof(float2)) + rnd256((size_t)nB * G2 * G2 * sizeof(float2)) +
                          rnd256((size_t)batch * m * sizeof(int32_t));
            if (h2) need += hgemm_pack_bytes(M, G2, nB) + rnd256(batch * sizeof(uint32_t));
            if (memspace == JSTSP_HOST)
                need += rnd256(szA * sizeof(float2)) + rnd256(szB * sizeof(float2)) + rnd256(batch * nm * sizeof(float2));
            JSTSP_TRY(ctx->arena.reserve(need));
            ctx->arena.reset();
            const float2 *Af, *Bf, *y;
            JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Af_), szA, memspace, &Af));
            JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Bf_), szB, memspace, &Bf));
            JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(y_), batch * nm, memspace, &y));
            Arena &ar = ctx->arena;
            float2 *c0 = ar.get<float2>(batch * g), *cw = ar.get<float2>(batch * g), *xh = ar.get<float2>(batch * g);
            float2 *Tc = ar.get<float2>(batch * ng);
            float2 *GA = ar.get<float2>((size_t)nA * Gr * Gr), *GB = ar.get<float2>((size_t)nB * G2 * G2);
            int32_t *io = ar.get<int32_t>((size_t)batch * m);
            JSTSP_REQUIRE(c0 && cw && xh && Tc && GA && GB && io, JSTSP_E_NOMEM, "omp_kron: workspace exhausted");
            const Mat Am{Af, strideA, N}, Bm{Bf, strideB, G2}, Ym{y, (long long)nm, N};
            JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, Gr, N, nA, Am, Am, GA, (long long)Gr * Gr, Gr));                 // G_A = Af^H Af
            if (h2) {
                HPack pk;
                JSTSP_TRY(hgemm_pack(ctx, pk, ar, Bf, strideB, G2, 1, 1, M, G2, nB, (long long)G2 * M));
                HGemmDesc hb{Bf, strideB, G2, pk.bmax, pk.data, pk.st, pk.bmax, 1, pk.KS, pk.JT, GB,
                             (long long)G2 * G2, G2, G2, G2, M, nB, EPI_NONE, nullptr, nullptr, nullptr};
                JSTSP_TRY(launch_hgemm(ctx, hb, nullptr));                                                      // G_B = Bf Bf^H
                uint32_t *ymax = ar.get<uint32_t>(batch);
                JSTSP_REQUIRE(ymax, JSTSP_E_NOMEM, "omp_kron: workspace exhausted");
                JSTSP_TRY(hgemm_absmax(ctx, y, (long long)nm, (long long)nm, batch, ymax));
                HGemmDesc hc{y, (long long)nm, N, ymax, pk.data, strideB ? pk.st : 0, pk.bmax, strideB ? 1 : 0, pk.KS,
                             pk.JT, Tc, (long long)ng, N, N, G2, M, batch, EPI_NONE, nullptr, nullptr, nullptr};
                JSTSP_TRY(launch_hgemm(ctx, hc, "correlate"));                                                  // Y Bf^H
            } else {
                JSTSP_TRY(gemm(ctx, 'N', 'C', G2, G2, M, nB, Bm, Bm, GB, (long long)G2 * G2, G2));
                JSTSP_TRY(gemm(ctx, 'N', 'C', N, G2, M, batch, Ym, Bm, Tc, (long long)ng, N, 1.f, nullptr, 0, 0, 0.f,
                               GEMM_CORRELATE));
            }
            JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, N, batch, Am, Mat{Tc, (long long)ng, N}, c0, (long long)g, Gr));  // Phi^H v
            // per launch, like every other kernel: the attribute is per device and a process may hold contexts on several
            JSTSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(omp_gram_kernel<1024>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            hipLaunchKernelGGL(omp_gram_kernel<1024>, dim3(batch), dim3(1024), lds, ctx->stream, size_d, m, Gr, G2, c0, cw,
                               GA, strideA ? (long long)Gr * Gr : 0, GB, strideB ? (long long)G2 * G2 : 0, xh, io);
            JSTSP_HIP(hipGetLastError());
            JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(x_hat), xh, batch * g, memspace));
            JSTSP_TRY(stage_out(ctx, index_out, io, (size_t)batch * m, memspace));
            if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
            return 0;
        }
    }
    const int meas = N * M, size_d = Gr * G2;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t szB = strideB ? (size_t)strideB * (batch - 1) + (size_t)G2 * M : (size_t)G2 * M;
    const size_t ng = (size_t)N * G2;
    size_t need = omp_bytes(meas, m, batch) + 2 * rnd256((size_t)batch * size_d * sizeof(float2)) +
                  rnd256((size_t)batch * ng * sizeof(float2)) + rnd256((size_t)batch * m * sizeof(int32_t));
    const bool h2 = use_hgemm(N, G2, M);         // split-f16 correlation, dictionary factor packed once (hgemm.hip)
    const int nB = strideB ? batch : 1;
    if (h2) need += hgemm_pack_bytes(M, G2, nB) + rnd256(batch * sizeof(uint32_t));
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
    HPack pk;
    uint32_t *rmax = nullptr;
    if (h2) {
        JSTSP_TRY(hgemm_pack(ctx, pk, ctx->arena, Bf, strideB, G2, 1, 1, M, G2, nB, (long long)G2 * M));
        rmax = ctx->arena.get<uint32_t>(batch);
        JSTSP_REQUIRE(rmax, JSTSP_E_NOMEM, "omp_kron: workspace exhausted");
    }
    bool reg_step = batch <= 64 && meas <= 2048 && omp_reg_step();
    int qc = reg_step ? omp_reg_qcols(meas, m) : 0;
    if (qc < 0) { reg_step = false; qc = 0; }
    for (int it = 0; it < m; ++it) {
        // Phi'*r = vec(Af^H R Bf^H) with R = reshape(r, N, M): the correlation kernel of the hot path
        if (h2) {
            JSTSP_TRY(hgemm_absmax(ctx, s.r, (long long)meas, (long long)meas, batch, rmax));
            HGemmDesc hd{s.r, (long long)meas, N, rmax, pk.data, strideB ? pk.st : 0, pk.bmax, strideB ? 1 : 0, pk.KS,
                         pk.JT, Tc, (long long)ng, N, N, G2, M, batch, EPI_NONE, nullptr, nullptr, nullptr};
            JSTSP_TRY(launch_hgemm(ctx, hd, "correlate"));
        } else
        JSTSP_TRY(gemm(ctx, 'N', 'C', N, G2, M, batch, Mat{s.r, (long long)meas, N}, Mat{Bf, strideB, G2}, Tc,
                       (long long)ng, N, 1.f, nullptr, 0, 0, 0.f, GEMM_CORRELATE));
        JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, N, batch, Mat{Af, strideA, N}, Mat{Tc, (long long)ng, N}, corr,
                       (long long)size_d, Gr));
        if (reg_step && meas <= 1024)
            hipLaunchKernelGGL(omp_step_reg_kernel<1>, dim3(batch), dim3(1024), (size_t)qc * meas * sizeof(float2), st, meas, size_d, m, it, corr, Af, strideA,
                           Bf, strideB, N, Gr, G2, s, qc);
        else if (reg_step)
            hipLaunchKernelGGL(omp_step_reg_kernel<2>, dim3(batch), dim3(1024), (size_t)qc * meas * sizeof(float2), st, meas, size_d, m, it, corr, Af, strideA,
                           Bf, strideB, N, Gr, G2, s, qc);
        else if (batch <= 64)
            hipLaunchKernelGGL(omp_step_kernel<1024>, dim3(batch), dim3(1024), 0, st, meas, size_d, m, it, corr, Af, strideA,
                           Bf, strideB, N, Gr, G2, s);
        else
            hipLaunchKernelGGL(omp_step_mgs_kernel, dim3(batch), dim3(256), 0, st, meas, size_d, m, it, corr, Af, strideA,
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
