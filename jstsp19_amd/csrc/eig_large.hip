// Hermitian eigen-decomposition for orders above 128 - beyond what the LDS-resident Jacobi kernels hold (eig.hip,
// eig2.hip, eig3.hip).  Nothing on the timed paths gets here: these are the one-off decompositions of large factors
// (VAMP's `svd` of vamp.m:32 when G2 = L*Gt > 128: BASELINE configs[4] has order 4096) and the SVT / spectral norms of
// inputs whose BOTH dimensions exceed 128 (benchmark_algorithms/svt.m:5).
//
// Round 3: hand-written two-sided BLOCK Jacobi (rounds 1-2 called rocSOLVER's cheevd here).  The matrix is cut into
// blocks of 64 rows / columns (order padded to an even number of blocks; the padding is decoupled: zero off-diagonal,
// distinct negative diagonal, so no rotation ever touches it).  A round pairs the blocks two by two (circle method,
// nb - 1 rounds meet every pair once = one sweep); each pair's 128 x 128 Hermitian sub-matrix gets one sweep (round 4;
// solved to convergence before) of the existing order-128 Jacobi kernel (eig3.hip) - all pairs of all matrices in one launch - and the 128 x 128
// unitaries J are applied as batched GEMMs on the fp32 MFMA kernel (cgemm.hip):
//      W <- J^H W J   as   X = W J,  W' = (X^H) J        (two column-panel products and one conjugate transposition:
//                                                          panels of 128 columns are contiguous, rows are not)
//      U <- U J
// Between rounds the blocks are physically permuted (rows and columns of W, columns of U) so that the partners of the
// next round are neighbours and every panel product is ONE strided-batched launch over (matrix, pair).  Block Jacobi
// with exactly solved sub-problems converges quadratically like the scalar method; a sweep costs 3 * n^3 complex MACs.
// The basis is then re-orthonormalised (one Newton-Schulz step) and the eigenvalues taken as Rayleigh quotients against the
// original matrix.  Measured (MI355X): order 4096, 1 matrix: 12 sweeps, 1.3 s (round 3: 13 sweeps, 2.8 s; rocSOLVER cheevd, rounds 1-2: 0.5 s); svt of
// 160 x 192 ... 700 x 520 inputs within 3e-5 of the float64 oracle (tests/test_gpu_large_orders.py).
#include "common.h"
#include "solver_common.h"

#include <vector>

namespace jstsp {
namespace {

constexpr int BS = 64;           // block size; sub-problems have order 2 * BS = 128

// Wp[t] = padded Hermitian part of sum_s Gpart[t][s] (order np >= n), U[t] = I.  Padding: zero coupling, diagonal
// -(1 + k/np) * dscale[t] (negative and distinct: G is positive semi-definite in every caller, and no rotation mixes them;
// inside [-2, -1] x the largest diagonal entry so that the sub-problems' convergence scale stays that of the data).
__global__ void init_kernel(int n, int np, const float2 *Gpart, long long sGt, int nsplit, long long sGs, float2 *W, float2 *U,
                            const float *dscale)
{
    const int t = blockIdx.y;
    const float2 *g = Gpart + (long long)t * sGt;
    float2 *w = W + (size_t)t * np * np, *u = U ? U + (size_t)t * np * np : nullptr;
    const float ds = dscale[t];
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)np * np; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % np), j = (int)(e / np);
        float2 v = make_float2(0.f, 0.f);
        if (i < n && j < n) {
            float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
            for (int s = 0; s < nsplit; ++s) {
                const float2 x = g[(long long)s * sGs + i + (long long)n * j], y = g[(long long)s * sGs + j + (long long)n * i];
                a.x += x.x; a.y += x.y; b.x += y.x; b.y += y.y;
            }
            v = (i == j) ? make_float2(a.x, 0.f) : make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
        } else if (i == j) {
            v = make_float2(-(1.f + (float)(i - n) / (float)np) * ds, 0.f);
        }
        w[e] = v;
        if (u) u[e] = make_float2(i == j ? 1.f : 0.f, 0.f);
    }
}

// dscale[t] = 1 + largest diagonal entry (of the sum of the partials); stat[t] = {sum |w_ij|^2 off the diagonal, sum w_ii^2}
__global__ void dscale_kernel(int n, const float2 *Gpart, long long sGt, int nsplit, long long sGs, float *dscale)
{
    const int t = blockIdx.x;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float d = 0.f;
        for (int s = 0; s < nsplit; ++s) d += Gpart[(long long)t * sGt + (long long)s * sGs + i + (long long)n * i].x;
        m = fmaxf(m, fabsf(d));
    }
    __shared__ float sh[256];
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) dscale[t] = 1.f + sh[0];
}

__global__ void offnorm_kernel(int np, const float2 *W, double *stat)
{
    const int t = blockIdx.y;
    const float2 *w = W + (size_t)t * np * np;
    double off = 0.0, dg = 0.0;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)np * np; e += (long long)gridDim.x * blockDim.x) {
        const float2 v = w[e];
        const double a = (double)v.x * v.x + (double)v.y * v.y;
        if (e % np == e / np) dg += a; else off += a;
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); dg += __shfl_xor(dg, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stat[2 * t], off); atomicAdd(&stat[2 * t + 1], dg); }
}

// Wn[i, j] = W[idx[i], idx[j]];  Un[:, j] = U[:, idx[j]]   (symmetric block permutation between two rounds)
__global__ void permute_kernel(int np, const int *idx, const float2 *W, float2 *Wn, const float2 *U, float2 *Un)
{
    const int t = blockIdx.y;
    const size_t o = (size_t)t * np * np;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)np * np; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % np), j = (int)(e / np);
        const int sj = idx[j];
        if (W) Wn[o + e] = W[o + idx[i] + (size_t)np * sj];
        if (U) Un[o + e] = U[o + i + (size_t)np * sj];
    }
}

// S[(t, k)] = the k-th diagonal block of order 2 BS of W[t]
__global__ void gather_diag_kernel(int np, int m, const float2 *W, float2 *S)
{
    const int b = blockIdx.x, t = b / m, k = b % m;
    const float2 *w = W + (size_t)t * np * np + (size_t)(2 * BS * k) * np + 2 * BS * k;
    float2 *s = S + (size_t)b * (2 * BS) * (2 * BS);
    for (int e = threadIdx.x; e < 4 * BS * BS; e += blockDim.x)
        s[e] = w[(e % (2 * BS)) + (size_t)np * (e / (2 * BS))];
}

// Xh[t] = X[t]^H (32 x 32 tiles through LDS)
__global__ void conj_transpose_kernel(int np, const float2 *X, float2 *Xh)
{
    __shared__ float2 tile[32][33];
    const int t = blockIdx.z;
    const float2 *x = X + (size_t)t * np * np;
    float2 *y = Xh + (size_t)t * np * np;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) tile[r][threadIdx.x] = x[(i0 + threadIdx.x) + (size_t)np * (j0 + r)];      // tile[j][i]
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const float2 v = tile[threadIdx.x][r];                  // element (i0 + r, j0 + tx) of X
        y[(j0 + threadIdx.x) + (size_t)np * (i0 + r)] = make_float2(v.x, -v.y);
    }
}

// Results in the callers' order: eigenpair of physical column j belongs to logical index lg[j]; padded ones (lg >= n) dropped.
__global__ void extract_kernel(int n, int np, const int *lg, const float2 *W, const float2 *U, float2 *Q, float *lam)
{
    const int t = blockIdx.y;
    const size_t o = (size_t)t * np * np;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)n * np; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % n), j = (int)(e / n);
        const int c = lg[j];
        if (c >= n) continue;
        if (Q) Q[(size_t)t * n * n + i + (size_t)n * c] = U[o + i + (size_t)np * j];
        if (i == 0 && lam) lam[(size_t)t * n + c] = W[o + j + (size_t)np * j].x;
    }
}

// T <- 1.5 I - 0.5 T   (one Newton-Schulz step towards U (U^H U)^(-1/2))
__global__ void ns_factor_kernel(int n, float2 *T)
{
    const int t = blockIdx.y;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * blockDim.x) {
        float2 v = T[(size_t)t * n * n + e];
        v.x = ((e % n == e / n) ? 1.5f : 0.f) - 0.5f * v.x;
        v.y = -0.5f * v.y;
        T[(size_t)t * n * n + e] = v;
    }
}

// lam[t][c] = Re(u_c^H (G u_c)): Rayleigh quotients of the columns of U with respect to the ORIGINAL matrix
__global__ void rayleigh_kernel(int n, const float2 *U, const float2 *GU, float *lam)
{
    const int t = blockIdx.y, c = blockIdx.x;
    const float2 *u = U + (size_t)t * n * n + (size_t)n * c, *g = GU + (size_t)t * n * n + (size_t)n * c;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)u[i].x * g[i].x + (double)u[i].y * g[i].y;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) lam[(size_t)t * n + c] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ void lmax_of_kernel(int n, int batch, const float *lam, float *out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch) return;
    float m = lam[(size_t)t * n];
    for (int i = 1; i < n; ++i) m = fmaxf(m, lam[(size_t)t * n + i]);
    out[t] = m;
}

// T[t](:, i) = U[t](:, i) * q_i,  q_i = min(1, tau_t / sqrt(max(lambda_i, 0)))  (1 for a zero singular value: svt.m:7-12
// then yields Q = I for the all-zero input, as the small kernels do)
__global__ void scale_cols_kernel(int n, const float2 *U, const float *lam, const TrialParams *prm, const float *tau, float2 *T)
{
    const int t = blockIdx.y;
    const float tv = tau ? tau[t] : prm[t].tauY_rho;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / n);
        const float sig = sqrtf(fmaxf(lam[(size_t)t * n + i], 0.f));
        const float q = (sig > 0.f) ? fminf(1.f, tv / sig) : 1.f;
        const float2 u = U[(size_t)t * n * n + e];
        T[(size_t)t * n * n + e] = make_float2(q * u.x, q * u.y);
    }
}

// Look-ahead of the block Jacobi (launch_eig_large): Wc = the permuted matrix of this round (pair k = rows / columns
// 128 k .. 128 k + 127), J = this round's rotations.  After W' = J^H Wc J the next round pairs the blocks
// (src[2 k'], src[2 k' + 1]) (block numbers of THIS round's order); block (u, v) of that pair's sub-matrix is
//      W'[bu, bv] = J_p(:, 64 hu ..)^H  Wc[128 p .., 128 q ..]  J_q(:, 64 hv ..),     bu = 2 p + hu, bv = 2 q + hv.
// One workgroup of 256 threads per 16 columns of a block (blockIdx.x = 8 u + 4 v + column slice): T = Wc[p, q] J_q(:, slice)
// (128 x 16: 4 x 2 outputs per thread, operands straight from memory - the rotation entries are wave-uniform), then
// J_p(:, half)^H T (64 x 16, 4 per thread; T from LDS, the rotation half through LDS in chunks of 16 rows).  24 KiB of LDS: the
// kernel has to find room on compute units that the panel products of the main stream are filling.  fp32 FMA chains of 128
// terms, like the panel products themselves.
__global__ __launch_bounds__(256) void lookahead_kernel(int np, int m, const int *src, const float2 *W, const float2 *J, float2 *S)
{
    __shared__ __attribute__((aligned(16))) float2 T[128 * 16];        // T[k][c]
    __shared__ __attribute__((aligned(16))) float2 Jc[16 * 64];        // Jc[k - k0][i] = J_p(k, 64 hu + i)
    __builtin_amdgcn_s_setprio(3);         // (on the critical chain, beside panel products that fill the chip)
    const int u = blockIdx.x >> 3, v = (blockIdx.x >> 2) & 1, sl = blockIdx.x & 3;
    const int b = blockIdx.y, t = b / m, kn = b % m, tid = threadIdx.x;
    const int bu = src[2 * kn + u], bv = src[2 * kn + v];
    const int p = bu >> 1, hu = bu & 1, q = bv >> 1, hv = bv & 1;
    const float2 *w = W + (size_t)t * np * np + (size_t)(128 * q) * np + 128 * p;                   // Wc[128 p + i, 128 q + j]
    const float2 *jq = J + ((size_t)t * m + q) * 128 * 128 + (size_t)(64 * hv + 16 * sl) * 128;     // J_q(k, 64 hv + 16 sl + c) = jq[k + 128 c]
    const float2 *jp = J + ((size_t)t * m + p) * 128 * 128 + (size_t)(64 * hu) * 128;
    {
        const int rg = tid & 31, cg = __builtin_amdgcn_readfirstlane(tid >> 6) * 2 + ((tid >> 5) & 1);     // rows 4 rg .., columns 2 cg ..
        float2 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = make_float2(0.f, 0.f); acc[i][1] = make_float2(0.f, 0.f); }
#pragma unroll 8
        for (int k = 0; k < 128; ++k) {
            const float4 w01 = *reinterpret_cast<const float4 *>(&w[4 * rg + (size_t)np * k]);
            const float4 w23 = *reinterpret_cast<const float4 *>(&w[4 * rg + 2 + (size_t)np * k]);
            const float2 a[4] = {make_float2(w01.x, w01.y), make_float2(w01.z, w01.w), make_float2(w23.x, w23.y), make_float2(w23.z, w23.w)};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float2 x = jq[k + 128 * (2 * cg + c)];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][c].x = fmaf(a[i].x, x.x, fmaf(-a[i].y, x.y, acc[i][c].x));
                    acc[i][c].y = fmaf(a[i].x, x.y, fmaf(a[i].y, x.x, acc[i][c].y));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { T[(4 * rg + i) * 16 + 2 * cg] = acc[i][0]; T[(4 * rg + i) * 16 + 2 * cg + 1] = acc[i][1]; }
    }
    const int i = tid & 63, jg = tid >> 6;                  // row i, columns 4 jg ..
    float2 o[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    for (int k0 = 0; k0 < 128; k0 += 16) {
        __syncthreads();                                    // (T complete / the previous chunk consumed)
        for (int e = tid; e < 16 * 64; e += 256) Jc[(e & 15) * 64 + (e >> 4)] = jp[k0 + (e & 15) + 128 * (e >> 4)];
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float2 a = Jc[kk * 64 + i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {                   // conj(a) x
                const float2 x = T[(k0 + kk) * 16 + 4 * jg + c];
                o[c].x = fmaf(a.x, x.x, fmaf(a.y, x.y, o[c].x));
                o[c].y = fmaf(a.x, x.y, fmaf(-a.y, x.x, o[c].y));
            }
        }
    }
    float2 *s = S + (size_t)b * 128 * 128;
#pragma unroll
    for (int c = 0; c < 4; ++c) s[(64 * u + i) + 128 * (64 * v + 16 * sl + 4 * jg + c)] = o[c];
}

// stream-ordered temporaries outside the context's arena (the callers sized that for the small kernels), freed on every path
struct Temps {
    hipStream_t st, also[4] = {nullptr, nullptr, nullptr, nullptr};      // `also`: further streams that work on these buffers (drained before they are freed)
    std::vector<void *> p;
    explicit Temps(hipStream_t s) : st(s) {}
    ~Temps()
    {
        for (hipStream_t a : also) if (a) (void)hipStreamSynchronize(a);
        for (void *q : p) (void)hipFreeAsync(q, st);
    }
    template <class T> T *get(size_t n)
    {
        void *q = nullptr;
        if (hipMallocAsync(&q, std::max<size_t>(n, 1) * sizeof(T), st) != hipSuccess) return nullptr;
        p.push_back(q);
        return static_cast<T *>(q);
    }
};

}  // namespace

int launch_eig_large(jstsp_ctx *ctx, int mode, int n, int batch, const float2 *Gpart, long long sGt, int nsplit,
                     long long sGs, const TrialParams *prm, const float *tau, float2 *Q, float *lam_out)
{
    JSTSP_REQUIRE(mode == EIG_VECS || mode == EIG_SVT_Q || mode == EIG_LMAX, JSTSP_E_ARG, "eig (order %d): bad mode %d", n, mode);
    JSTSP_REQUIRE(n <= 8192, JSTSP_E_UNSUPPORTED, "eig: matrix order %d above 8192", n);
    JSTSP_REQUIRE(batch <= 32768, JSTSP_E_UNSUPPORTED, "eig (order %d): %d matrices in one call", n, batch);
    hipStream_t st = ctx->stream;
    int nb = (n + BS - 1) / BS;
    nb += nb & 1;                                   // an even number of blocks
    const int np = nb * BS, m = nb / 2, sub = 2 * BS;
    const size_t nn = (size_t)np * np, cnt = (size_t)batch * m;
    const bool vecs = mode != EIG_LMAX;
    Temps tmp(st);
    float2 *W = tmp.get<float2>(batch * nn), *Wp = tmp.get<float2>(batch * nn);
    float2 *U = vecs ? tmp.get<float2>(batch * nn) : nullptr, *Up = vecs ? tmp.get<float2>(batch * nn) : nullptr;
    float2 *S = tmp.get<float2>(cnt * sub * sub), *J = tmp.get<float2>(2 * cnt * sub * sub);
    float *lamJ = tmp.get<float>(cnt * sub), *dscale = tmp.get<float>(batch), *lam = tmp.get<float>((size_t)batch * n);
    double *stat = tmp.get<double>(2 * (size_t)batch);
    int *idx = tmp.get<int>(np), *lgd = tmp.get<int>(np);
    JSTSP_REQUIRE(W && Wp && (!vecs || (U && Up)) && S && J && lamJ && dscale && lam && stat && idx && lgd, JSTSP_E_NOMEM,
                  "eig (order %d, %d matrices): out of device memory", n, batch);
    const dim3 grid((unsigned)std::min<size_t>((nn + 255) / 256, 4096), (unsigned)batch);
    hipLaunchKernelGGL(dscale_kernel, dim3(batch), dim3(256), 0, st, n, Gpart, sGt, nsplit, sGs, dscale);
    hipLaunchKernelGGL(init_kernel, grid, dim3(256), 0, st, n, np, Gpart, sGt, nsplit, sGs, W, U, dscale);

    // relative block permutation between two rounds (circle method on the slots a_k = 2k, b_k = 2k + 1; a_0 stays):
    // new a_1 <- b_0, new a_k <- a_{k-1} (k >= 2), new b_k <- b_{k+1} (k <= m - 2), new b_{m-1} <- a_{m-1}
    std::vector<int> src(nb), hidx(np), lg(np), lg2(np);
    for (int s = 0; s < nb; ++s) src[s] = s;
    if (m > 1) {
        src[2] = 1;
        for (int k = 2; k < m; ++k) src[2 * k] = 2 * (k - 1);
        for (int k = 0; k + 1 < m; ++k) src[2 * k + 1] = 2 * (k + 1) + 1;
        src[2 * (m - 1) + 1] = 2 * (m - 1);
    }
    for (int i = 0; i < np; ++i) { hidx[i] = src[i / BS] * BS + i % BS; lg[i] = i; }
    JSTSP_TRY(upload(ctx, idx, hidx.data(), np * sizeof(int)));
    int *srcd = tmp.get<int>(nb);
    JSTSP_REQUIRE(srcd, JSTSP_E_NOMEM, "eig (order %d): out of device memory", n);
    JSTSP_TRY(upload(ctx, srcd, src.data(), nb * sizeof(int)));

    const int max_sweeps = 18;
    // ONE sweep of the scalar method inside each pair sub-problem per round: the unitary it returns is applied whatever it
    // achieved, so the outer iteration is a similarity transformation all the same - and it converges in the same number of
    // outer sweeps as with exactly solved sub-problems (order 4096: off/diag after each sweep equal to two digits, 12 sweeps
    // either way) while the sub-problem kernel, which runs on nb / 2 compute units only, takes 0.85 instead of 2.4 ms per round.
    const int inner_sweeps = 1;
    JSTSP_HIP(hipMemsetAsync(lamJ, 0, cnt * sub * sizeof(float), st));
    bool polished = false, restarted = false, converged = false;
    double last_worst = 0.0;
    const long long sPanel = (long long)sub * np, sSub = (long long)sub * sub;
    double prev = -1.0;
    // Three streams (nb > 2).  The sub-problem kernel holds nb / 2 compute units only and the panel products all of them, so
    // the chain of sub-problems is taken off the main stream:
    //   main : permutation of W, the two panel products W <- J^H W J
    //   sj   : sub-problems of round r, then the LOOK-AHEAD: the diagonal blocks of the NEXT round's pairs formed directly
    //          from the (permuted) W of this round and its rotations (lookahead_kernel: 2 x 2 blocks of 64 x 64, each
    //          J_p(:, half)^H W[p, q] J_q(:, half) - a few small products), so that the sub-problems of round r + 1 run beside
    //          the panel products of round r
    //   su   : the basis update U <- U J (not read until the iteration ends or restarts)
    // The rotations alternate between two buffers: round r + 2 overwrites J[r & 1] only after the panel products (main, implied
    // by the permutation the look-ahead waits for) and the basis update (ev_u) of round r have read it.
    const bool piped = m > 1;
    // With compute-unit masks (runtime.hip: ensure_cu_streams) the sub-problems own 32 units and everything else - the panel
    // products included, which is why the "main" work leaves the caller's stream for the length of the iteration - is kept off
    // them: a 1024-thread workgroup with 140 KiB of LDS otherwise waits for a unit that the panel products have drained by chance.
    float2 *Tm = (np > 1024 && vecs) ? tmp.get<float2>(batch * nn) : nullptr;       // (the restart's intermediate product)
    JSTSP_REQUIRE(!(np > 1024 && vecs) || Tm, JSTSP_E_NOMEM, "eig (order %d): out of device memory", n);
    hipStream_t sp = st, su = st, sj = st, sl = st;
    hipEvent_t ev_j = nullptr, ev_u[2] = {nullptr, nullptr}, ev_perm = nullptr, ev_la = nullptr, ev_io = nullptr;
    if (piped) {
        JSTSP_TRY(ensure_bj_resources(ctx));
        // (only when the sub-problems of a round fit the reserved units one each; more of them want the whole chip)
        if (tune().bj_mask && cnt <= 32 && ensure_cu_streams(ctx)) { sp = ctx->cu_stream[0]; su = ctx->cu_stream[1]; sl = ctx->cu_stream[2]; sj = ctx->cu_stream[3]; }
        else { su = ctx->bj_stream[0]; sj = sl = ctx->bj_stream[1]; }
        tmp.also[0] = su; tmp.also[1] = sj; tmp.also[2] = sl; tmp.also[3] = sp != st ? sp : nullptr;
        ev_j = ctx->bj_ev[0]; ev_u[0] = ctx->bj_ev[1]; ev_u[1] = ctx->bj_ev[2]; ev_perm = ctx->bj_ev[3]; ev_la = ctx->bj_ev[4]; ev_io = ctx->bj_ev[5];
        if (sp != st) { JSTSP_HIP(hipEventRecord(ev_io, st)); JSTSP_HIP(hipStreamWaitEvent(sp, ev_io, 0)); }
    }
    StreamScope main_sc(ctx, sp);
    long long round_no = 0;
    bool la_valid = false;                  // S holds the next round's diagonal blocks (look-ahead of the previous round)
    auto join_u = [&]() -> int {            // the main stream continues behind the last basis update
        if (piped && vecs && round_no > 0) JSTSP_HIP(hipStreamWaitEvent(sp, ev_u[(round_no - 1) & 1], 0));
        return 0;
    };
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        for (int r = 0; r < std::max(1, nb - 1); ++r) {
            if (!piped) {                           // two blocks: one sub-problem, nothing to overlap
                hipLaunchKernelGGL(gather_diag_kernel, dim3((unsigned)cnt), dim3(256), 0, sp, np, m, W, S);
                JSTSP_TRY(launch_eig128(ctx, sub, (int)cnt, S, sSub, 1, 0, nullptr, lamJ /* tau = 0 */, nullptr, J, 0, inner_sweeps));
                const Mat Jm{J, sSub, sub};
                JSTSP_TRY(gemm(ctx, 'N', 'N', np, sub, sub, (int)cnt, Mat{W, sPanel, np}, Jm, Wp, sPanel, np));           // X = W J
                hipLaunchKernelGGL(conj_transpose_kernel, dim3(np / 32, np / 32, batch), dim3(32, 8), 0, sp, np, Wp, W);
                JSTSP_TRY(gemm(ctx, 'N', 'N', np, sub, sub, (int)cnt, Mat{W, sPanel, np}, Jm, Wp, sPanel, np));           // W' = X^H J
                std::swap(W, Wp);
                if (vecs) {
                    JSTSP_TRY(gemm(ctx, 'N', 'N', np, sub, sub, (int)cnt, Mat{U, sPanel, np}, Jm, Up, sPanel, np));       // U' = U J
                    std::swap(U, Up);
                }
                ++round_no;
                continue;
            }
            float2 *Jr = J + (size_t)(round_no & 1) * cnt * sub * sub;
            const Mat Jm{Jr, sSub, sub};
            // main: partners of this round side by side
            hipLaunchKernelGGL(permute_kernel, grid, dim3(256), 0, sp, np, idx, W, Wp, (const float2 *)nullptr, (float2 *)nullptr);
            for (int i = 0; i < np; ++i) lg2[i] = lg[hidx[i]];
            lg.swap(lg2);
            JSTSP_HIP(hipEventRecord(ev_perm, sp));
            {   // sj: every pair's 128 x 128 sub-problem (eig3.hip's register-resident kernel, 1024 threads per matrix, basis only):
                // rotations J (columns); then the look-ahead
                StreamScope sc(ctx, sj);
                const bool gather = !la_valid;      // (first round, or the matrix has just been replaced)
                if (gather) {
                    JSTSP_HIP(hipStreamWaitEvent(sj, ev_perm, 0));
                    hipLaunchKernelGGL(gather_diag_kernel, dim3((unsigned)cnt), dim3(256), 0, sj, np, m, Wp, S);
                }
                if (vecs && round_no >= 2) JSTSP_HIP(hipStreamWaitEvent(sj, ev_u[round_no & 1], 0));
                JSTSP_TRY(launch_eig128(ctx, sub, (int)cnt, S, sSub, 1, 0, nullptr, lamJ /* tau = 0 */, nullptr, Jr, 0, inner_sweeps));
                JSTSP_HIP(hipEventRecord(ev_j, sj));
                if (sl != sj) JSTSP_HIP(hipStreamWaitEvent(sl, ev_j, 0));
                if (!gather || sl != sj) JSTSP_HIP(hipStreamWaitEvent(sl, ev_perm, 0));   // (the look-ahead reads this round's permuted W)
                hipLaunchKernelGGL(lookahead_kernel, dim3(16, (unsigned)cnt), dim3(256), 0, sl, np, m, srcd, Wp, Jr, S);
                JSTSP_HIP(hipEventRecord(ev_la, sl));
                if (sl != sj) JSTSP_HIP(hipStreamWaitEvent(sj, ev_la, 0));            // (the next sub-problems read S)
                la_valid = true;
            }
            if (vecs) {                             // su: U' = U J
                StreamScope sc(ctx, su);
                JSTSP_HIP(hipStreamWaitEvent(su, ev_j, 0));
                hipLaunchKernelGGL(permute_kernel, grid, dim3(256), 0, su, np, idx, (const float2 *)nullptr, (float2 *)nullptr, U, Up);
                JSTSP_TRY(gemm(ctx, 'N', 'N', np, sub, sub, (int)cnt, Mat{Up, sPanel, np}, Jm, U, sPanel, np));
                JSTSP_HIP(hipEventRecord(ev_u[round_no & 1], su));
            }
            // main: W' = J^H W J (the conjugate transposition overwrites the permuted W the look-ahead reads)
            JSTSP_HIP(hipStreamWaitEvent(sp, ev_j, 0));
            JSTSP_TRY(gemm(ctx, 'N', 'N', np, sub, sub, (int)cnt, Mat{Wp, sPanel, np}, Jm, W, sPanel, np));               // X = W J
            JSTSP_HIP(hipStreamWaitEvent(sp, ev_la, 0));
            hipLaunchKernelGGL(conj_transpose_kernel, dim3(np / 32, np / 32, batch), dim3(32, 8), 0, sp, np, W, Wp);
            JSTSP_TRY(gemm(ctx, 'N', 'N', np, sub, sub, (int)cnt, Mat{Wp, sPanel, np}, Jm, W, sPanel, np));               // W' = X^H J
            JSTSP_HIP(hipGetLastError());
            ++round_no;
        }
        // stop when the off-diagonal mass is at the fp32 level of the matrix, or no longer shrinking
        JSTSP_HIP(hipMemsetAsync(stat, 0, 2 * (size_t)batch * sizeof(double), sp));
        hipLaunchKernelGGL(offnorm_kernel, grid, dim3(256), 0, sp, np, W, stat);
        std::vector<double> hs(2 * (size_t)batch);
        JSTSP_HIP(hipMemcpyAsync(hs.data(), stat, hs.size() * sizeof(double), hipMemcpyDeviceToHost, sp));
        JSTSP_HIP(hipStreamSynchronize(sp));
        double worst = 0.0;
        for (int t = 0; t < batch; ++t) worst = std::max(worst, hs[2 * t + 1] > 0 ? std::sqrt(hs[2 * t] / hs[2 * t + 1]) : 0.0);
        last_worst = worst;
        if (tune().bj_trace) fprintf(stderr, "block Jacobi order %d (%d blocks): sweep %d off/diag %.3e\n", n, nb, sweep, worst);
        // (orders above 1024 do not wait for the stagnation: below 1e-5 the next sweep would only meet the noise floor)
        const bool early_restart = np > 1024 && vecs && !restarted && worst < 1e-5;
        if (worst < 3e-8 * std::sqrt((double)np) || (prev >= 0 && worst > 0.5 * prev && worst < 1e-5) || early_restart) {
            // Converged to the level the transformed matrix can reach: W has been through (nb - 1) x sweeps two-sided fp32
            // updates and its own rounding noise (off/diag about 2e-6 at order 4096) is what is left.  Orders above 1024:
            // restart ONCE from W = U^H G U formed from the ORIGINAL matrix - the accumulated noise is gone, the couplings
            // that remain are the true ones, and one or two more sweeps bring them to the level of a single sweep's rounding.
            if (np > 1024) {
                if (restarted || !vecs) { converged = true; break; }
                restarted = true;
                JSTSP_TRY(join_u());
                la_valid = false;                   // (the blocks formed ahead belong to the matrix that is being replaced)
                float2 *Gp = Wp;
                hipLaunchKernelGGL(init_kernel, grid, dim3(256), 0, sp, n, np, Gpart, sGt, nsplit, sGs, Gp, (float2 *)nullptr, dscale);
                JSTSP_TRY(gemm(ctx, 'N', 'N', np, np, np, batch, Mat{Gp, (long long)nn, np}, Mat{U, (long long)nn, np}, Tm, (long long)nn, np));
                JSTSP_TRY(gemm(ctx, 'C', 'N', np, np, np, batch, Mat{U, (long long)nn, np}, Mat{Tm, (long long)nn, np}, W, (long long)nn, np));
                prev = -1.0;
                continue;
            }
            // up to order 1024 one more sweep is cheap and settles the small eigenvalues
            if (polished) { converged = true; break; }
            polished = true;
        }
        prev = worst;
    }
    // No silent failure: a matrix whose couplings are still large when the sweeps run out has no usable basis (the Rayleigh
    // refinement below repairs eigenvalues, not eigenvectors).  A run that ends within 1e-4 of its diagonal without having met
    // the stop rule is accepted - the rule asks for the fp32 floor.
    JSTSP_TRY(join_u());
    if (sp != st) {                          // back on the caller's stream
        JSTSP_HIP(hipEventRecord(ev_io, sp));
        JSTSP_HIP(hipStreamWaitEvent(st, ev_io, 0));
        ctx->stream = st;
    }
    JSTSP_REQUIRE(converged || last_worst < 1e-4, JSTSP_E_ILLCOND,
                  "eig (order %d, %d matrices): block Jacobi did not converge in %d sweeps (off-diagonal / diagonal mass %.2e)", n,
                  batch, max_sweeps, last_worst);
    JSTSP_TRY(upload(ctx, lgd, lg.data(), np * sizeof(int)));
    float2 *Uout = nullptr;
    if (mode == EIG_VECS) Uout = Q;
    else if (mode == EIG_SVT_Q) { Uout = tmp.get<float2>((size_t)batch * n * n); JSTSP_REQUIRE(Uout, JSTSP_E_NOMEM, "eig: out of device memory"); }
    float *lout = (mode == EIG_VECS) ? lam_out : lam;
    const dim3 gx((unsigned)std::min<size_t>(((size_t)n * np + 255) / 256, 4096), (unsigned)batch);
    hipLaunchKernelGGL(extract_kernel, gx, dim3(256), 0, st, n, np, lgd, W, U, Uout, lout);
    int rcode = 0;
    if (vecs) {
        // The basis is a product of (nb - 1) x sweeps fp32 panel products and the diagonal of W has been through as many
        // two-sided updates: restore orthonormality (one Newton-Schulz step, U <- U (1.5 I - 0.5 U^H U)) and take the
        // eigenvalues as Rayleigh quotients u^H G u against the ORIGINAL matrix.  Three n^3 products.
        const long long s2 = (long long)n * n;
        const dim3 g2((unsigned)std::min<size_t>(((size_t)n * n + 255) / 256, 2048), (unsigned)batch);
        float2 *G0 = Wp, *T1 = W, *U2 = Up;                      // (the padded work arrays are free now: np >= n)
        hipLaunchKernelGGL(init_kernel, g2, dim3(256), 0, st, n, n, Gpart, sGt, nsplit, sGs, G0, (float2 *)nullptr, dscale);
        JSTSP_TRY(gemm(ctx, 'C', 'N', n, n, n, batch, Mat{Uout, s2, n}, Mat{Uout, s2, n}, T1, s2, n));
        hipLaunchKernelGGL(ns_factor_kernel, g2, dim3(256), 0, st, n, T1);
        JSTSP_TRY(gemm(ctx, 'N', 'N', n, n, n, batch, Mat{Uout, s2, n}, Mat{T1, s2, n}, U2, s2, n));
        JSTSP_HIP(hipMemcpyAsync(Uout, U2, (size_t)batch * s2 * sizeof(float2), hipMemcpyDeviceToDevice, st));
        JSTSP_TRY(gemm(ctx, 'N', 'N', n, n, n, batch, Mat{G0, s2, n}, Mat{Uout, s2, n}, T1, s2, n));
        hipLaunchKernelGGL(rayleigh_kernel, dim3(n, batch), dim3(256), 0, st, n, Uout, T1, lout);
    }
    if (mode == EIG_LMAX) {
        hipLaunchKernelGGL(lmax_of_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, n, batch, lam, lam_out);
    } else if (mode == EIG_SVT_Q) {
        float2 *T = tmp.get<float2>((size_t)batch * n * n);
        JSTSP_REQUIRE(T, JSTSP_E_NOMEM, "eig: out of device memory");
        const dim3 g2((unsigned)std::min<size_t>(((size_t)n * n + 255) / 256, 2048), (unsigned)batch);
        hipLaunchKernelGGL(scale_cols_kernel, g2, dim3(256), 0, st, n, Uout, lam, prm, tau, T);
        rcode = gemm(ctx, 'N', 'C', n, n, n, batch, Mat{T, (long long)n * n, n}, Mat{Uout, (long long)n * n, n}, Q, (long long)n * n, n);
    }
    JSTSP_TRY(rcode);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
