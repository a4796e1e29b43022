// Batched inverse of Hermitian positive-definite Gram matrices — the least-squares branch of
// proposed_algorithm ('std', proposed_algorithm.m:29,53): v = U\(L\k) with [L,U] = lu(K2) is the
// least-squares solution K2^+ k; with K2 = kron(B.', A) of full column rank
//     K2^+ k = vec( (A^H A)^-1 A^H K B^H (B B^H)^-1 ),
// so the LU of the (N*M) x (Gr*G2) Kronecker matrix becomes two small Hermitian inverses.
//   n <= 128 : eigen-decomposition (Jacobi, eig.hip)  G^-1 = U diag(1/lambda) U^H, eigenvalues below
//              n*eps*lambda_max dropped (pinv semantics at fp32 resolution)
//   n  > 128 : Newton-Schulz  X <- X (2I - G X),  X0 = I / ||G||_1, all on the MFMA GEMM, 24 steps, then the
//              residual max|I - G X| is measured
// This is the route for factors too large for the float64 pinv kernel (pinv.hip), which every shape of the
// reference's own drivers takes instead.  Accuracy here is cond(G) * eps_fp32: lambda_min/lambda_max and the
// Newton-Schulz residual go to the context's conditioning record (jstsp_last_conditioning), and JSTSP_HOST calls
// fail with JSTSP_E_ILLCOND instead of returning digits that are not there.
#include "solver_common.h"
#include <algorithm>

namespace jstsp {

// T = U * diag(1/lam)   (column scaling).  Eigenvalues of the fp32 Gram below n*eps*lam_max (and non-positive ones)
// are noise: their components are dropped, as pinv drops singular values below its tolerance.  The ratio
// lam_min/lam_max of each matrix is folded into the context's conditioning record.
__global__ __launch_bounds__(256) void scale_cols_inv_kernel(int n, const float2 *U, const float *lam, float2 *T,
                                                             uint32_t *rcond_min_bits)
{
    const int t = blockIdx.y;
    const long long base = (long long)t * n * n;
    const float *l = lam + (long long)t * n;
    float lmax = 0.f, lmin = 3.4e38f;
    for (int i = 0; i < n; ++i) { lmax = fmaxf(lmax, l[i]); lmin = fminf(lmin, l[i]); }
    const float cut = (float)n * 1.1920929e-7f * lmax;
    if (blockIdx.x == 0 && threadIdx.x == 0 && rcond_min_bits)
        atomicMin(rcond_min_bits, __float_as_uint(lmax > 0.f ? fmaxf(lmin, 0.f) / lmax : 0.f));
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n; e += gridDim.x * 256) {
        const float lv = l[e / n];
        const float s = lv > cut ? 1.f / lv : 0.f;
        const float2 u = U[base + e];
        T[base + e] = make_float2(u.x * s, u.y * s);
    }
}

// res = max over the batch of max_ij |delta_ij - P_ij|   (P = G X after Newton-Schulz), float bits via atomicMax
__global__ __launch_bounds__(256) void ns_residual_kernel(int n, const float2 *P, uint32_t *res_max_bits)
{
    const int t = blockIdx.y;
    const long long base = (long long)t * n * n;
    float m = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n; e += gridDim.x * 256) {
        const float2 p = P[base + e];
        m = fmaxf(m, fmaxf(fabsf(((e % n == e / n) ? 1.f : 0.f) - p.x), fabsf(p.y)));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m == m) atomicMax(res_max_bits, __float_as_uint(m));
    if ((threadIdx.x & 63) == 0 && m != m) atomicMax(res_max_bits, 0x7f800000u);      // NaN -> +inf
}

// X0 = I / ||G||_1   (one workgroup per matrix)
__global__ __launch_bounds__(256) void ns_init_kernel(int n, const float2 *G, float2 *X)
{
    __shared__ float sh[4];
    const int t = blockIdx.x;
    const float2 *g = G + (long long)t * n * n;
    float best = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) { const float2 v = g[i + (long long)n * j]; s += sqrtf(v.x * v.x + v.y * v.y); }
        best = fmaxf(best, s);
    }
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
    __syncthreads();
    const float nrm = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float2 *x = X + (long long)t * n * n;
    for (int e = threadIdx.x; e < n * n; e += 256) x[e] = make_float2((e % n == e / n) ? 1.f / nrm : 0.f, 0.f);
}

// P <- 2I - P
__global__ __launch_bounds__(256) void two_i_minus_kernel(int n, float2 *P)
{
    const int t = blockIdx.y;
    const long long base = (long long)t * n * n;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n; e += gridDim.x * 256) {
        float2 p = P[base + e];
        p.x = ((e % n == e / n) ? 2.f : 0.f) - p.x;
        p.y = -p.y;
        P[base + e] = p;
    }
}

size_t hinv_bytes(int n, int count)
{
    const size_t nn = (size_t)n * n;
    const int ne = (n + 1) & ~1;
    size_t b = 2 * rnd256(count * nn * sizeof(float2)) + rnd256((size_t)count * n * sizeof(float));
    if (n <= 128 && eig_needs_global_v(n)) b += rnd256((size_t)count * ne * ne * sizeof(float2));
    return b;
}

// Ginv[t] = G[t]^-1 for `count` Hermitian positive-definite n x n matrices.
int hermitian_inverse(jstsp_ctx *ctx, int n, int count, const float2 *G, float2 *Ginv)
{
    const size_t nn = (size_t)n * n;
    Arena &a = ctx->arena;
    JSTSP_TRY(ensure_diag(ctx));
    float2 *T1 = a.get<float2>(count * nn), *T2 = a.get<float2>(count * nn);
    float *lam = a.get<float>((size_t)count * n);
    JSTSP_REQUIRE(T1 && T2 && lam, JSTSP_E_NOMEM, "hermitian_inverse: workspace exhausted");
    const long long s = (long long)nn;
    const dim3 grid((unsigned)std::min<size_t>((nn + 255) / 256, 64), (unsigned)count);
    if (n <= 128) {
        float2 *Vg = nullptr;
        if (eig_needs_global_v(n)) {
            const int ne = (n + 1) & ~1;
            Vg = a.get<float2>((size_t)count * ne * ne);
            JSTSP_REQUIRE(Vg, JSTSP_E_NOMEM, "hermitian_inverse: workspace exhausted");
        }
        JSTSP_TRY(launch_eig(ctx, EIG_VECS, n, count, G, s, 1, 0, nullptr, nullptr, T1, lam, Vg));   // T1 = U
        hipLaunchKernelGGL(scale_cols_inv_kernel, grid, dim3(256), 0, ctx->stream, n, T1, lam, T2, ctx->diag + 2);   // T2 = U / lam
        JSTSP_HIP(hipGetLastError());
        return gemm(ctx, 'N', 'C', n, n, n, count, Mat{T2, s, n}, Mat{T1, s, n}, Ginv, s, n);           // (U/lam) U^H
    }
    // Newton-Schulz: X_{k+1} = X_k (2I - G X_k); quadratic once ||I - G X|| < 1 (true from X0 for PD G)
    hipLaunchKernelGGL(ns_init_kernel, dim3(count), dim3(256), 0, ctx->stream, n, G, Ginv);
    float2 *X = Ginv, *Xn = T2;
    for (int it = 0; it < 24; ++it) {
        JSTSP_TRY(gemm(ctx, 'N', 'N', n, n, n, count, Mat{G, s, n}, Mat{X, s, n}, T1, s, n));           // T1 = G X
        hipLaunchKernelGGL(two_i_minus_kernel, grid, dim3(256), 0, ctx->stream, n, T1);                 // T1 = 2I - G X
        JSTSP_TRY(gemm(ctx, 'N', 'N', n, n, n, count, Mat{X, s, n}, Mat{T1, s, n}, Xn, s, n));          // Xn = X T1
        std::swap(X, Xn);
    }
    // what the fixed number of steps left: max |I - G X| over the batch, into the conditioning record
    JSTSP_TRY(gemm(ctx, 'N', 'N', n, n, n, count, Mat{G, s, n}, Mat{X, s, n}, T1, s, n));
    hipLaunchKernelGGL(ns_residual_kernel, grid, dim3(256), 0, ctx->stream, n, T1, ctx->diag + 1);
    JSTSP_HIP(hipGetLastError());
    if (X != Ginv) JSTSP_HIP(hipMemcpyAsync(Ginv, X, count * nn * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

}  // namespace jstsp
