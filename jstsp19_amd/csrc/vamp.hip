// jstsp_vamp_kron_c32 / jstsp_vamp_c32 — benchmark_algorithms/vamp.m:1-55 driving
// MPbased_solvers/VAMP/VampGlmEst.m:347-511 with the Bernoulli-Gaussian denoiser
// (main/SparseScaEstim.m:92-165 around main/CAwgnEstimIn.m:94-102,181-184) and the AWGN likelihood
// (main/CAwgnEstimOut.m:97-108), batched over problems.
//
// The reference real-stacks the complex system (vamp.m:3-4) and takes a full SVD of the 2M x 2N
// matrix (:32).  Every singular value of the real-stacked matrix appears twice and its left vectors
// are the realification of the complex ones, so the LMMSE stage (VampGlmEst.m:397-403) is run here in
// complex arithmetic; only the denoiser and likelihood act per REAL coordinate.  For the dictionary the
// drivers pass, Phi = kron((B*B').', A) (plot_errorVSsnr.m:79), nothing of size Phi is ever formed:
//     Phi vec(X) = vec(Af X Gb),  U = conj(Ub) (x) Ua,  d = sa_i^2 lb_j^2
// with Af Af^H = Ua diag(sa^2) Ua^H and Gb = Ub diag(lb) Ub^H (Jacobi, n <= 128).
// Quirks kept: r1init = eps*1i makes SparseScaEstim use its COMPLEX log-likelihood branch on the
// real-stacked coordinates (:100-103) (the O(eps) imaginary parts themselves are dropped); r2 / p2 use
// the unclipped gam2x / gam2z (:367,:381); alf carries "- eps" (:398); no stopping rule (:505-511).
// NOTE: with sigma fixed at 1 the reference iteration does not converge and amplifies
// rounding differences ~1e9 over 100 iterations; fp32 results agree with float64 per iteration for the
// first tens of iterations and statistically (NMSE) thereafter.
#include "vamp_kernels.h"
#include <algorithm>

namespace jstsp {

static inline dim3 gsz(long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096))); }

static int vamp_run(jstsp_ctx *ctx, int Na, int Gr, int G2, int batch, const float2 *Y, const float2 *Af,
                    long long sA, const float2 *Gb, long long sG, double sigma, double Lnz, int nit, double damp,
                    float2 *Xout)
{
    Arena &a = ctx->arena;
    const int Nc = Gr * G2, Mc = Na * G2;
    const bool tall = Na > Gr;                     // M > N: VampGlmEst.m:407-411 with V, d from eig(A'A) (:196-218)
    const int Da = tall ? Gr : Na, Dc = tall ? Nc : Mc;
    const int nA = sA ? batch : 1, nG = sG ? batch : 1;
    const size_t bN = (size_t)batch * Nc, bM = (size_t)batch * Mc;
    float2 *r1 = a.get<float2>(bN), *x1 = a.get<float2>(bN), *r2 = a.get<float2>(bN), *x2 = a.get<float2>(bN),
           *u3 = a.get<float2>(bN);
    float2 *p1 = a.get<float2>(bM), *p2 = a.get<float2>(bM), *z2 = a.get<float2>(bM), *z2o = a.get<float2>(bM),
           *Ar2 = a.get<float2>(bM), *E = a.get<float2>(bM), *T1 = a.get<float2>(bM), *T2 = a.get<float2>(bM),
           *tq = a.get<float2>(bM), *tdq = a.get<float2>(bM);
    float *q = a.get<float>(bM), *dq = a.get<float>(bM);
    float2 *AAh = a.get<float2>((size_t)nA * Na * Na), *Ua = a.get<float2>((size_t)nA * Na * Na);
    float2 *Ub = a.get<float2>((size_t)nG * G2 * G2);
    float *lamA = a.get<float>((size_t)nA * Na), *lamB = a.get<float>((size_t)nG * G2);
    const int nea = (Da + 1) & ~1, neb = (G2 + 1) & ~1;     // order of the decomposition (Da), as vamp_bytes() budgets it
    // (orders above 128 go to the block Jacobi of eig_large.hip, which brings its own stream-ordered temporaries)
    float2 *Vga = a.get<float2>(Da <= 128 ? (size_t)nA * nea * nea : 16), *Vgb = a.get<float2>(G2 <= 128 ? (size_t)nG * neb * neb : 16);
    VampScal *sc = a.get<VampScal>(batch);
    JSTSP_REQUIRE(r1 && x1 && r2 && x2 && u3 && p1 && p2 && z2 && z2o && Ar2 && E && T1 && T2 && tq && tdq && q && dq &&
                      AAh && Ua && Ub && lamA && lamB && Vga && Vgb && sc,
                  JSTSP_E_NOMEM, "vamp: workspace exhausted");
    hipStream_t st = ctx->stream;
    // ---- decompositions (the `svd(B)` of vamp.m:32 in factored complex form)
    const Mat Am{Af, sA, Na}, Gm{Gb, sG, G2};
    // (the Gram in float64, rounded once: its eigenvalues are the SQUARES of the singular values the reference's svd returns,
    //  so every digit lost in forming it is lost twice)
    if (tall) JSTSP_TRY(gram_f64(ctx, 'L', Af, sA, Na, Gr, nA, AAh, (long long)Gr * Gr));              // A'A = Va diag(la) Va'
    else JSTSP_TRY(gram_f64(ctx, 'R', Af, sA, Na, Gr, nA, AAh, (long long)Na * Na));                   // A A' = Ua diag(sa^2) Ua'
    JSTSP_TRY(launch_eig(ctx, EIG_VECS, Da, nA, AAh, (long long)Da * Da, 1, 0, nullptr, nullptr, Ua, lamA, Vga));
    JSTSP_TRY(launch_eig(ctx, EIG_VECS, G2, nG, Gb, sG, 1, 0, nullptr, nullptr, Ub, lamB, Vgb));
    const Mat Uam{Ua, sA ? (long long)Da * Da : 0, Da}, Ubm{Ub, sG ? (long long)G2 * G2 : 0, G2};
    JSTSP_HIP(hipMemsetAsync(r1, 0, bN * sizeof(float2), st));            // r1init = eps*1i ~ 0 (vamp.m:45)
    JSTSP_HIP(hipMemsetAsync(p1, 0, bM * sizeof(float2), st));            // VampGlmEst.m:331
    JSTSP_HIP(hipMemsetAsync(x1, 0, bN * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(z2o, 0, bM * sizeof(float2), st));
    hipLaunchKernelGGL((vamp_init_kernel<0>), dim3((batch + 255) / 256), dim3(256), 0, st, batch, sc);
    const long long sN = Nc, sM = Mc;
    for (int it = 0; it < nit; ++it) {
        hipLaunchKernelGGL((vamp_first_half_kernel<float2, float>), dim3(batch), dim3(256), 0, st, Nc, Mc, Dc, Da, G2, it, damp, sigma, Lnz, Y,
                           r1, p1, x1, r2, p2, lamA, sA ? (long long)Da : 0, lamB, sG ? (long long)G2 : 0, q, dq, sc);
        if (tall) {
            // Vr2Ap2 = V'(r2 gam2x/gam2z + A'p2);  x2 = V(Vr2Ap2 .* q);  z2 = A x2                 (:408-410)
            // with Phi^H vec(Z) = vec(Af^H Z Gb),  V^H vec(X) = vec(Va^H X Ub),  V vec(T) = vec(Va T Ub^H)
            JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, Na, batch, Am, Mat{p2, sM, Na}, u3, sN, Gr));
            JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, G2, batch, Mat{u3, sN, Gr}, Gm, x2, sN, Gr));
            hipLaunchKernelGGL((vamp_add_ratio_kernel<float2, float>), dim3((unsigned)std::min(64, (Nc + 255) / 256), batch), dim3(256), 0, st, Nc,
                               x2, r2, sc);
            JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, Gr, batch, Uam, Mat{x2, sN, Gr}, u3, sN, Gr));
            JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, G2, batch, Mat{u3, sN, Gr}, Ubm, x2, sN, Gr));
            hipLaunchKernelGGL((vamp_scale_kernel<float2, float>), gsz((long long)bN), dim3(256), 0, st, (long long)bN, x2, q, dq, u3, tdq);
            JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, Gr, batch, Uam, Mat{u3, sN, Gr}, T2, sN, Gr));
            JSTSP_TRY(gemm(ctx, 'N', 'C', Gr, G2, G2, batch, Mat{T2, sN, Gr}, Ubm, x2, sN, Gr));
            JSTSP_TRY(gemm(ctx, 'N', 'N', Na, G2, Gr, batch, Am, Mat{x2, sN, Gr}, T1, sM, Na));
            JSTSP_TRY(gemm(ctx, 'N', 'N', Na, G2, G2, batch, Mat{T1, sM, Na}, Gm, z2, sM, Na));
            hipLaunchKernelGGL((vamp_second_half_kernel<float2, float>), dim3(batch), dim3(256), 0, st, Nc, Mc, it, damp, x2, r2, z2, z2o, p2,
                               r1, p1, sc);
            continue;
        }
        // Ar2 = Af R2 Gb                                                              (:400)
        JSTSP_TRY(gemm(ctx, 'N', 'N', Na, G2, Gr, batch, Am, Mat{r2, sN, Gr}, T1, sM, Na));
        JSTSP_TRY(gemm(ctx, 'N', 'N', Na, G2, G2, batch, Mat{T1, sM, Na}, Gm, Ar2, sM, Na));
        // t = (U^H (p2 - Ar2)) .* q,  U^H vec(Z) = vec(Ua^H Z Ub)                      (:401)
        hipLaunchKernelGGL((vamp_sub_kernel<float2, float>), gsz((long long)bM), dim3(256), 0, st, (long long)bM, p2, Ar2, E);
        JSTSP_TRY(gemm(ctx, 'C', 'N', Na, G2, Na, batch, Uam, Mat{E, sM, Na}, T1, sM, Na));
        JSTSP_TRY(gemm(ctx, 'N', 'N', Na, G2, G2, batch, Mat{T1, sM, Na}, Ubm, T2, sM, Na));
        hipLaunchKernelGGL((vamp_scale_kernel<float2, float>), gsz((long long)bM), dim3(256), 0, st, (long long)bM, T2, q, dq, tq, tdq);
        // x2 = r2 + Phi^H U t,  U vec(T) = vec(Ua T Ub^H),  Phi^H vec(Z) = vec(Af^H Z Gb)   (:402)
        JSTSP_TRY(gemm(ctx, 'N', 'N', Na, G2, Na, batch, Uam, Mat{tq, sM, Na}, T1, sM, Na));
        JSTSP_TRY(gemm(ctx, 'N', 'C', Na, G2, G2, batch, Mat{T1, sM, Na}, Ubm, T2, sM, Na));
        JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, Na, batch, Am, Mat{T2, sM, Na}, u3, sN, Gr));
        JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, G2, batch, Mat{u3, sN, Gr}, Gm, x2, sN, Gr, 1.f, r2, sN, Gr, 1.f));
        // z2 = Ar2 + U (d .* t)                                                       (:403)
        JSTSP_TRY(gemm(ctx, 'N', 'N', Na, G2, Na, batch, Uam, Mat{tdq, sM, Na}, T1, sM, Na));
        JSTSP_TRY(gemm(ctx, 'N', 'C', Na, G2, G2, batch, Mat{T1, sM, Na}, Ubm, z2, sM, Na, 1.f, Ar2, sM, Na, 1.f));
        hipLaunchKernelGGL((vamp_second_half_kernel<float2, float>), dim3(batch), dim3(256), 0, st, Nc, Mc, it, damp, x2, r2, z2, z2o, p2,
                           r1, p1, sc);
    }
    JSTSP_HIP(hipGetLastError());
    JSTSP_HIP(hipMemcpyAsync(Xout, x1, bN * sizeof(float2), hipMemcpyDeviceToDevice, st));   // x = x1 (vamp.m:54)
    return 0;
}

static size_t vamp_bytes(int Na, int Gr, int G2, int batch, int nA, int nG)
{
    const size_t bN = (size_t)batch * Gr * G2, bM = (size_t)batch * Na * G2;
    const int Da = std::min(Na, Gr), nea = (Da + 1) & ~1, neb = (G2 + 1) & ~1;
    return 6 * rnd256(bN * sizeof(float2)) + 10 * rnd256(bM * sizeof(float2)) + 2 * rnd256(bM * sizeof(float)) +
           2 * rnd256((size_t)nA * Na * Na * sizeof(float2)) + rnd256((size_t)nG * G2 * G2 * sizeof(float2)) +
           rnd256((size_t)nA * Na * sizeof(float)) + rnd256((size_t)nG * G2 * sizeof(float)) +
           rnd256((Da <= 128 ? (size_t)nA * nea * nea : 16) * sizeof(float2)) + rnd256((G2 <= 128 ? (size_t)nG * neb * neb : 16) * sizeof(float2)) +
           rnd256(batch * sizeof(VampScal)) + 4096;
}

}  // namespace jstsp

using namespace jstsp;

extern "C" {

int jstsp_vamp_kron_c32(jstsp_ctx *ctx, int Na, int Gr, int G2, int batch, const jstsp_c32 *Y_, const jstsp_c32 *Af_,
                        long long strideA, const jstsp_c32 *Gb_, long long strideG, double sigma, double Lnz, int nit,
                        jstsp_c32 *X_out, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(Y_ && Af_ && Gb_ && X_out, JSTSP_E_NULL, "vamp_kron: NULL array argument");
    JSTSP_REQUIRE(Na > 0 && Gr > 0 && G2 > 0 && batch > 0 && nit >= 1, JSTSP_E_SHAPE, "vamp_kron: bad shape");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    // the dense call of the drivers, vamp(y, kron((B*B').', A), 1, numOfnz) (plot_errorVSsnr.m:79-80,100), arrives here with
    // G2 = 1 and a 512 x 512 "A factor": orders above 128 take the block Jacobi of eig_large.hip on either side
    JSTSP_REQUIRE(std::min(Na, Gr) <= 2048 && G2 <= 8192, JSTSP_E_UNSUPPORTED,
                  "vamp_kron: min(Na, Gr) = %d, G2 = %d: the factor eigenproblems are limited to orders 2048 and 8192",
                  std::min(Na, Gr), G2);
    JSTSP_REQUIRE(sigma > 0 && Lnz > 0 && Lnz < 2.0 * Gr * G2, JSTSP_E_ARG, "vamp: need sigma > 0 and 0 < L < nx");
    JSTSP_ENTER(ctx);
    const int nA = strideA ? batch : 1, nG = strideG ? batch : 1;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)Na * Gr : (size_t)Na * Gr;
    const size_t szG = strideG ? (size_t)strideG * (batch - 1) + (size_t)G2 * G2 : (size_t)G2 * G2;
    const size_t bN = (size_t)batch * Gr * G2, bM = (size_t)batch * Na * G2;
    size_t need = vamp_bytes(Na, Gr, G2, batch, nA, nG) + rnd256(bN * sizeof(float2));
    if (memspace == JSTSP_HOST) need += rnd256(bM * sizeof(float2)) + rnd256(szA * sizeof(float2)) + rnd256(szG * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *Y, *Af, *Gb;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Y_), bM, memspace, &Y));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Af_), szA, memspace, &Af));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Gb_), szG, memspace, &Gb));
    float2 *X = ctx->arena.get<float2>(bN);
    JSTSP_REQUIRE(X, JSTSP_E_NOMEM, "vamp: workspace exhausted");
    JSTSP_TRY(vamp_run(ctx, Na, Gr, G2, batch, Y, Af, strideA, Gb, strideG, sigma, Lnz, nit, 0.85, X));   // damp: vamp.m:11
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(X_out), X, bN, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// x = vamp(y, A, sigma, L) with a dense dictionary A (M x N, min(M, N) <= 2048): the Kronecker form
// with G2 = 1, Gb = 1.
int jstsp_vamp_c32(jstsp_ctx *ctx, int M, int N, int batch, const jstsp_c32 *y, const jstsp_c32 *A, long long strideA,
                   double sigma, double Lnz, int nit, jstsp_c32 *x_out, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_ENTER(ctx);
    static const jstsp_c32 one_h = {1.f, 0.f};
    if (memspace == JSTSP_HOST)
        return jstsp_vamp_kron_c32(ctx, M, N, 1, batch, y, A, strideA, &one_h, 0, sigma, Lnz, nit, x_out, memspace);
    // device arrays: a device copy of the 1 x 1 identity factor, owned by the context (its device, its lifetime)
    if (!ctx->unit) {
        JSTSP_HIP(hipMalloc((void **)&ctx->unit, sizeof(float2)));
        JSTSP_HIP(hipMemcpy(ctx->unit, &one_h, sizeof(float2), hipMemcpyHostToDevice));
    }
    return jstsp_vamp_kron_c32(ctx, M, N, 1, batch, y, A, strideA, reinterpret_cast<const jstsp_c32 *>(ctx->unit), 0, sigma,
                               Lnz, nit, x_out, memspace);
}

}  // extern "C"
