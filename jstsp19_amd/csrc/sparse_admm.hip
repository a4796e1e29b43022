// jstsp_sparse_admm_c32 — benchmark_algorithms/sparse_admm.m:1-36 in structured form.
//
// The reference builds A = kron(conj(Dt), Dr) (:15), B = A'A - rho I (:16) and solves the dense
// system B \ rhs every iteration (:26).  With the factor Grams
//     Gr_ = Dr^H Dr = Ur diag(lr) Ur^H,   Gt_ = Dt^T conj(Dt) = Ut diag(lt) Ut^H
// A'A = Gt_ (x) Gr_, so  B vec(R) = vec(Gr_ R Gt_^T - rho R)  and the solve is diagonal in the
// factors' eigenbases:  R = Ur [ (Ur^H RHS conj(Ut)) ./ (lr lt^T - rho) ] Ut^T.
// A'*vec(OH) = vec(Dr^H OH Dt).  Constants rho = 0.01, tau_s = 1e-4 are the reference's (:12-13).
// The first iteration adds R (Gr x Gt) to Z (Mr x Mt) (:21), so Gr == Mr and Gt == Mt.
#include "solver_common.h"
#include <algorithm>

namespace jstsp {

// v = R + Z/rho (:21); S = soft(v, tau_s/rho) (:22); RHS = Z - rho S + A'vec(OH) (:26)
__global__ __launch_bounds__(256) void sadmm_soft_rhs_kernel(long long n, const float2 *R, const float2 *Z,
                                                             const float2 *AhOH, float2 *S, float2 *RHS,
                                                             float rho, float thr)
{
    const float ir = 1.f / rho;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float2 r = R[i], z = Z[i], a = AhOH[i];
        const float2 s = make_float2(sadmm_soft1(r.x, z.x, ir, thr), sadmm_soft1(r.y, z.y, ir, thr));
        S[i] = s;
        RHS[i] = make_float2(sadmm_rhs1(z.x, s.x, a.x, rho), sadmm_rhs1(z.y, s.y, a.y, rho));
    }
}

// T ./ (lr_i lt_j - rho), per problem nm = Mr*Mt
__global__ __launch_bounds__(256) void sadmm_scale_kernel(long long total, int Mr, int Mt, float2 *T,
                                                          const float *lr, const float *lt, float rho)
{
    const long long stride = (long long)gridDim.x * 256;
    const long long nm = (long long)Mr * Mt;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long long e = i % nm;
        const float den = sadmm_den(lr[e % Mr], lt[e / Mr], rho);
        const float2 v = T[i];
        T[i] = make_float2(v.x / den, v.y / den);
    }
}

// Z = Z + rho (R - S) (:30)
__global__ __launch_bounds__(256) void sadmm_dual_kernel(long long n, float2 *Z, const float2 *R,
                                                         const float2 *S, float rho)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float2 z = Z[i];
        const float2 r = R[i], s = S[i];
        z.x = sadmm_dual1(z.x, r.x, s.x, rho);
        z.y = sadmm_dual1(z.y, r.y, s.y, rho);
        Z[i] = z;
    }
}

__global__ __launch_bounds__(256) void sadmm_diff_kernel(long long n, const float2 *a, const float2 *b, float2 *o)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        o[i] = make_float2(a[i].x - b[i].x, a[i].y - b[i].y);
}

__global__ void sadmm_ratio_kernel(int batch, const float *num, const float *den, double *ce, int Imax, int it)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < batch) ce[(long long)t * Imax + it] = (double)num[t] / (double)den[t];
}

static inline dim3 g1(long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 8192))); }

}  // namespace jstsp

using namespace jstsp;

extern "C" int jstsp_sparse_admm_c32(jstsp_ctx *ctx, int Mr, int Mt, int Gr, int Gt, int batch,
                                     const jstsp_c32 *Htrue_, const jstsp_c32 *OH_, const jstsp_c32 *Dr_,
                                     const jstsp_c32 *Dt_, int Imax, jstsp_c32 *S_out, double *ce_out,
                                     int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(OH_ && Dr_ && Dt_ && S_out, JSTSP_E_NULL, "sparse_admm: NULL array argument");
    JSTSP_REQUIRE(!ce_out || Htrue_, JSTSP_E_NULL, "sparse_admm: convergence_error needs Htrue");
    JSTSP_REQUIRE(Mr > 0 && Mt > 0 && batch > 0 && Imax >= 0, JSTSP_E_SHAPE, "sparse_admm: bad shape");
    JSTSP_REQUIRE(Gr == Mr && Gt == Mt, JSTSP_E_SHAPE,
                  "sparse_admm: the reference adds R (Gr x Gt) to Z (Mr x Mt) (sparse_admm.m:21) and forms "
                  "A'A - rho*eye(Mr*Mt) (:16): Gr must equal Mr and Gt must equal Mt (got %dx%d vs %dx%d)",
                  Gr, Gt, Mr, Mt);
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_REQUIRE(std::max(Mr, Mt) <= 2048, JSTSP_E_UNSUPPORTED,
                  "sparse_admm: max(Mr, Mt) = %d > 2048 (order of the factor-Gram eigenproblems)", std::max(Mr, Mt));
    JSTSP_ENTER(ctx);
    const bool want_ce = ce_out != nullptr;
    const size_t nm = (size_t)Mr * Mt;
    const float rho = 0.01f, tau_s = 0.0001f;                                 // :12-13
    const int ner = (Mr + 1) & ~1, net = (Mt + 1) & ~1;

    size_t need = (want_ce ? 8 : 7) * rnd256(batch * nm * sizeof(float2)) + 2 * rnd256((size_t)Mr * Mr * sizeof(float2)) +
                  2 * rnd256((size_t)Mt * Mt * sizeof(float2)) + rnd256(Mr * sizeof(float)) +
                  rnd256(Mt * sizeof(float)) + rnd256((size_t)ner * ner * sizeof(float2)) +
                  rnd256((size_t)net * net * sizeof(float2)) + 2 * rnd256(batch * sizeof(float)) +
                  rnd256((size_t)batch * std::max(Imax, 1) * sizeof(double));
    if (want_ce) need += GramWS::bytes(Mr, Mt, batch, false);
    if (memspace == JSTSP_HOST)
        need += 2 * rnd256(batch * nm * sizeof(float2)) + rnd256((size_t)Mr * Gr * sizeof(float2)) +
                rnd256((size_t)Mt * Gt * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    Arena &a = ctx->arena;
    const float2 *OH, *Dr, *Dt, *Htrue = nullptr;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(OH_), batch * nm, memspace, &OH));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Dr_), (size_t)Mr * Gr, memspace, &Dr));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Dt_), (size_t)Mt * Gt, memspace, &Dt));
    if (want_ce) JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Htrue_), batch * nm, memspace, &Htrue));

    float2 *R = a.get<float2>(batch * nm), *Z = a.get<float2>(batch * nm), *S = a.get<float2>(batch * nm),
           *RHS = a.get<float2>(batch * nm), *P = a.get<float2>(batch * nm), *AhOH = a.get<float2>(batch * nm),
           *Dd = a.get<float2>(batch * nm);
    float2 *Gr_ = a.get<float2>((size_t)Mr * Mr), *Ur = a.get<float2>((size_t)Mr * Mr);
    float2 *Gt_ = a.get<float2>((size_t)Mt * Mt), *Ut = a.get<float2>((size_t)Mt * Mt);
    float *lr = a.get<float>(Mr), *lt = a.get<float>(Mt);
    float2 *Vgr = a.get<float2>((size_t)ner * ner), *Vgt = a.get<float2>((size_t)net * net);
    float *num = a.get<float>(batch), *den = a.get<float>(batch);
    double *ce = a.get<double>((size_t)batch * std::max(Imax, 1));
    JSTSP_REQUIRE(R && Z && S && RHS && P && AhOH && Dd && Gr_ && Ur && Gt_ && Ut && lr && lt && Vgr && Vgt &&
                      num && den && ce,
                  JSTSP_E_NOMEM, "sparse_admm: workspace exhausted");
    float2 *P2 = want_ce ? a.get<float2>(batch * nm) : nullptr;        // the error chain's own temporary (it runs beside the solve)
    JSTSP_REQUIRE(!want_ce || P2, JSTSP_E_NOMEM, "sparse_admm: workspace exhausted");
    GramWS wn;
    if (want_ce) JSTSP_TRY(wn.alloc(a, Mr, Mt, batch, false));
    hipStream_t st = ctx->stream;
    const long long snm = (long long)nm, tot = (long long)batch * nm;

    // ---- setup: factor Grams and their eigen-decompositions (shared by the batch) -----------------
    const Mat Drm{Dr, 0, Mr}, Dtm{Dt, 0, Mt};
    JSTSP_TRY(gemm(ctx, 'C', 'N', Mr, Mr, Mr, 1, Drm, Drm, Gr_, 0, Mr));          // Dr^H Dr
    JSTSP_TRY(gemm(ctx, 'T', 'J', Mt, Mt, Mt, 1, Dtm, Dtm, Gt_, 0, Mt));          // Dt^T conj(Dt)
    JSTSP_TRY(launch_eig(ctx, EIG_VECS, Mr, 1, Gr_, 0, 1, 0, nullptr, nullptr, Ur, lr, Vgr));
    JSTSP_TRY(launch_eig(ctx, EIG_VECS, Mt, 1, Gt_, 0, 1, 0, nullptr, nullptr, Ut, lt, Vgt));
    const Mat Urm{Ur, 0, Mr}, Utm{Ut, 0, Mt};
    // A'*vec(OH) = vec(Dr^H OH Dt)
    JSTSP_TRY(gemm(ctx, 'C', 'N', Mr, Mt, Mr, batch, Drm, Mat{OH, snm, Mr}, P, snm, Mr));
    JSTSP_TRY(gemm(ctx, 'N', 'N', Mr, Mt, Mt, batch, Mat{P, snm, Mr}, Dtm, AhOH, snm, Mr));
    JSTSP_HIP(hipMemsetAsync(R, 0, batch * nm * sizeof(float2), st));              // :8-9
    JSTSP_HIP(hipMemsetAsync(Z, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(S, 0, batch * nm * sizeof(float2), st));
    if (want_ce) JSTSP_TRY(sigma_max_sq(ctx, wn, Htrue, den));
    if (want_ce) JSTSP_TRY(lanczos_warm_reset(ctx, wn));       // the error curve's lambda_max, warm-started from iteration to iteration

    // JSTSP_SADMM_FUSE=0: every element-wise step as its own kernel.  Default: they ride on the products' epilogues (EPI_SADMM,
    // common.h) - the diagonal solve on the second transform, the dual update AND the next iteration's soft threshold / right-hand
    // side on the fourth (R itself is then never stored; S alternates between two buffers because convergence_error still reads
    // this iteration's S), the difference to Htrue on the error product.  Same expressions (sadmm_*1), same bits.
    const char *fuse_env = xp_getenv("JSTSP_SADMM_FUSE");          // (read at every call)
    const bool fuse = !fuse_env || atoi(fuse_env) != 0;
    float2 *Sb[2] = {S, fuse ? R : S};
    const float2 *Sfin = S;
    // convergence_error of iteration it needs S(it) only, the solve goes on from S(it) without it: with the fused epilogues (two
    // S buffers) the error chain - two products, the Gram, lambda_max - runs on a side stream beside the next solve
    // (JSTSP_SADMM_OVERLAP=0: in line).  ev_s[it & 1]: S(it) is complete; ev_r[it & 1]: the error chain has read S(it), whose
    // buffer the epilogue of iteration it + 1 overwrites.
    const char *ov_env = xp_getenv("JSTSP_SADMM_OVERLAP");
    const bool overlap = fuse && want_ce && (!ov_env || atoi(ov_env) != 0);
    if (overlap) JSTSP_TRY(ensure_side_streams(ctx));
    hipStream_t sc = overlap ? ctx->side[0] : st;
    hipEvent_t ev_s[2] = {ctx->ev[0], ctx->ev[1]}, ev_r[2] = {ctx->ev[2], ctx->ev[3]};
    for (int it = 0; it < Imax; ++it) {                                            // :18
        float2 *Sc = Sb[it & 1], *Sn = Sb[(it + 1) & 1];
        Sfin = Sc;
        if (!fuse || it == 0)
            hipLaunchKernelGGL(sadmm_soft_rhs_kernel, g1(tot), dim3(256), 0, st, tot, R, Z, AhOH, Sc, RHS, rho,
                               tau_s / rho);                                       // :21-23
        if (overlap) {
            if (it == 0) JSTSP_HIP(hipEventRecord(ev_s[0], st));
            JSTSP_HIP(hipStreamWaitEvent(sc, ev_s[it & 1], 0));
            StreamScope scope(ctx, sc);
            JSTSP_TRY(gemm(ctx, 'N', 'N', Mr, Mt, Mr, batch, Drm, Mat{Sc, snm, Mr}, P2, snm, Mr));
            JSTSP_HIP(hipEventRecord(ev_r[it & 1], sc));
            JSTSP_TRY(gemm(ctx, 'N', 'C', Mr, Mt, Mt, batch, Mat{P2, snm, Mr}, Dtm, Dd, snm, Mr, 1.f, Htrue, snm, Mr, -1.f));
            JSTSP_TRY(sigma_max_sq(ctx, wn, Dd, num, true));
            hipLaunchKernelGGL(sadmm_ratio_kernel, dim3((batch + 255) / 256), dim3(256), 0, sc, batch, num, den, ce, Imax, it);
        }
        // (the last iteration's R and Z feed nothing that is returned: S and convergence_error are complete before them)
        if (!fuse || it + 1 < Imax) {
            // :26  R = Ur [ (Ur^H RHS conj(Ut)) ./ (lr lt^T - rho) ] Ut^T
            JSTSP_TRY(gemm(ctx, 'C', 'N', Mr, Mt, Mr, batch, Urm, Mat{RHS, snm, Mr}, P, snm, Mr));
            if (fuse) {
                GemmDesc d2 = make_gemm('N', 'J', Mr, Mt, Mt, batch, Mat{P, snm, Mr}, Utm, RHS, snm, Mr);
                d2.epi = EPI_SADMM; d2.sa_mode = 1; d2.sa_lr = lr; d2.sa_lt = lt; d2.sa_rho = rho;
                JSTSP_TRY(launch_cgemm(ctx, d2, GEMM_MISC));
            } else {
                JSTSP_TRY(gemm(ctx, 'N', 'J', Mr, Mt, Mt, batch, Mat{P, snm, Mr}, Utm, RHS, snm, Mr));
                hipLaunchKernelGGL(sadmm_scale_kernel, g1(tot), dim3(256), 0, st, tot, Mr, Mt, RHS, lr, lt, rho);
            }
            JSTSP_TRY(gemm(ctx, 'N', 'N', Mr, Mt, Mr, batch, Urm, Mat{RHS, snm, Mr}, P, snm, Mr));
            if (fuse) {
                GemmDesc d4 = make_gemm('N', 'T', Mr, Mt, Mt, batch, Mat{P, snm, Mr}, Utm, RHS, snm, Mr);   // (C unused: R is not stored)
                d4.epi = EPI_SADMM; d4.sa_mode = 2; d4.sa_rho = rho; d4.sa_thr = tau_s / rho;
                d4.e_rw0 = Z; d4.e_r0 = Sc; d4.e_w1 = Sn; d4.e_r2 = AhOH; d4.e_w2 = RHS;
                if (overlap && it > 0) JSTSP_HIP(hipStreamWaitEvent(st, ev_r[(it + 1) & 1], 0));    // S(it - 1) has been read
                JSTSP_TRY(launch_cgemm(ctx, d4, GEMM_MISC));
                if (overlap) JSTSP_HIP(hipEventRecord(ev_s[(it + 1) & 1], st));
            } else {
                JSTSP_TRY(gemm(ctx, 'N', 'T', Mr, Mt, Mt, batch, Mat{P, snm, Mr}, Utm, R, snm, Mr));
                hipLaunchKernelGGL(sadmm_dual_kernel, g1(tot), dim3(256), 0, st, tot, Z, R, Sc, rho);   // :30
            }
        }
        if (want_ce && !overlap) {                                                 // :32
            JSTSP_TRY(gemm(ctx, 'N', 'N', Mr, Mt, Mr, batch, Drm, Mat{Sc, snm, Mr}, P, snm, Mr));
            if (fuse)
                JSTSP_TRY(gemm(ctx, 'N', 'C', Mr, Mt, Mt, batch, Mat{P, snm, Mr}, Dtm, Dd, snm, Mr, 1.f, Htrue, snm, Mr, -1.f));
            else {
                JSTSP_TRY(gemm(ctx, 'N', 'C', Mr, Mt, Mt, batch, Mat{P, snm, Mr}, Dtm, Dd, snm, Mr));
                hipLaunchKernelGGL(sadmm_diff_kernel, g1(tot), dim3(256), 0, st, tot, Dd, Htrue, Dd);
            }
            JSTSP_TRY(sigma_max_sq(ctx, wn, Dd, num, true));
            hipLaunchKernelGGL(sadmm_ratio_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, batch, num, den, ce,
                               Imax, it);
        }
    }
    if (overlap && Imax > 0) {          // the error chain joins the main stream
        JSTSP_HIP(hipEventRecord(ev_r[0], sc));
        JSTSP_HIP(hipStreamWaitEvent(st, ev_r[0], 0));
    }
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(S_out), Sfin, batch * nm, memspace));
    if (want_ce && Imax > 0) JSTSP_TRY(stage_out(ctx, ce_out, ce, (size_t)batch * Imax, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(st));
    return 0;
}
