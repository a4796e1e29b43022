// The _c64 entry points of include/jstsp.h: the reference's own element type (MATLAB double complex, interleaved) at the
// boundary.  The arithmetic is the _c32 path's (fp32 storage, split-f16 / fp32 MFMA contractions);
// what these add is the narrowing of the inputs and the widening of the outputs ON THE DEVICE, so that a double-precision
// host (the MEX gateway, a numpy caller) hands over its arrays as they are.  Temporaries are stream-ordered allocations
// outside the context's workspace (the solvers reset that); a JSTSP_DEVICE call stays asynchronous.
#include "common.h"
#include "solver_common.h"

#include <vector>

namespace jstsp {
namespace {

__global__ void narrow_kernel(const double *__restrict__ src, float *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}
__global__ void widen_kernel(const float *__restrict__ src, double *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (double)src[i];
}
inline unsigned conv_grid(size_t n) { return (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 8192)); }

// One call's conversions.  in*(): device fp32 copy of a caller array; out*(): device buffer whose content finish()
// delivers to the caller.  Any failure is latched in rc (checked once before the solver runs).
struct Conv {
    jstsp_ctx *ctx;
    int memspace;
    int rc = 0;
    std::vector<void *> held;
    struct Out { void *dst; void *dev; size_t n; bool widen; };
    std::vector<Out> outs;

    Conv(jstsp_ctx *c, int ms) : ctx(c), memspace(ms) {}
    ~Conv()
    {
        for (void *p : held) (void)hipFreeAsync(p, ctx->stream);
    }
    void *dmalloc(size_t bytes)
    {
        if (rc) return nullptr;
        void *p = nullptr;
        hipError_t e = hipMallocAsync(&p, std::max<size_t>(bytes, 16), ctx->stream);
        if (e != hipSuccess) {
            set_error("hipMallocAsync(%zu) failed: %s", bytes, hipGetErrorString(e));
            rc = (int)e;
            return nullptr;
        }
        held.push_back(p);
        return p;
    }
    bool ok(hipError_t e, const char *what)
    {
        if (e == hipSuccess) return true;
        if (!rc) { set_error("%s failed: %s", what, hipGetErrorString(e)); rc = (int)e; }
        return false;
    }
    // n doubles -> n floats on the device
    const float *in_real(const double *src, size_t n)
    {
        if (!src || rc) return nullptr;
        const double *d64 = src;
        if (memspace == JSTSP_HOST) {
            double *t = (double *)dmalloc(n * sizeof(double));
            if (!t || !ok(hipMemcpyAsync(t, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream), "upload")) return nullptr;
            d64 = t;
        }
        float *f = (float *)dmalloc(n * sizeof(float));
        if (!f) return nullptr;
        narrow_kernel<<<conv_grid(n), 256, 0, ctx->stream>>>(d64, f, n);
        return f;
    }
    const jstsp_c32 *in(const jstsp_c64 *src, size_t n)
    {
        return reinterpret_cast<const jstsp_c32 *>(in_real(reinterpret_cast<const double *>(src), 2 * n));
    }
    // a dictionary factor: elems per problem, `stride` elements between problems (0 = shared)
    const jstsp_c32 *in_dict(const jstsp_c64 *src, size_t elems, long long stride, int batch)
    {
        return in(src, stride ? (size_t)(batch - 1) * (size_t)stride + elems : elems);
    }
    // the per-trial dictionaries B of a JSTSP_HOST proposed_algorithm call: tested for the block-Toeplitz structure on the host
    // and uploaded as fp32 first blocks + leading columns (hostpack.hip) - the link carries 1 / (2 L) of the doubles.  *gt = the
    // block height (the caller hands it to the solver as a hint: no device probe), 0 = the plain route was taken.
    const jstsp_c32 *in_dict_toeplitz(const jstsp_c64 *src, int G2, int M, long long stride, int batch, int *gt)
    {
        *gt = 0;
        const size_t per = (size_t)G2 * M;
        const bool tryit = memspace == JSTSP_HOST && tune().host_compact != 0 && tune().toeplitz != 0 && batch > 1 && G2 >= 32 &&
                           stride == (long long)per && (per * batch * sizeof(jstsp_c64) >= ((size_t)128 << 20) || tune().host_compact >= 2) &&
                           (long long)per < (1ll << 31);
        if (tryit && !rc) {
            const size_t ce = host_toeplitz_compact_elems(G2, M, batch);
            float2 *bd = (float2 *)dmalloc(per * batch * sizeof(float2)), *cd = (float2 *)dmalloc(ce * sizeof(float2));
            if (bd && cd) {
                const int r = host_toeplitz_stage(ctx, reinterpret_cast<const double2 *>(src), G2, M, batch, bd, cd, ce, gt);
                if (r) { rc = r; return nullptr; }
                if (*gt) return reinterpret_cast<const jstsp_c32 *>(bd);
            }
            if (rc) return nullptr;
        }
        return in_dict(src, per, stride, batch);
    }
    // arrays passed through unchanged (int32 indices, doubles the _c32 entry already takes)
    template <class T> const T *in_raw(const T *src, size_t n)
    {
        if (!src || rc || memspace == JSTSP_DEVICE) return src;
        T *t = (T *)dmalloc(n * sizeof(T));
        if (!t || !ok(hipMemcpyAsync(t, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream), "upload")) return nullptr;
        return t;
    }
    jstsp_c32 *out(jstsp_c64 *dst, size_t n)
    {
        if (!dst || rc) return nullptr;
        float *f = (float *)dmalloc(2 * n * sizeof(float));
        if (f) outs.push_back({dst, f, 2 * n, true});
        return reinterpret_cast<jstsp_c32 *>(f);
    }
    template <class T> T *out_raw(T *dst, size_t n)
    {
        if (!dst || rc || memspace == JSTSP_DEVICE) return dst;
        T *t = (T *)dmalloc(n * sizeof(T));
        if (t) outs.push_back({dst, t, n * sizeof(T), false});
        return t;
    }
    // fp32 device array -> the caller's doubles (host memspace), enqueued on the context's stream
    int deliver(const float *dev, double *dst, size_t n)
    {
        if (!dst) return 0;
        double *d64 = (double *)dmalloc(n * sizeof(double));
        if (!d64) return rc;
        widen_kernel<<<conv_grid(n), 256, 0, ctx->stream>>>(dev, d64, n);
        JSTSP_HIP(hipMemcpyAsync(dst, d64, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        return 0;
    }
    int finish()
    {
        for (const Out &o : outs) {
            if (o.widen) {
                double *d64 = (double *)o.dst;
                if (memspace == JSTSP_HOST) {
                    d64 = (double *)dmalloc(o.n * sizeof(double));
                    if (!d64) return rc;
                }
                widen_kernel<<<conv_grid(o.n), 256, 0, ctx->stream>>>((const float *)o.dev, d64, o.n);
                if (memspace == JSTSP_HOST)
                    JSTSP_HIP(hipMemcpyAsync(o.dst, d64, o.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            } else {
                JSTSP_HIP(hipMemcpyAsync(o.dst, o.dev, o.n, hipMemcpyDeviceToHost, ctx->stream));
            }
        }
        JSTSP_HIP(hipGetLastError());
        if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
        return rc;
    }
};

int enter(jstsp_ctx *ctx, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    return 0;
}

}  // namespace
}  // namespace jstsp

using namespace jstsp;

#define C64_BEGIN(shape_ok, what)                                                               \
    JSTSP_TRY(enter(ctx, memspace));                                                            \
    JSTSP_ENTER(ctx);                                                                           \
    JSTSP_REQUIRE(shape_ok, JSTSP_E_SHAPE, what ": bad shape");                                 \
    Conv cv(ctx, memspace)

extern "C" {

int jstsp_correlate_c64(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c64 *K,
                        const jstsp_c64 *A, long long strideA, const jstsp_c64 *B, long long strideB,
                        jstsp_c64 *out, int memspace)
{
    C64_BEGIN(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0 && strideA >= 0 && strideB >= 0, "correlate");
    JSTSP_REQUIRE(K && A && B && out, JSTSP_E_NULL, "correlate: NULL argument");
    const jstsp_c32 *k = cv.in(K, (size_t)N * M * batch), *a = cv.in_dict(A, (size_t)N * Gr, strideA, batch),
                    *b = cv.in_dict(B, (size_t)G2 * M, strideB, batch);
    jstsp_c32 *o = cv.out(out, (size_t)Gr * G2 * batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_correlate_c32(ctx, N, M, Gr, G2, batch, k, a, strideA, b, strideB, o, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_synthesize_c64(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c64 *S,
                         const jstsp_c64 *A, long long strideA, const jstsp_c64 *B, long long strideB,
                         jstsp_c64 *out, int memspace)
{
    C64_BEGIN(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0 && strideA >= 0 && strideB >= 0, "synthesize");
    JSTSP_REQUIRE(S && A && B && out, JSTSP_E_NULL, "synthesize: NULL argument");
    const jstsp_c32 *s = cv.in(S, (size_t)Gr * G2 * batch), *a = cv.in_dict(A, (size_t)N * Gr, strideA, batch),
                    *b = cv.in_dict(B, (size_t)G2 * M, strideB, batch);
    jstsp_c32 *o = cv.out(out, (size_t)N * M * batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_synthesize_c32(ctx, N, M, Gr, G2, batch, s, a, strideA, b, strideB, o, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_proposed_algorithm_c64(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c64 *subY,
                                 const double *Omega, const jstsp_c64 *A, long long strideA, const jstsp_c64 *B,
                                 long long strideB, int Imax, const double *tau_Y, const double *tau_S,
                                 const double *rho, int type, const int32_t *indx_S, jstsp_c64 *S_out,
                                 jstsp_c64 *Y_out, double *ce_out, int memspace)
{
    C64_BEGIN(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0 && Imax > 0 && strideA >= 0 && strideB >= 0,
              "proposed_algorithm");
    JSTSP_REQUIRE(subY && Omega && A && B && tau_Y && tau_S && rho && S_out, JSTSP_E_NULL,
                  "proposed_algorithm: NULL argument");
    // JSTSP_HOST with a large batch: two halves on two contexts, as the _c32 entry does (proposed.hip) - the upload and narrowing
    // of the second half (doubles: 9.5 GiB per 256 trials at BASELINE configs[1]) run while the first half is being solved.
    const size_t in_bytes = (size_t)batch * ((size_t)N * M * 24 + (strideB ? (size_t)G2 * M * 16 : 0));
    if (memspace == JSTSP_HOST && type == JSTSP_TYPE_APPROXIMATE && batch >= 128 && Imax > 1 && Y_out && tune().host_pipeline &&
        in_bytes >= ((size_t)512 << 20)) {
        if (!ctx->helper) JSTSP_TRY(jstsp_create(ctx->device, &ctx->helper));
        jstsp_ctx *cx[2] = {ctx, ctx->helper};
        const size_t nm1 = (size_t)N * M, g1 = (size_t)Gr * G2;
        const int h = ((batch / 2 + 7) / 8) * 8, cnt[2] = {h, batch - h}, t0[2] = {0, h};
        Conv half0(cx[0], JSTSP_HOST), half1(cx[1], JSTSP_HOST);
        Conv *cvk[2] = {&half0, &half1};
        // (a failure inside the pipeline must not return while the other half still reads the caller's host arrays or writes its
        //  outputs: both streams are drained first - as in the _c32 entry)
        auto drained = [&](int rc) {
            if (rc) for (int k = 0; k < 2; ++k) { DeviceScope ds(cx[k]->device); (void)hipStreamSynchronize(cx[k]->stream); }
            return rc;
        };
#define JSTSP_TRY_PIPE(expr) do { int rc_ = drained(expr); if (rc_ != 0) return rc_; } while (0)
        PendingSolve pend[2];
        const jstsp_c32 *yk[2], *ak[2], *bk[2];
        const float *ok[2];
        const int32_t *ik[2];
        for (int k = 0; k < 2; ++k) {
            Conv &c = *cvk[k];
            yk[k] = c.in(subY + t0[k] * nm1, cnt[k] * nm1);
            ok[k] = c.in_real(Omega + t0[k] * nm1, cnt[k] * nm1);
            ak[k] = c.in_dict(A + (size_t)t0[k] * strideA, (size_t)N * Gr, strideA, cnt[k]);
            int hgt = 0;
            bk[k] = c.in_dict_toeplitz(B + (size_t)t0[k] * strideB, G2, M, strideB, cnt[k], &hgt);
            ik[k] = c.in_raw(indx_S ? indx_S + t0[k] * g1 : nullptr, cnt[k] * g1);
            JSTSP_TRY_PIPE(c.rc);
            cx[k]->dict_block_hint = hgt;               // (consumed by the solve enqueued next on this context)
            JSTSP_TRY_PIPE(proposed_enqueue_device(cx[k], N, M, Gr, G2, cnt[k], yk[k], ok[k], ak[k], strideA, bk[k], strideB, Imax, tau_Y + t0[k],
                                              tau_S + t0[k], rho + t0[k], type, ik[k], ce_out != nullptr, &pend[k]));
        }
        int fallbacks = 0;
        for (int k = 0; k < 2; ++k) {
            Conv &c = *cvk[k];
            const PendingSolve &p = pend[k];
            JSTSP_TRY_PIPE(c.deliver(reinterpret_cast<const float *>(p.dS), reinterpret_cast<double *>(S_out + t0[k] * g1), 2 * cnt[k] * g1));
            JSTSP_TRY_PIPE(c.deliver(reinterpret_cast<const float *>(p.dY), reinterpret_cast<double *>(Y_out + t0[k] * nm1), 2 * cnt[k] * nm1));
            if (p.want_ce)
            {
                const hipError_t e_ = hipMemcpyAsync(ce_out + (size_t)t0[k] * 3 * Imax, p.dce, (size_t)cnt[k] * 3 * Imax * sizeof(double),
                                                     hipMemcpyDeviceToHost, cx[k]->stream);
                if (e_ != hipSuccess) set_error("proposed_algorithm: copying convergence_error failed: %s", hipGetErrorString(e_));
                JSTSP_TRY_PIPE((int)e_);
            }
            JSTSP_TRY_PIPE(c.rc);
            std::vector<int> o;
            JSTSP_TRY_PIPE(proposed_pending_flags(cx[k], p, &o));
            for (size_t i = 0; i < o.size();) {            // runs of flagged trials again, without the fused pass
                size_t j = i + 1;
                while (j < o.size() && o[j] == o[j - 1] + 1) ++j;
                const int r0 = o[i], nrun = (int)(j - i), tg = t0[k] + r0;
                Conv cr(cx[k], JSTSP_HOST);
                jstsp_c32 *s = cr.out(S_out + tg * g1, nrun * g1), *yo = cr.out(Y_out + tg * nm1, nrun * nm1);
                double *ce = cr.out_raw(ce_out ? ce_out + (size_t)tg * 3 * Imax : nullptr, (size_t)nrun * 3 * Imax);
                JSTSP_TRY_PIPE(cr.rc);
                JSTSP_TRY_PIPE(proposed_resolve_device(cx[k], N, M, Gr, G2, nrun, yk[k] + r0 * nm1, ok[k] + r0 * nm1, ak[k] + (size_t)r0 * strideA,
                                                  strideA, bk[k] + (size_t)r0 * strideB, strideB, Imax, tau_Y + tg, tau_S + tg, rho + tg,
                                                  type, ik[k] ? ik[k] + r0 * g1 : nullptr, s, yo, ce));
                JSTSP_TRY_PIPE(cr.finish());
                fallbacks += nrun;
                i = j;
            }
        }
#undef JSTSP_TRY_PIPE
        ctx->fused_fallbacks = fallbacks;
        ctx->last_dict_block = (cx[0]->last_dict_block == cx[1]->last_dict_block) ? cx[0]->last_dict_block : 0;
        return 0;
    }
    const size_t nm = (size_t)N * M * batch;
    int hgt = 0;
    const jstsp_c32 *y = cv.in(subY, nm), *a = cv.in_dict(A, (size_t)N * Gr, strideA, batch),
                    *b = cv.in_dict_toeplitz(B, G2, M, strideB, batch, &hgt);
    const float *om = cv.in_real(Omega, nm);
    const int32_t *ix = cv.in_raw(indx_S, (size_t)Gr * G2 * batch);
    jstsp_c32 *s = cv.out(S_out, (size_t)Gr * G2 * batch), *yo = cv.out(Y_out, nm);
    double *ce = cv.out_raw(ce_out, (size_t)Imax * 3 * batch);
    JSTSP_TRY(cv.rc);
    ctx->dict_block_hint = hgt;
    JSTSP_TRY(jstsp_proposed_algorithm_c32(ctx, N, M, Gr, G2, batch, y, om, a, strideA, b, strideB, Imax, tau_Y, tau_S,
                                           rho, type, ix, s, yo, ce, JSTSP_DEVICE));
    JSTSP_TRY(cv.finish());
    if (memspace == JSTSP_HOST && type != JSTSP_TYPE_APPROXIMATE) JSTSP_TRY(diag_check_host(ctx, "proposed_algorithm 'std'"));
    return 0;
}

int jstsp_svt_c64(jstsp_ctx *ctx, int Mr, int Mt, int batch, const jstsp_c64 *Y, const double *tau, jstsp_c64 *X,
                  int memspace)
{
    C64_BEGIN(Mr > 0 && Mt > 0 && batch > 0, "svt");
    JSTSP_REQUIRE(Y && tau && X, JSTSP_E_NULL, "svt: NULL argument");
    const size_t n = (size_t)Mr * Mt * batch;
    const jstsp_c32 *y = cv.in(Y, n);
    jstsp_c32 *x = cv.out(X, n);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_svt_c32(ctx, Mr, Mt, batch, y, tau, x, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_omp_c64(jstsp_ctx *ctx, int measures, int size_d, int batch, const jstsp_c64 *A, long long strideA,
                  const jstsp_c64 *v, int m, jstsp_c64 *x_hat, int32_t *index_out, jstsp_c64 *target_out, int memspace)
{
    C64_BEGIN(measures > 0 && size_d > 0 && batch > 0 && m > 0 && strideA >= 0, "OMP");
    JSTSP_REQUIRE(A && v && x_hat, JSTSP_E_NULL, "OMP: NULL argument");
    const jstsp_c32 *a = cv.in_dict(A, (size_t)measures * size_d, strideA, batch), *vv = cv.in(v, (size_t)measures * batch);
    jstsp_c32 *x = cv.out(x_hat, (size_t)size_d * batch), *tg = cv.out(target_out, (size_t)measures * m * batch);
    int32_t *ix = cv.out_raw(index_out, (size_t)m * batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_omp_c32(ctx, measures, size_d, batch, a, strideA, vv, m, x, ix, tg, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_sparse_admm_c64(jstsp_ctx *ctx, int Mr, int Mt, int Gr, int Gt, int batch, const jstsp_c64 *Htrue,
                          const jstsp_c64 *OH, const jstsp_c64 *Dr, const jstsp_c64 *Dt, int Imax, jstsp_c64 *S_out,
                          double *ce_out, int memspace)
{
    C64_BEGIN(Mr > 0 && Mt > 0 && Gr > 0 && Gt > 0 && batch > 0 && Imax > 0, "sparse_admm");
    JSTSP_REQUIRE(OH && Dr && Dt && S_out && (Htrue || !ce_out), JSTSP_E_NULL, "sparse_admm: NULL argument");
    const size_t n = (size_t)Mr * Mt * batch;
    const jstsp_c32 *h = cv.in(Htrue, n), *oh = cv.in(OH, n), *dr = cv.in(Dr, (size_t)Mr * Gr),
                    *dt = cv.in(Dt, (size_t)Mt * Gt);
    jstsp_c32 *s = cv.out(S_out, n);
    double *ce = cv.out_raw(ce_out, (size_t)Imax * batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_sparse_admm_c32(ctx, Mr, Mt, Gr, Gt, batch, h, oh, dr, dt, Imax, s, ce, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_mc_svt_c64(jstsp_ctx *ctx, int Mr, int Mt, int batch, const jstsp_c64 *OH, const double *Omega, int Imax,
                     const double *tau, const double *rho, jstsp_c64 *X_out, int memspace)
{
    C64_BEGIN(Mr > 0 && Mt > 0 && batch > 0 && Imax > 0, "mc_svt");
    JSTSP_REQUIRE(OH && Omega && tau && rho && X_out, JSTSP_E_NULL, "mc_svt: NULL argument");
    const size_t n = (size_t)Mr * Mt * batch;
    const jstsp_c32 *oh = cv.in(OH, n);
    const float *om = cv.in_real(Omega, n);
    jstsp_c32 *x = cv.out(X_out, n);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_mc_svt_c32(ctx, Mr, Mt, batch, oh, om, Imax, tau, rho, x, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_mc_admm_c64(jstsp_ctx *ctx, int Mr, int Mt, int batch, const jstsp_c64 *Htrue, const jstsp_c64 *OH,
                      const double *Omega, int Imax, const double *tau, const double *rho, jstsp_c64 *X_out,
                      double *ce_out, int memspace)
{
    C64_BEGIN(Mr > 0 && Mt > 0 && batch > 0 && Imax > 0, "mc_admm");
    JSTSP_REQUIRE(OH && Omega && tau && rho && X_out && (Htrue || !ce_out), JSTSP_E_NULL, "mc_admm: NULL argument");
    const size_t n = (size_t)Mr * Mt * batch;
    const jstsp_c32 *h = cv.in(Htrue, n), *oh = cv.in(OH, n);
    const float *om = cv.in_real(Omega, n);
    jstsp_c32 *x = cv.out(X_out, n);
    double *ce = cv.out_raw(ce_out, (size_t)Imax * batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_mc_admm_c32(ctx, Mr, Mt, batch, h, oh, om, Imax, tau, rho, x, ce, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_ls_c64(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c64 *Y, const jstsp_c64 *A,
                 long long strideA, const jstsp_c64 *B, long long strideB, jstsp_c64 *S_out, int memspace)
{
    C64_BEGIN(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0 && strideA >= 0 && strideB >= 0, "ls");
    JSTSP_REQUIRE(Y && A && B && S_out, JSTSP_E_NULL, "ls: NULL argument");
    const jstsp_c32 *y = cv.in(Y, (size_t)N * M * batch), *a = cv.in_dict(A, (size_t)N * Gr, strideA, batch),
                    *b = cv.in_dict(B, (size_t)G2 * M, strideB, batch);
    jstsp_c32 *s = cv.out(S_out, (size_t)Gr * G2 * batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_ls_c32(ctx, N, M, Gr, G2, batch, y, a, strideA, b, strideB, s, JSTSP_DEVICE));
    JSTSP_TRY(cv.finish());
    if (memspace == JSTSP_HOST) JSTSP_TRY(diag_check_host(ctx, "ls"));
    return 0;
}

int jstsp_pinv_c64(jstsp_ctx *ctx, int rows, int cols, int batch, const jstsp_c64 *A, jstsp_c64 *P, int memspace)
{
    C64_BEGIN(rows > 0 && cols > 0 && batch > 0, "pinv");
    JSTSP_REQUIRE(A && P, JSTSP_E_NULL, "pinv: NULL argument");
    const size_t n = (size_t)rows * cols * batch;
    const jstsp_c32 *a = cv.in(A, n);
    jstsp_c32 *p = cv.out(P, n);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_pinv_c32(ctx, rows, cols, batch, a, p, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_mmv_omp_c64(jstsp_ctx *ctx, int N, int Gr, int S, int batch, const jstsp_c64 *A, long long strideA,
                      const jstsp_c64 *Y, int K, int pnorm, jstsp_c64 *Z_out, int32_t *index_out, int32_t *count_out,
                      int memspace)
{
    C64_BEGIN(N > 0 && Gr > 0 && S > 0 && batch > 0 && K > 0 && strideA >= 0, "mmv_omp");
    JSTSP_REQUIRE(A && Y && Z_out, JSTSP_E_NULL, "mmv_omp: NULL argument");
    const jstsp_c32 *a = cv.in_dict(A, (size_t)N * Gr, strideA, batch), *y = cv.in(Y, (size_t)N * S * batch);
    jstsp_c32 *z = cv.out(Z_out, (size_t)Gr * S * batch);
    int32_t *ix = cv.out_raw(index_out, (size_t)K * batch), *cn = cv.out_raw(count_out, (size_t)batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_mmv_omp_c32(ctx, N, Gr, S, batch, a, strideA, y, K, pnorm, z, ix, cn, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_nmse_spectral_c64(jstsp_ctx *ctx, int R, int C, int batch, const jstsp_c64 *S, const jstsp_c64 *Zbar,
                            double *nmse, int memspace)
{
    C64_BEGIN(R > 0 && C > 0 && batch > 0, "nmse_spectral");
    JSTSP_REQUIRE(S && Zbar && nmse, JSTSP_E_NULL, "nmse_spectral: NULL argument");
    const size_t n = (size_t)R * C * batch;
    const jstsp_c32 *s = cv.in(S, n), *z = cv.in(Zbar, n);
    double *o = cv.out_raw(nmse, (size_t)batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_nmse_spectral_c32(ctx, R, C, batch, s, z, o, JSTSP_DEVICE));
    return cv.finish();
}

int jstsp_rate_c64(jstsp_ctx *ctx, int R, int C, int batch, const jstsp_c64 *S, const jstsp_c64 *Zbar, double noise_var,
                   double *rate, int memspace)
{
    C64_BEGIN(R > 0 && C > 0 && batch > 0, "rate");
    JSTSP_REQUIRE(S && Zbar && rate, JSTSP_E_NULL, "rate: NULL argument");
    const size_t n = (size_t)R * C * batch;
    const jstsp_c32 *s = cv.in(S, n), *z = cv.in(Zbar, n);
    double *o = cv.out_raw(rate, (size_t)batch);
    JSTSP_TRY(cv.rc);
    JSTSP_TRY(jstsp_rate_c32(ctx, R, C, batch, s, z, noise_var, o, JSTSP_DEVICE));
    return cv.finish();
}

}  // extern "C"
