// Context, workspace arena, error reporting, event-based kernel timing, and the host-side
// helpers (GEMM shorthands, batched SVT / spectral norm) of libjstsp_mi355x.so.
#include "solver_common.h"
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <algorithm>

namespace jstsp {

static thread_local char g_err[512] = "";

static thread_local Tuning g_tune;
const Tuning &tune() { return g_tune; }

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// every switch of the library, read here and nowhere else (common.h: Tuning; include/jstsp.h: "Environment")
void load_tuning()
{
    Tuning t;
    t.h2 = env_int("JSTSP_H2", t.h2);
    t.fused = env_int("JSTSP_FUSED", t.fused);
    t.fused_parts = env_int("JSTSP_FUSED_PARTS", t.fused_parts);
    t.fused_kback = env_int("JSTSP_FUSED_KBACK", t.fused_kback);
    t.toeplitz = env_int("JSTSP_TOEPLITZ", t.toeplitz);
    t.overlap = env_int("JSTSP_OVERLAP", t.overlap);
    t.lanczos = env_int("JSTSP_LANCZOS", t.lanczos);
    t.lanczos_warm = env_int("JSTSP_LANCZOS_WARM", t.lanczos_warm);
    t.lanczos_verify = env_int("JSTSP_LANCZOS_VERIFY", t.lanczos_verify);
    t.eig128 = env_int("JSTSP_EIG128", t.eig128);
    t.bj_mask = env_int("JSTSP_BJ_MASK", t.bj_mask);
    t.host_pipeline = env_int("JSTSP_HOST_PIPELINE", t.host_pipeline);
    t.host_compact = env_int("JSTSP_HOST_COMPACT", t.host_compact);
#ifdef JSTSP_EXPERIMENTS        // measured and dropped, or outside the accuracy statement: tools/ only (common.h)
    t.rv_refresh = env_int("JSTSP_RV_REFRESH", t.rv_refresh);
    t.svt_skip = env_int("JSTSP_SVT_SKIP", t.svt_skip);
    t.omp_gram = env_int("JSTSP_OMP_GRAM", t.omp_gram);
    t.bj_trace = env_int("JSTSP_BJ_TRACE", t.bj_trace);
    t.gram_refine = env_int("JSTSP_GRAM_REFINE", t.gram_refine);
    t.pass_acc = env_int("JSTSP_PASS_ACC", t.pass_acc);
    t.inv_two_float = env_int("JSTSP_INV2", t.inv_two_float);
    t.grad_head = env_int("JSTSP_GRAD_HEAD", t.grad_head);
    t.rv_always = env_int("JSTSP_RV_ALWAYS", t.rv_always);
    t.rv_comp = env_int("JSTSP_RV_COMP", t.rv_comp);
#endif
    g_tune = t;
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int Arena::reserve(size_t bytes)
{
    if (bytes <= cap) return 0;
    if (base) {
        JSTSP_HIP(hipDeviceSynchronize());
        JSTSP_HIP(hipFree(base));
        base = nullptr;
        cap = 0;
    }
    // a little headroom so that nearby shapes do not reallocate
    size_t want = bytes + (bytes >> 4) + (1u << 20);
    hipError_t e = hipMalloc((void **)&base, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        e = hipMalloc((void **)&base, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        base = nullptr;
        set_error("workspace allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return JSTSP_E_NOMEM;
    }
    cap = want;
    return 0;
}

void Arena::release()
{
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = off = 0;
}

// ---- event timing of tagged kernels ----------------------------------------------------------
static hipEvent_t take_event(jstsp_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(jstsp_ctx *ctx, const char *name)
{
    if (!ctx->profiling) return;
    ProfileSlot &s = ctx->prof[name];
    hipEvent_t a = take_event(ctx), b = take_event(ctx);
    (void)hipEventRecord(a, ctx->stream);
    s.pending.emplace_back(a, b);
}

void prof_end(jstsp_ctx *ctx, const char *name)
{
    if (!ctx->profiling) return;
    ProfileSlot &s = ctx->prof[name];
    (void)hipEventRecord(s.pending.back().second, ctx->stream);
    s.launches++;
}

static void prof_collect(jstsp_ctx *ctx)
{
    for (auto &kv : ctx->prof) {
        for (auto &pr : kv.second.pending) {
            float ms = 0.f;
            (void)hipEventSynchronize(pr.second);
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.total_ms += ms;
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

// ---- GEMM shorthand ---------------------------------------------------------------------------
GemmDesc make_gemm(char opA, char opB, int m, int n, int k, int batch, Mat A, Mat B, float2 *C, long long sCt,
                   int ldc, float alpha, const float2 *D, long long sDt, int ldd, float beta, int splitk,
                   long long sCsplit)
{
    GemmDesc d;
    d.A = A.p; d.sAt = A.st;
    // op: 'N' as stored, 'T' transpose, 'C' conjugate transpose, 'J' conjugate (no transpose)
    const bool tA = (opA == 'T' || opA == 'C'), tB = (opB == 'T' || opB == 'C');
    d.conjA = (opA == 'C' || opA == 'J');
    d.conjB = (opB == 'C' || opB == 'J');
    if (!tA) { d.sAi = 1; d.sAk = A.ld; } else { d.sAi = A.ld; d.sAk = 1; }   // a(i,kk) = A[i + ld*kk] | A[kk + ld*i]
    d.B = B.p; d.sBt = B.st; d.B2 = nullptr;
    if (!tB) { d.sBk = 1; d.sBj = B.ld; } else { d.sBk = B.ld; d.sBj = 1; }   // b(kk,j) = B[kk + ld*j] | B[j + ld*kk]
    d.C = C; d.sCt = sCt; d.ldc = ldc;
    d.D = D; d.sDt = sDt; d.ldd = ldd;
    d.alpha = alpha; d.beta = beta;
    d.m = m; d.n = n; d.k = k; d.batch = batch;
    d.splitk = splitk < 1 ? 1 : splitk; d.sCsplit = sCsplit;
    d.epi = EPI_NONE; d.prm = nullptr; d.e_rw0 = d.e_w1 = d.e_w2 = d.e_w3 = nullptr;
    d.e_r0 = d.e_r1 = d.e_r2 = d.e_r3 = nullptr; d.e_f0 = nullptr; d.epi_store_c = 1; d.amax_out = nullptr; d.amax_x = d.amax_v1 = d.amax_z = nullptr;
    d.force_m64 = 0; d.C_lo = nullptr; d.herm_upper = 0; d.D_lo = nullptr;
    d.sa_mode = 0; d.sa_lr = d.sa_lt = nullptr; d.sa_rho = d.sa_thr = 0.f;
    return d;
}

int gemm(jstsp_ctx *ctx, char opA, char opB, int m, int n, int k, int batch, Mat A, Mat B, float2 *C,
         long long sCt, int ldc, float alpha, const float2 *D, long long sDt, int ldd, float beta, int tag,
         int splitk, long long sCsplit)
{
    return launch_cgemm(ctx, make_gemm(opA, opB, m, n, k, batch, A, B, C, sCt, ldc, alpha, D, sDt, ldd, beta, splitk,
                                       sCsplit), tag);
}

// ---- Gram-form SVT ------------------------------------------------------------------------------
static int pick_nsplit(int n, int kc, int batch)
{
    const int tiles = ((n + 63) / 64) * ((n + 63) / 64);
    long long want = (1024 + (long long)batch * tiles - 1) / ((long long)batch * tiles);
    const int kmax = std::max(1, kc / 256);
    want = std::max<long long>(1, std::min<long long>(want, kmax));
    return (int)std::min<long long>(want, 32);
}

size_t GramWS::bytes(int rows, int cols, int batch, bool need_q, int force_nsplit)
{
    const int n = std::min(rows, cols), kc = std::max(rows, cols);
    const int ns = force_nsplit > 0 ? force_nsplit : pick_nsplit(n, kc, batch);
    size_t b = rnd256((size_t)batch * ns * n * n * sizeof(float2));
    if (!need_q && n <= 128) b += rnd256((size_t)batch * lanczos_ne(n) * sizeof(float2)) + rnd256((size_t)batch * sizeof(int));
    if (need_q) {
        b += rnd256((size_t)batch * n * n * sizeof(float2));
        if (n <= 64) b += rnd256((size_t)batch * eig_fast_ne(n) * eig_fast_ne(n) * sizeof(float2));
        else b += 2 * rnd256((size_t)batch * n * n * sizeof(float2));       // warm-start basis + transform temporary
        if (n > 64 && eig_needs_global_v(n)) {
            const int ne = (n + 1) & ~1;
            b += rnd256((size_t)batch * ne * ne * sizeof(float2));
        }
    }
    return b;
}

int GramWS::alloc(Arena &a, int rows_, int cols_, int batch_, bool need_q, int force_nsplit)
{
    rows = rows_; cols = cols_; batch = batch_;
    n = std::min(rows, cols);
    left = rows <= cols;
    nsplit = force_nsplit > 0 ? force_nsplit : pick_nsplit(n, std::max(rows, cols), batch);
    Gpart = a.get<float2>((size_t)batch * nsplit * n * n);
    JSTSP_REQUIRE(Gpart, JSTSP_E_NOMEM, "workspace exhausted (Gram partials)");
    Q = nullptr; Vg = nullptr; Uwarm = nullptr; Twarm = nullptr; warm = 0;
    lz = LanczosWarm();
    if (!need_q && n <= 128) {
        lz.ne = lanczos_ne(n);
        lz.x = a.get<float2>((size_t)batch * lz.ne);
        lz.state = a.get<int>((size_t)batch);
        JSTSP_REQUIRE(lz.x && lz.state, JSTSP_E_NOMEM, "workspace exhausted (Lanczos warm-start vectors)");
    }
    if (need_q) {
        Q = a.get<float2>((size_t)batch * n * n);
        JSTSP_REQUIRE(Q, JSTSP_E_NOMEM, "workspace exhausted (SVT projector)");
        if (n <= 64) {
            Uwarm = a.get<float2>((size_t)batch * eig_fast_ne(n) * eig_fast_ne(n));
            JSTSP_REQUIRE(Uwarm, JSTSP_E_NOMEM, "workspace exhausted (warm-start basis)");
        } else {
            Uwarm = a.get<float2>((size_t)batch * n * n);
            Twarm = a.get<float2>((size_t)batch * n * n);
            JSTSP_REQUIRE(Uwarm && Twarm, JSTSP_E_NOMEM, "workspace exhausted (warm-start basis)");
        }
        if (n > 64 && eig_needs_global_v(n)) {
            const int ne = (n + 1) & ~1;
            Vg = a.get<float2>((size_t)batch * ne * ne);
            JSTSP_REQUIRE(Vg, JSTSP_E_NOMEM, "workspace exhausted (eigenvectors)");
        }
    }
    return 0;
}

int gram_partials(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, long long sZt)
{
    const Mat Zm{Z, sZt, w.rows};
    const long long sG = (long long)w.n * w.n;
    if (w.left)     // G = Z Z^H, contraction over the columns
        return gemm(ctx, 'N', 'C', w.n, w.n, w.cols, w.batch, Zm, Zm, w.Gpart, sG * w.nsplit, w.n, 1.f,
                    nullptr, 0, 0, 0.f, GEMM_GRAM, w.nsplit, sG);
    // G = Z^H Z, contraction over the rows
    return gemm(ctx, 'C', 'N', w.n, w.n, w.rows, w.batch, Zm, Zm, w.Gpart, sG * w.nsplit, w.n, 1.f, nullptr,
                0, 0, 0.f, GEMM_GRAM, w.nsplit, sG);
}

int gram_partials_range(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, long long sZt, int t0, int count,
                        const uint32_t *amax, const TrialParams *skip_prm, const float2 *Z2, const TrialParams *zprm, bool norm_only)
{
    if (amax && w.left && w.n <= 64)       // split-f16 path: amax[t0 + i] bounds problem t0 + i
        return launch_hgram(ctx, Z + (long long)t0 * sZt, sZt, w.rows, w.cols, count, w.nsplit, amax + t0,
                            w.Gpart + (long long)t0 * w.n * w.n * w.nsplit, skip_prm ? skip_prm + t0 : nullptr,
                            Z2 ? Z2 + (long long)t0 * sZt : nullptr, zprm ? zprm + t0 : nullptr, norm_only);
    JSTSP_REQUIRE(!Z2, JSTSP_E_ARG, "gram_partials_range: on-the-fly Z needs the split-f16 Gram path");
    const Mat Zm{Z + (long long)t0 * sZt, sZt, w.rows};
    const long long sG = (long long)w.n * w.n;
    float2 *G = w.Gpart + (long long)t0 * sG * w.nsplit;
    if (w.left)
        return gemm(ctx, 'N', 'C', w.n, w.n, w.cols, count, Zm, Zm, G, sG * w.nsplit, w.n, 1.f, nullptr, 0, 0, 0.f,
                    GEMM_GRAM, w.nsplit, sG);
    return gemm(ctx, 'C', 'N', w.n, w.n, w.rows, count, Zm, Zm, G, sG * w.nsplit, w.n, 1.f, nullptr, 0, 0, 0.f,
                GEMM_GRAM, w.nsplit, sG);
}

int lanczos_warm_reset(jstsp_ctx *ctx, const GramWS &w)
{
    w.lz.call = 0;
    w.lz.mismatch = nullptr;
    if (!w.lz.x) return 0;
    // (the counters are CUMULATIVE over the solves of a context - a re-solve of overflowed trials, the chunks of a sweep and the
    //  second half of a pipelined host call must not wipe what the solve before them recorded; they are read AND cleared by
    //  jstsp_last_lanczos_mismatches / jstsp_debug_lanczos_counters)
    if (!ctx->lz_mismatch) {
        JSTSP_HIP(hipMalloc((void **)&ctx->lz_mismatch, 256));
        JSTSP_HIP(hipMemsetAsync(ctx->lz_mismatch, 0, 8 * sizeof(unsigned), ctx->stream));
    }
    JSTSP_HIP(hipMemsetAsync(w.lz.state, 0, (size_t)w.batch * sizeof(int), ctx->stream));
    w.lz.mismatch = ctx->lz_mismatch;
    return 0;
}

// (a record that was never reset - lz.mismatch == NULL - is not used: the vectors' states are uninitialised memory)
static const LanczosWarm *lz_of(const GramWS &w, bool lanczos) { return (lanczos && w.lz.x && w.lz.mismatch) ? &w.lz : nullptr; }

int lmax_from_partials(jstsp_ctx *ctx, const GramWS &w, float *lam, bool lanczos)
{
    const long long sG = (long long)w.n * w.n;
    return launch_lmax(ctx, w.n, w.batch, w.Gpart, sG * w.nsplit, w.nsplit, sG, lam, lanczos, lz_of(w, lanczos), 0);
}

int lmax_from_partials_range(jstsp_ctx *ctx, const GramWS &w, int first, int count, float *lam, bool lanczos)
{
    const long long sG = (long long)w.n * w.n;
    return launch_lmax(ctx, w.n, count, w.Gpart + (long long)first * sG * w.nsplit, sG * w.nsplit, w.nsplit, sG, lam + first, lanczos,
                       lz_of(w, lanczos), first);
}

int svt_prepare(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, const TrialParams *prm, const float *tau,
                bool sequence, const uint32_t *amax, bool allow_skip, const float2 *Z2, bool gram_done)
{
    const long long sZ = (long long)w.rows * w.cols;
    const long long sG = (long long)w.n * w.n;
    const bool skip = allow_skip && amax && prm && !tau && w.left && w.n <= 64;
    JSTSP_REQUIRE(!Z2 || (amax && prm), JSTSP_E_ARG, "svt_prepare: on-the-fly Z needs the split-f16 Gram path");
    if (gram_done) { /* nothing */ }
    else if (amax) JSTSP_TRY(gram_partials_range(ctx, w, Z, sZ, 0, w.batch, amax, skip ? prm : nullptr, Z2, Z2 ? prm : nullptr));
    else JSTSP_TRY(gram_partials(ctx, w, Z, sZ));
    if (w.n <= 64) {
        JSTSP_TRY(launch_eig_fast(ctx, EIG_SVT_Q, w.n, w.batch, w.Gpart, sG * w.nsplit, w.nsplit, sG, prm, tau,
                                  w.Q, nullptr, w.Uwarm, sequence ? w.warm : 0, skip ? amax : nullptr));
        w.warm = sequence ? 1 : 0;
        return 0;
    }
    // orders 65..128: G in LDS, the eigenvector basis in registers (eig3.hip); JSTSP_EIG128=0: the general kernel with
    // the basis in HBM (eig.hip)
    if (w.n <= 128 && tune().eig128 != 0) {
        // warm start (successive calls of an ADMM loop): G <- Uw^H (G Uw) with the previous basis, two batched GEMMs
        const int warm = (sequence && w.warm && w.nsplit == 1 && w.Uwarm && w.Twarm) ? 1 : 0;
        if (warm) {
            const Mat Gm{w.Gpart, sG, w.n}, Um{w.Uwarm, sG, w.n};
            JSTSP_TRY(gemm(ctx, 'N', 'N', w.n, w.n, w.n, w.batch, Gm, Um, w.Twarm, sG, w.n));
            JSTSP_TRY(gemm(ctx, 'C', 'N', w.n, w.n, w.n, w.batch, Um, Mat{w.Twarm, sG, w.n}, w.Gpart, sG, w.n));
        }
        JSTSP_TRY(launch_eig128(ctx, w.n, w.batch, w.Gpart, sG * w.nsplit, w.nsplit, sG, prm, tau, w.Q,
                                sequence ? w.Uwarm : nullptr, warm, 16, warm ? w.eig_stop : 0.f));
        w.warm = sequence ? 1 : 0;
        return 0;
    }
    return launch_eig(ctx, EIG_SVT_Q, w.n, w.batch, w.Gpart, sG * w.nsplit, w.nsplit, sG, prm, tau, w.Q, nullptr,
                      w.Vg);
}

int svt_apply(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, float2 *Y)
{
    const long long sZ = (long long)w.rows * w.cols;
    const long long sG = (long long)w.n * w.n;
    const Mat Zm{Z, sZ, w.rows}, Qm{w.Q, sG, w.n};
    if (w.left)     // Y = Z - Q Z
        return gemm(ctx, 'N', 'N', w.rows, w.cols, w.n, w.batch, Qm, Zm, Y, sZ, w.rows, -1.f, Z, sZ, w.rows,
                    1.f);
    // Y = Z - Z Q
    return gemm(ctx, 'N', 'N', w.rows, w.cols, w.n, w.batch, Zm, Qm, Y, sZ, w.rows, -1.f, Z, sZ, w.rows, 1.f);
}

int svt_batched(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, const TrialParams *prm,
                const float *tau, float2 *Y, bool sequence)
{
    JSTSP_TRY(svt_prepare(ctx, w, Z, prm, tau, sequence));
    return svt_apply(ctx, w, Z, Y);
}

int ensure_side_streams(jstsp_ctx *ctx)
{
    // (default priority: lowest-priority side streams measured 3.96 vs 3.92 ms per iteration at configs[1], a high-priority
    //  svt chain no better - rounds 2 and 3; the switch for it is gone)
    for (int i = 0; i < 2; ++i)
        if (!ctx->side[i]) JSTSP_HIP(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
    for (int i = 0; i < 10; ++i)
        if (!ctx->ev[i]) JSTSP_HIP(hipEventCreateWithFlags(&ctx->ev[i], hipEventDisableTiming));
    return 0;
}

int ensure_bj_resources(jstsp_ctx *ctx)
{
    for (int i = 0; i < 3; ++i)
        if (!ctx->bj_stream[i]) JSTSP_HIP(hipStreamCreateWithFlags(&ctx->bj_stream[i], hipStreamNonBlocking));
    for (int i = 0; i < 6; ++i)
        if (!ctx->bj_ev[i]) JSTSP_HIP(hipEventCreateWithFlags(&ctx->bj_ev[i], hipEventDisableTiming));
    return 0;
}

// Compute-unit masks: 32 of the units are set aside (4 per XCD whichever way the mask bits are dealt over the XCDs: bit
// 32 x + 8 y + (x + y) % 8 for x < 8, y < 4 lands in XCD x under "32 consecutive bits per XCD" and in XCD (x + y) % 8 under
// "bits dealt round-robin"), the other streams get the complement.
bool ensure_cu_streams(jstsp_ctx *ctx)
{
    if (ctx->cu_state) return ctx->cu_state > 0;
    ctx->cu_state = -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess || prop.multiProcessorCount != 256) return false;
    uint32_t keep[8], rest[8];
    for (int w = 0; w < 8; ++w) { keep[w] = 0u; rest[w] = 0xffffffffu; }
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 4; ++y) {
            const int bit = 32 * x + 8 * y + (x + y) % 8;
            keep[bit >> 5] |= 1u << (bit & 31);
            rest[bit >> 5] &= ~(1u << (bit & 31));
        }
    for (int i = 0; i < 4; ++i)
        if (hipExtStreamCreateWithCUMask(&ctx->cu_stream[i], 8, i == 3 ? keep : rest) != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < i; ++j) { (void)hipStreamDestroy(ctx->cu_stream[j]); ctx->cu_stream[j] = nullptr; }
            ctx->cu_stream[i] = nullptr;
            return false;
        }
    ctx->cu_state = 1;
    return true;
}

// ---- conditioning record ------------------------------------------------------------------------
static const uint32_t DIAG_INIT[3] = {0x7f800000u /* +inf: no pinv yet */, 0u, 0x7f800000u /* no Gram inverse yet */};

int ensure_diag(jstsp_ctx *ctx)
{
    if (ctx->diag) return 0;
    JSTSP_HIP(hipMalloc((void **)&ctx->diag, sizeof(DIAG_INIT)));
    return upload(ctx, ctx->diag, DIAG_INIT, sizeof(DIAG_INIT));
}

int diag_reset(jstsp_ctx *ctx)
{
    if (!ctx->diag) return ensure_diag(ctx);
    return upload(ctx, ctx->diag, DIAG_INIT, sizeof(DIAG_INIT));
}

static int diag_read(jstsp_ctx *ctx, double *rcond_min, double *res_max, double *gram_ratio = nullptr)
{
    uint32_t h[3] = {DIAG_INIT[0], DIAG_INIT[1], DIAG_INIT[2]};
    if (ctx->diag) JSTSP_HIP(hipMemcpy(h, ctx->diag, sizeof(h), hipMemcpyDeviceToHost));
    float f0, f1, f2;
    memcpy(&f0, &h[0], 4); memcpy(&f1, &h[1], 4); memcpy(&f2, &h[2], 4);
    // one scale for both routes: sigma_min/sigma_max of the factor (= sqrt of the Gram's eigenvalue ratio)
    double rc = 1.0;
    if (h[0] != DIAG_INIT[0]) rc = std::min(rc, (double)f0);
    if (h[2] != DIAG_INIT[2]) rc = std::min(rc, std::sqrt((double)f2));
    if (rcond_min) *rcond_min = rc;
    if (res_max) *res_max = (double)f1;
    if (gram_ratio) *gram_ratio = (h[2] == DIAG_INIT[2]) ? 1.0 : (double)f2;
    return 0;
}

int diag_check_host(jstsp_ctx *ctx, const char *what)
{
    double rc = 1.0, res = 0.0, gr = 1.0;
    JSTSP_TRY(diag_read(ctx, &rc, &res, &gr));
    JSTSP_REQUIRE(gr >= 1e-6, JSTSP_E_ILLCOND,
                  "%s: a factor Gram has lambda_min/lambda_max = %.3g < 1e-6: its fp32 inverse (factors too large for "
                  "the float64 pinv kernel) would carry no correct digit", what, gr);
    JSTSP_REQUIRE(res < 1e-2, JSTSP_E_ILLCOND,
                  "%s: the Newton-Schulz inverse of a factor Gram of order > 128 did not converge (max|I - G X| = %.3g): "
                  "the factor is rank-deficient or too ill-conditioned for fp32", what, res);
    return 0;
}

int sigma_max_sq(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, float *lam, bool lanczos)
{
    const long long sZ = (long long)w.rows * w.cols;
    const long long sG = (long long)w.n * w.n;
    JSTSP_TRY(gram_partials(ctx, w, Z, sZ));
    // (with a reset warm-start record: one call per iteration of the owning loop - the record's call counter advances here)
    JSTSP_TRY(launch_lmax(ctx, w.n, w.batch, w.Gpart, sG * w.nsplit, w.nsplit, sG, lam, lanczos, lz_of(w, lanczos), 0));
    if (lanczos) ++w.lz.call;
    return 0;
}

int upload(jstsp_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    // Stream-ordered upload of a small parameter block through a pinned staging buffer owned by
    // the context, so the caller's source may die immediately and earlier in-flight work on
    // the stream is not disturbed.  The buffer is recycled once its last copy has completed.
    if (ctx->pinned_off + bytes > ctx->pinned_cap) {
        if (ctx->pinned_pending) {
            JSTSP_HIP(hipEventSynchronize(ctx->pinned_done));
            ctx->pinned_pending = false;
        }
        ctx->pinned_off = 0;
        if (bytes > ctx->pinned_cap) {
            if (ctx->pinned) (void)hipHostFree(ctx->pinned);
            ctx->pinned = nullptr;
            const size_t cap = std::max(bytes, (size_t)1 << 20);
            JSTSP_HIP(hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
            ctx->pinned_cap = cap;
        }
    }
    char *stage = (char *)ctx->pinned + ctx->pinned_off;
    memcpy(stage, src, bytes);
    ctx->pinned_off += (bytes + 63) & ~size_t(63);
    JSTSP_HIP(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->pinned_done) JSTSP_HIP(hipEventCreateWithFlags(&ctx->pinned_done, hipEventDisableTiming));
    JSTSP_HIP(hipEventRecord(ctx->pinned_done, ctx->stream));
    ctx->pinned_pending = true;
    return 0;
}

}  // namespace jstsp

using namespace jstsp;

extern "C" {

const char *jstsp_last_error(void) { return g_err; }
const char *jstsp_version(void) { return "jstsp-mi355x 0.1 (gfx950)"; }

int jstsp_create(int device_id, jstsp_ctx **out)
{
    JSTSP_REQUIRE(out, JSTSP_E_NULL, "jstsp_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("jstsp_create: no HIP device available (%s) — this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return e != hipSuccess ? (int)e : (int)hipErrorNoDevice;
    }
    JSTSP_REQUIRE(device_id >= 0 && device_id < ndev, JSTSP_E_ARG, "jstsp_create: device %d of %d", device_id,
                  ndev);
    DeviceScope dev_scope_(device_id);          // the caller's current device is restored on return
    JSTSP_HIP(dev_scope_.err);
    jstsp_ctx *c = new (std::nothrow) jstsp_ctx();
    JSTSP_REQUIRE(c, JSTSP_E_NOMEM, "jstsp_create: out of host memory");
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete c;
        set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
        return (int)se;
    }
    c->own_stream = true;
    *out = c;
    return 0;
}

int jstsp_destroy(jstsp_ctx *ctx)
{
    if (!ctx) return 0;
    if (ctx->helper) { (void)jstsp_destroy(ctx->helper); ctx->helper = nullptr; }
    DeviceScope dev_scope_(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    prof_collect(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->pinned_done) (void)hipEventDestroy(ctx->pinned_done);
    for (int i = 0; i < 2; ++i) if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); }
    for (int i = 0; i < 3; ++i) if (ctx->bj_stream[i]) { (void)hipStreamSynchronize(ctx->bj_stream[i]); (void)hipStreamDestroy(ctx->bj_stream[i]); }
    for (int i = 0; i < 6; ++i) if (ctx->bj_ev[i]) (void)hipEventDestroy(ctx->bj_ev[i]);
    for (int i = 0; i < 4; ++i) if (ctx->cu_stream[i]) { (void)hipStreamSynchronize(ctx->cu_stream[i]); (void)hipStreamDestroy(ctx->cu_stream[i]); }
    for (int i = 0; i < 10; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->diag) (void)hipFree(ctx->diag);
    if (ctx->lz_mismatch) (void)hipFree(ctx->lz_mismatch);
    if (ctx->hpin_done) { (void)hipEventSynchronize(ctx->hpin_done); (void)hipEventDestroy(ctx->hpin_done); }
    if (ctx->hpin) (void)hipHostFree(ctx->hpin);
    if (ctx->unit) (void)hipFree(ctx->unit);
    if (ctx->unit64) (void)hipFree(ctx->unit64);
    ctx->arena.release();
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int jstsp_set_stream(jstsp_ctx *ctx, void *hip_stream)
{
    // NULL is a valid handle: HIP's default (null) stream — what torch.cuda.current_stream().cuda_stream
    // is unless the caller opened a stream context.  The library's work must be ordered on the SAME stream
    // as the caller's producers / consumers of the device arrays, so it is taken literally.
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_ENTER(ctx);
    if (ctx->stream == (hipStream_t)hip_stream && !ctx->own_stream) return 0;
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return 0;
}

int jstsp_use_own_stream(jstsp_ctx *ctx)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_ENTER(ctx);
    if (ctx->own_stream) return 0;
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    JSTSP_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
    return 0;
}

int jstsp_synchronize(jstsp_ctx *ctx)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_ENTER(ctx);
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_last_conditioning(jstsp_ctx *ctx, double *rcond_min, double *ns_residual_max)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_ENTER(ctx);
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return diag_read(ctx, rcond_min, ns_residual_max);
}

// the four counters of this context AND of its helper context (the second half of a pipelined JSTSP_HOST call runs there),
// summed into out4 and cleared: what was recorded since the previous read
static int lanczos_counters_take(jstsp_ctx *ctx, unsigned *out4)
{
    jstsp_ctx *cx[2] = {ctx, ctx->helper};
    for (int k = 0; k < 2; ++k) {
        if (!cx[k] || !cx[k]->lz_mismatch) continue;
        DeviceScope ds(cx[k]->device);
        unsigned h[4];
        JSTSP_HIP(hipStreamSynchronize(cx[k]->stream));
        JSTSP_HIP(hipMemcpy(h, cx[k]->lz_mismatch, sizeof(h), hipMemcpyDeviceToHost));
        JSTSP_HIP(hipMemset(cx[k]->lz_mismatch, 0, 8 * sizeof(unsigned)));
        for (int i = 0; i < 4; ++i) out4[i] += h[i];
    }
    return 0;
}

// (diagnostics, not in the header: [0] verification mismatches, [1] warm attempts that did not converge, [2] verifications,
//  [3] Lanczos steps of the converged warm attempts - since the previous read of the counters)
extern "C" int jstsp_debug_lanczos_counters(jstsp_ctx *ctx, unsigned *out4)
{
    JSTSP_REQUIRE(ctx && out4, JSTSP_E_NULL, "ctx/out is NULL");
    JSTSP_ENTER(ctx);
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    return lanczos_counters_take(ctx, out4);
}

int jstsp_last_lanczos_mismatches(jstsp_ctx *ctx, int *count)
{
    JSTSP_REQUIRE(ctx && count, JSTSP_E_NULL, "ctx/count is NULL");
    JSTSP_ENTER(ctx);
    unsigned h[4] = {0, 0, 0, 0};
    JSTSP_TRY(lanczos_counters_take(ctx, h));
    *count = (int)h[0];
    return 0;
}

size_t jstsp_workspace_bytes(const jstsp_ctx *ctx) { return ctx ? ctx->arena.cap : 0; }

int jstsp_set_profiling(jstsp_ctx *ctx, int enable)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_ENTER(ctx);
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    ctx->profiling = enable != 0;
    if (enable) for (auto &kv : ctx->prof) { kv.second.launches = 0; kv.second.total_ms = 0; }
    return 0;
}

int jstsp_get_profile(jstsp_ctx *ctx, const char *kernel, int *launches, double *total_ms)
{
    JSTSP_REQUIRE(ctx && kernel, JSTSP_E_NULL, "ctx/kernel is NULL");
    JSTSP_ENTER(ctx);
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    auto it = ctx->prof.find(kernel);
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.total_ms;
    return 0;
}

}  // extern "C"
