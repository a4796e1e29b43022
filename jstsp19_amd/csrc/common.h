// Internal declarations shared by the translation units of libjstsp_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>
#include <vector>
#include <map>
#include "../../include/jstsp.h"

namespace jstsp {

void set_error(const char *fmt, ...);

#define JSTSP_HIP(call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            jstsp::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                             __FILE__, __LINE__);                                        \
            return (int)e_;                                                              \
        }                                                                                \
    } while (0)

#define JSTSP_TRY(expr)                                                                  \
    do {                                                                                 \
        int rc_ = (expr);                                                                \
        if (rc_ != 0) return rc_;                                                        \
    } while (0)

#define JSTSP_REQUIRE(cond, code, ...)                                                   \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            jstsp::set_error(__VA_ARGS__);                                               \
            return (code);                                                               \
        }                                                                                \
    } while (0)

// Makes the context's device current for the duration of an API call and restores the caller's current device on
// every exit path (a caller that keeps its tensors on another GPU must not find its current device changed).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
// Environment switches of the library (JSTSP_*; the list and what each one is for: include/jstsp.h, "Environment").  They
// are diagnostic / opt-in settings, parsed ONCE at the entry of every API call (JSTSP_ENTER) into this struct - no switch is
// latched in a static, none is read anywhere else - so a test can change them between two calls of one process and a call
// sees one consistent setting from its first launch to its last.
// Round 6: the opt-in paths that were measured and dropped (HISTORY.md), and every switch whose non-default value leaves the accuracy
// statement of include/jstsp.h, exist only in a -DJSTSP_EXPERIMENTS build (JSTSP_EXPERIMENTS=1 python jstsp19_amd/build.py; tools/
// only).  In the shipped library their settings are compile-time constants: the environment variable is not read, the branch is dead.
#ifdef JSTSP_EXPERIMENTS
#define JSTSP_XP int
inline const char *xp_getenv(const char *name) { return getenv(name); }
#else
#define JSTSP_XP static constexpr int
inline const char *xp_getenv(const char *) { return nullptr; }
#endif
struct Tuning {
    int h2 = 1;             // JSTSP_H2: 0 strict complex-fp32 MFMA everywhere, 1 split-f16 MFMA for big contractions, 2 always
    int fused = 1;          // JSTSP_FUSED: 0 three-kernel ADMM iteration instead of the fused pass
    int fused_parts = 0;    // JSTSP_FUSED_PARTS: column ranges per problem in the pass (0: chosen from M)
    int fused_kback = 4;    // JSTSP_FUSED_KBACK: headroom bits of the predicted k scale (test hook: negative forces the re-solve)
    int toeplitz = 2;       // JSTSP_TOEPLITZ: 0 dictionary taken as unstructured, 1 compact image only, 2 + window kernel (block 64)
    JSTSP_XP rv_refresh = 4;     // (experiments build) JSTSP_RV_REFRESH: R v recomputed from v every this many iterations
    int overlap = -1;       // JSTSP_OVERLAP: side streams between the kernels of an iteration (-1: on with the fused pass)
    JSTSP_XP svt_skip = 0;       // (experiments build) JSTSP_SVT_SKIP: 1 trials whose threshold is below fp32 resolution skip the eigen-decomposition (opt-in)
    int lanczos = 1;        // JSTSP_LANCZOS: 0 Householder + Sturm instead of Lanczos for the convergence_error norms
    int lanczos_warm = 1;   // JSTSP_LANCZOS_WARM: 0 every lambda_max of an ADMM loop by the cold n-step Lanczos run (rounds 2-4)
    int lanczos_verify = 32; // JSTSP_LANCZOS_VERIFY: every this many calls a warm-started lambda_max is checked against the cold run (0: never, 1: always)
    int eig128 = 1;         // JSTSP_EIG128: 0 general Jacobi kernel for Gram orders 65..128
    JSTSP_XP omp_gram = 1;       // (experiments build) JSTSP_OMP_GRAM: 0 measurement-space OMP on a Kronecker dictionary
    JSTSP_XP grad_head = 0;      // (experiments build) JSTSP_GRAD_HEAD: bit 0 - Res / P1 of the gradient step, bit 1 - the first factor of a recomputed R v, on the
                            // f16 pipe in one launch (hsmall.hip) instead of fp32-MFMA products; measured: more accurate products,
                            // +2 % channel-estimates/s, but WORSE parity (rms |dNMSE| 2.18e-7 against 1.73e-7, a +5e-8 bias): off
    JSTSP_XP rv_comp = 0;        // (experiments build) JSTSP_RV_COMP: 1 v and R v carried as two floats each (compensated accumulation of alpha res / alpha R res)
    JSTSP_XP rv_always = 0;      // (experiments build) JSTSP_RV_ALWAYS: R v recomputed from v in each of the first n iterations (then every JSTSP_RV_REFRESH-th)
    JSTSP_XP inv_two_float = 1;  // (experiments build) JSTSP_INV2: 0 the pass reads 1 / (Omega + 2 rho) as one rounded float per entry (rounds 1-4)
    JSTSP_XP pass_acc = 1;       // (experiments build; applies to the window pass fused_pass64_kernel only - fused_pass_kernel always sums per tile first) JSTSP_PASS_ACC: 0 products of K B^H straight into the window pass's running sums (rounds 2-4), 1 per-tile block sums first (fused.hip)
    int host_compact = 1;   // JSTSP_HOST_COMPACT: 0 a JSTSP_HOST dictionary is uploaded whole (no host-side block-Toeplitz test / compaction)
    int host_pipeline = 1;  // JSTSP_HOST_PIPELINE: 0 a JSTSP_HOST solve as ONE staged call (no overlap of the copies with the solve)
    JSTSP_XP gram_refine = 1;    // (experiments build) JSTSP_GRAM_REFINE: 0 the dictionary Grams G_A, G_B as plain fp32 products, no low-order parts in R*v
    int bj_mask = 1;        // JSTSP_BJ_MASK: 0 the block Jacobi above order 128 without compute-unit masks (its sub-problems then compete with the panel products for units)
    JSTSP_XP bj_trace = 0;       // (experiments build) JSTSP_BJ_TRACE: 1 print the block Jacobi's convergence per sweep (stderr)
};
const Tuning &tune();       // the calling thread's setting, as parsed by the API call in progress
void load_tuning();

#define JSTSP_ENTER(ctx)                                                                 \
    jstsp::DeviceScope dev_scope_((ctx)->device);                                        \
    JSTSP_HIP(dev_scope_.err);                                                           \
    jstsp::load_tuning()

// Per-problem scalars of the ADMM solvers, resident on the device.
// Round 5: every coefficient of the iteration map is held as TWO floats, hi + lo, all derived in float64 from the caller's
// double rho.  Rounding each of rho, 1/rho, rho/(rho+1), 1 - rho ... to fp32 ON ITS OWN breaks the relations between them
// (irho * rho = 1, 1 - c = 1/(1 + rho)) at the 3e-8 level - a CONSTANT perturbation of the iteration map, applied to every
// entry in every iteration, that the dual variables integrate: measured with the float64 restatement (tools/precision_study.py,
// mask 1048576) it alone costs 1.0e-7 rms of dNMSE at BASELINE configs[1] - more than all fp32 storage roundings together
// (0.4e-7) - while a CONSISTENT fp32 rho costs 2e-9.  A product c * x is formed as fma(c_hi, x, c_lo * x): one random
// rounding of the result, no systematic one.
struct TrialParams {
    float rho, irho;        // rho, 1/rho                                   (hi parts)
    float tauY_rho;         // tau_Y / rho   (svt threshold, proposed_algorithm.m:35)
    float tauS_rho;         // tau_S / rho   (soft threshold, :56)
    float c_coef;           // rho/(rho+1)   (:61)
    float rho_lo, irho_lo, c_lo;            // value - (float)value of the three above
    float omc, omc_lo;      // 1 - rho/(rho+1) = 1/(1+rho)
    float omr, omr_lo;      // 1 - rho
    float omir, omir_lo;    // 1 - 1/rho
    float pad[2];
};
#if defined(__cplusplus)
inline TrialParams make_trial_params(double rho, double tau_Y, double tau_S)
{
    TrialParams p;
    auto split = [](double v, float &h, float &l) { h = (float)v; l = (float)(v - (double)h); };
    split(rho, p.rho, p.rho_lo);
    split(1.0 / rho, p.irho, p.irho_lo);
    split(rho / (rho + 1.0), p.c_coef, p.c_lo);
    split(1.0 / (rho + 1.0), p.omc, p.omc_lo);
    split(1.0 - rho, p.omr, p.omr_lo);
    split(1.0 - 1.0 / rho, p.omir, p.omir_lo);
    p.tauY_rho = (float)(tau_Y / rho);
    p.tauS_rho = (float)(tau_S / rho);
    p.pad[0] = p.pad[1] = 0.f;
    return p;
}
#if defined(__HIPCC__)
// c * x with c = ch + cl
__device__ __forceinline__ float mul2(float ch, float cl, float x) { return fmaf(ch, x, cl * x); }
// The element-wise updates of an ADMM iteration (C == -V2 eliminated: DESIGN.md section 3), one real component, the same
// expressions in every kernel that applies them (fused.hip, cgemm.hip / hgemm.hip epilogues):
//   V2 <- (1 - cc)(V2 - rho (X - Xs))                                   (:61 + :65)
__device__ __forceinline__ float admm_v2(const TrialParams &p, float v2, float x, float xs)
{
    const float dx = x - xs;
    const float t = fmaf(-p.rho, dx, v2) - p.rho_lo * dx;
    return mul2(p.omc, p.omc_lo, t);
}
//   X <- (V1 + rho Y + subY + (1 - rho) V2 + rho Xs) / (Omega + 2 rho)    (:38-40 with V2 + rho C = (1 - rho) V2)
__device__ __forceinline__ float admm_x(const TrialParams &p, float v1, float y, float sy, float v2, float xs, float invd)
{
    const float ys = y + xs;
    return ((v1 + sy) + mul2(p.rho, p.rho_lo, ys) + mul2(p.omr, p.omr_lo, v2)) * invd;
}
//   the same with 1 / (Omega + 2 rho) as two floats (admm_invd)
__device__ __forceinline__ float admm_x2(const TrialParams &p, float v1, float y, float sy, float v2, float xs, float rh, float rl)
{
    const float ys = y + xs;
    const float s = (v1 + sy) + mul2(p.rho, p.rho_lo, ys) + mul2(p.omr, p.omr_lo, v2);
    return fmaf(s, rh, s * rl);
}
//   1 / (Omega + 2 rho) = rh + rl: the sum as two floats (two-sum + the low part of rho), a 1-ulp reciprocal of its high part and
//   one Newton correction carrying the low parts.  (An fp32 ARRAY of these values - iK1 of :14-20 - rounds 1/(2 rho) and
//   1/(1 + 2 rho) once and applies the same two relative errors to every entry in every iteration: 0.5e-7 rms of dNMSE,
//   tools/precision_study.py mask 4194304.)
__device__ __forceinline__ void admm_invd(const TrialParams &p, float om, float &rh, float &rl)
{
    const float a = 2.f * p.rho, al = 2.f * p.rho_lo;
    const float dh = om + a;
    const float bb = dh - om;
    const float dl = ((om - (dh - bb)) + (a - bb)) + al;
    rh = __builtin_amdgcn_rcpf(dh);
    rl = rh * (fmaf(-dh, rh, 1.f) - dl * rh);
}
//   k = X - V2/rho - C = X + (1 - 1/rho) V2                               (:43)
__device__ __forceinline__ float admm_k(const TrialParams &p, float x, float v2) { return fmaf(p.omir, v2, x) + p.omir_lo * v2; }
//   V1 <- V1 + rho (Y - X)                                                (:64)
__device__ __forceinline__ float admm_v1(const TrialParams &p, float v1, float y, float x)
{
    const float d = y - x;
    return fmaf(p.rho, d, v1) + p.rho_lo * d;
}
//   Z = X - V1/rho                                                        (:35)
__device__ __forceinline__ float admm_z(const TrialParams &p, float x, float v1) { return fmaf(-p.irho, v1, x) - p.irho_lo * v1; }
#endif
#endif

// Grow-only device arena: one hipMalloc'd slab, bump allocation, reset per call.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    int reserve(size_t bytes);           // ensure capacity (may reallocate; invalidates pointers)
    void reset() { off = 0; }
    template <class T> T *get(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        if (off + bytes > cap) return nullptr;
        T *p = reinterpret_cast<T *>(base + off);
        off += bytes;
        return p;
    }
    void release();
};

struct ProfileSlot {
    int launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0;
};

}  // namespace jstsp

struct jstsp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    jstsp::Arena arena;
    bool profiling = false;
    std::map<std::string, jstsp::ProfileSlot> prof;
    std::vector<hipEvent_t> event_pool;
    int num_cus = 256;
    // pinned staging buffer for small host->device parameter blocks (stream-ordered uploads)
    void *pinned = nullptr;
    size_t pinned_cap = 0, pinned_off = 0;
    hipEvent_t pinned_done = nullptr;
    bool pinned_pending = false;
    // side streams + events used by the ADMM driver to run the next iteration's SVT preparation
    // and the convergence-error norms concurrently with the MFMA-bound GEMMs of the main stream
    // conditioning record of the last call that (pseudo-)inverted something (pinv.hip / hinv.hip), device memory:
    // float bits of [0] the smallest sigma_min/sigma_max met by the float64 pinv kernel, [1] the largest Newton-Schulz
    // residual max|I - G X|, [2] the smallest lambda_min/lambda_max of an eigen-inverted factor Gram
    uint32_t *diag = nullptr;
    // [0] warm-started lambda_max values of the last ADMM solve that a periodic cold verification contradicted (eig2.hip:
    // lanczos_lmax_kernel; jstsp_last_lanczos_mismatches), device memory
    unsigned *lz_mismatch = nullptr;
    // pinned staging buffer of the host-side block-Toeplitz compaction (hostpack.hip), grow-only; hpin_done: its last upload
    void *hpin = nullptr;
    size_t hpin_cap = 0;
    hipEvent_t hpin_done = nullptr;
    bool hpin_pending = false;
    int dict_block_hint = 0;     // set by a caller that has staged an EXPANDED block-Toeplitz dictionary itself (c64.hip): the block height,
                                 // consumed by the next proposed_algorithm solve on this context instead of the device probe
    int last_dict_block = 0;     // block height of the block-Toeplitz structure the last fused solve found in its dictionary (0: none)
    int fused_fallbacks = 0;     // trials of the last proposed_algorithm call re-solved after a k-scale overflow in the fused pass
    float2 *unit = nullptr;      // device copy of the 1 x 1 identity factor (vamp.hip: the dense call is the Kronecker call with it)
    double2 *unit64 = nullptr;   // the same for the float64 path (vamp64.hip)
    jstsp_ctx *helper = nullptr; // second context of the same device (own stream, own workspace): the other half of a pipelined JSTSP_HOST solve (proposed.hip)
    hipStream_t side[2] = {nullptr, nullptr};
    // streams with a compute-unit mask (runtime.hip: ensure_cu_streams): [0..2] everything but 32 reserved units, [3] the
    // reserved units only - the block Jacobi's chain of sub-problems runs there beside panel products that fill the rest
    hipStream_t cu_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    int cu_state = 0;            // 0 not tried, 1 available, -1 the runtime refused
    // the block Jacobi's own plain streams and events (eig_large.hip is called from inside solvers that are using side[] and
    // ev[] themselves - an SVT of order > 128 on a side stream of proposed_algorithm - and must not re-record their events)
    hipStream_t bj_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t bj_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace jstsp {

// ---- generic batched complex GEMM on the fp32 MFMA (cgemm.hip) ------------------------
// C[t] (m x n, column-major, ld = ldc) = alpha * opA(A[t]) * opB(B[t]) + beta * D[t]
// with element addressing  a(i,kk) = A[t*sAt + i*sAi + kk*sAk]  (conjugated if conjA)
//                          b(kk,j) = B[t*sBt + kk*sBk + j*sBj]  (conjugated if conjB).
// splitk > 1: the k range is cut in `splitk` chunks, chunk s written to C + s*sCsplit
// (the consumer sums; alpha applied, beta/D ignored).
struct GemmDesc {
    const float2 *A; long long sAt, sAi, sAk; int conjA;
    const float2 *B; long long sBt, sBk, sBj; int conjB;
    const float2 *B2;           // EPI_UPDATE_X only, may be NULL: b = B - prm[t].irho * B2 (same strides: Z = X - V1/rho on the fly)
    float2 *C; long long sCt; int ldc;
    const float2 *D; long long sDt; int ldd;
    float alpha, beta;
    int m, n, k, batch;
    int splitk; long long sCsplit;
    // Fused ADMM epilogues (EPI_*): extra N x M arrays indexed like C (t*sCt + i + ldc*j)
    int epi;
    const TrialParams *prm;
    float2 *e_rw0, *e_w1, *e_w2, *e_w3;     // in/out and output arrays
    const float2 *e_r0, *e_r1, *e_r2, *e_r3;
    const float *e_f0;
    int epi_store_c;                        // EPI_UPDATE_X: 1 also store Y into C; 2 store Y ONLY (no update applied)
    uint32_t *amax_x, *amax_v1, *amax_z;    // EPI_UPDATE_X: optional [batch] maxima of the new X, V1, Znext
    uint32_t *amax_out;                     // optional [batch]: atomicMax of max(|re|,|im|) (float bits) over the
                                            // stored product (EPI_NONE) / over K (EPI_UPDATE_X); caller zeroes it
    // Products that become OPERATORS of the iteration (the Grams of the dictionary factors, gram64.hip's note): fp64 master
    // accumulators whatever k (force_m64), and the part of the float64 sum that the fp32 result C does not hold stored beside
    // it (C_lo, same layout as C; alpha = 1, no D): C + C_lo carries the product to about 1e-9 relative
    int force_m64;
    float2 *C_lo;
    const float2 *D_lo;                     // low-order part of D (same layout): C = alpha acc + beta (D + D_lo); NULL = none
    int herm_upper;                         // the product is Hermitian (a Gram): tiles entirely below the diagonal are skipped, the caller
                                            // mirrors them (hermitian_fill_lower)
    int sa_mode;                            // EPI_SADMM (below)
    const float *sa_lr, *sa_lt;
    float sa_rho, sa_thr;
};
// element-wise steps of sparse_admm.m, shared by the stand-alone kernels (sparse_admm.hip) and the EPI_SADMM epilogue so that both
// round the same way (explicit fmaf: no contraction left to the context)
__device__ __forceinline__ float sadmm_den(float lr, float lt, float rho) { return fmaf(lr, lt, -rho); }
__device__ __forceinline__ float sadmm_dual1(float z, float r, float s, float rho) { return fmaf(rho, r - s, z); }          // :30
__device__ __forceinline__ float sadmm_soft1(float r, float z, float ir, float thr)                                          // :21-22
{
    const float v = fmaf(ir, z, r);
    const float m = fmaxf(fabsf(v) - thr, 0.f);
    return (v > 0.f) ? m : ((v < 0.f) ? -m : 0.f);
}
__device__ __forceinline__ float sadmm_rhs1(float z, float s, float a, float rho) { return fmaf(-rho, s, z) + a; }           // :26
// The N x M array C of the reference is never stored: with cc = rho/(rho+1), D = X - Xs,
//   C   = cc (D - V2/rho)                       (proposed_algorithm.m:61)
//   V2' = V2 + rho (C - D) = (1 - cc)(V2 - rho D) (:65)     and   -C = (cc/rho)(V2 - rho D) = (1 - cc)(V2 - rho D)
// because cc/rho = 1/(rho+1) = 1 - cc: after every iteration C == -V2 exactly (both start at 0).
// EPI_UPDATE_C (after Xs = A S B, :58):  V2 <- (1 - cc)(V2 - rho (X - Xs)).   e_r0 = X, e_rw0 = V2; Xs -> d.C
// EPI_UPDATE_X (after Y = Z - Q Z, :35), with C = -V2:
//   X = (V1 + rho Y + subY + (1 - rho) V2 + rho Xs) .* invD (:38-40);  K = X + (1 - 1/rho) V2 (:43);
//   V1 += rho (Y - X) (:64);  Znext = X - V1/rho (the next iteration's svt argument, :35).
//   e_rw0 = V1, e_w1 = X, e_w2 = K, e_r0 = V2, e_r2 = Xs, e_r3 = subY, e_f0 = invD, e_w3 = Znext (may be NULL);
//   Y -> d.C if epi_store_c
// EPI_SADMM (sparse_admm.m:21-30 applied to the accumulator tile; sa_mode selects the step):
//   1: C = acc ./ (sa_lr[i] sa_lt[j] - sa_rho)                            (the diagonal solve between the two transforms, :26)
//   2: acc = R (not stored);  Z += rho (R - S) (:30);  S' = soft(R + Z/rho, sa_thr) (:21-22 of the NEXT iteration);
//      RHS = Z - rho S' + A'vec(OH) (:26).   e_rw0 = Z, e_r0 = S, e_w1 = S', e_r2 = A'vec(OH), e_w2 = RHS
enum { EPI_NONE = 0, EPI_UPDATE_C = 1, EPI_UPDATE_X = 2, EPI_SADMM = 3 };
enum { GEMM_MISC = 0, GEMM_CORRELATE = 1, GEMM_SYNTH = 2, GEMM_GRAM = 3 };
int launch_cgemm(jstsp_ctx *ctx, const GemmDesc &d, int tag = GEMM_MISC);

// ---- split-f16 complex GEMM (hgemm.hip): fp32-equivalent accuracy on the f16 matrix pipe ----
// C[t] (m x n) = a[t] (m x k, fp32, a(i,kk) = A[t*sAt + i + kk*sAk]) * b[t] (k x n, packed once by hgemm_pack)
struct HPack {
    uint4 *data = nullptr;      // [count][JT][KS][4 planes][64 lanes] 16-byte fragments
    uint32_t *bmax = nullptr;   // [count] float bits of max(|re|, |im|)
    int KS = 0, JT = 0, count = 0;
    long long st = 0;           // uint4 per problem
};
struct HGemmDesc {
    const float2 *A; long long sAt, sAk;
    const uint32_t *amax;       // [batch] float bits of max(|re|, |im|) of a[t]
    const uint4 *Bp; long long sPt;
    const uint32_t *bmax; int sbmax;
    int KS, JT;
    float2 *C; long long sCt; int ldc;
    int m, n, k, batch;
    int epi;                    // EPI_NONE or EPI_UPDATE_C (fields as in GemmDesc)
    const TrialParams *prm;
    const float2 *e_r0;
    float2 *e_rw0;
    uint32_t *amax_v2;          // EPI_UPDATE_C: optional [batch] atomicMax of max(|re|,|im|) of the new V2
    // optional: the a operand packed like b (hgemm_pack with j = i): then A is not read, amax = the pack's bmax
    const uint4 *Ap; long long sApt; int aKS;
    // the product is Hermitian (a Gram: b = conj(a)^T): tiles entirely below the diagonal are not computed - the caller fills the
    // lower triangle from the upper one (hermitian_fill_lower)
    int herm_upper;
    // block -> (trial, tile) map, set by launch_hgemm.  0: all tiles of a trial on one XCD (the a panel of the trial stays in that
    // XCD's L2; right when every trial has its own b).  map_tb > 0 (b shared by the trials, sPt == 0): the workgroups resident on
    // one XCD at a time are map_tb trials x map_tt tiles, so a b panel fetched for one trial is an L2 hit for the other map_tb - 1
    int map_tb, map_tt;
};
// G[r, c] = conj(G[c, r]) for r > c, count matrices of order n (column-major, leading dimension n, stride sGt)
int hermitian_fill_lower(jstsp_ctx *ctx, float2 *G, long long sGt, int n, int count);
// Gram partials of a rows x cols matrix, rows <= 64 (same layout as the GEMM_GRAM split-K output):
// Gpart[(t*nsplit + s)*rows*rows + i + rows*j];  amax[t] bounds max(|re|,|im|) of Z[t]
// skip_prm != nullptr: problems with prm[t].tauY_rho <= 2^-27 amax[t] are skipped (see jacobi2_kernel)
int launch_hgram(jstsp_ctx *ctx, const float2 *Z, long long sZt, int rows, int cols, int count, int nsplit,
                 const uint32_t *amax, float2 *Gpart, const TrialParams *skip_prm = nullptr, const float2 *Z2 = nullptr,
                 const TrialParams *zprm = nullptr,      // Z2: Gram of Z - zprm[t].irho * Z2 (same layout as Z)
                 bool norm_only = false);                // the Gram's only use is its lambda_max in convergence_error: high f16 plane only
// G_x = X X^H, G_v = V1 V1^H, G_z = (X - V1/rho)(X - V1/rho)^H in one pass over X and V1 (rows <= 64)
int launch_hgram3(jstsp_ctx *ctx, const float2 *X, const float2 *V1, long long sZt, int rows, int cols, int count, int nsplit,
                  const uint32_t *xmax, const uint32_t *vmax, const uint32_t *zmax, const TrialParams *prm, float2 *Gz,
                  float2 *Gx, float2 *Gv);
size_t hgemm_pack_bytes(int Kd, int J, int count);
// true when launch_hgemm gives a product with ONE packed b operand for the batch to the two-trials-per-workgroup kernel (hgemm.hip)
bool hgemm_pair_shape(int m, int n, int batch);
// amax[t] = max(|re|, |im|) over n contiguous elements of X[t*sXt ...]
int hgemm_absmax(jstsp_ctx *ctx, const float2 *X, long long n, long long sXt, int count, uint32_t *amax);
// b(kk, j) = B[t*sBt + kk*sBk + j*sBj] (conjugated if conj), kk < Kd, j < J; each B[t] spans n_contig contiguous elements
int hgemm_pack(jstsp_ctx *ctx, HPack &p, Arena &ar, const float2 *B, long long sBt, long long sBk, long long sBj,
               int conj, int Kd, int J, int count, long long n_contig, const uint32_t *bmax_known = nullptr);
int hgemm_repack(jstsp_ctx *ctx, const HPack &p, const float2 *B, long long sBt, long long sBk, long long sBj, int conj,
                 int Kd, int J, const uint32_t *bmax);
int launch_hgemm(jstsp_ctx *ctx, const HGemmDesc &d, const char *prof_name = nullptr);
bool use_hgemm(long long m, long long n, long long k);   // policy (env JSTSP_H2), read at every call

// ---- batched Hermitian eigen-solver (eig.hip) -------------------------------------------
// G[t] = sum_{s<nsplit} Gpart[t*sGt + s*sGs + i + n*j]   (n x n Hermitian PSD Gram, n <= 128)
// mode EIG_SVT_Q : Q[t] = U diag(min(1, tau_t/sigma_i)) U^H with sigma = sqrt(lambda)
//                  (so that svt(Z, tau) = Z - Q Z);  tau_t = prm[t].tauY_rho or tau[t]
// mode EIG_LMAX  : lam_out[t] = largest eigenvalue
// mode EIG_VECS  : Q[t] = eigenvectors (columns), lam_out[t*n + i] = eigenvalues (unsorted)
enum { EIG_SVT_Q = 0, EIG_LMAX = 1, EIG_VECS = 2 };
int launch_eig(jstsp_ctx *ctx, int mode, int n, int batch, const float2 *Gpart, long long sGt,
               int nsplit, long long sGs, const TrialParams *prm, const float *tau,
               float2 *Q, float *lam_out, float2 *Vg);
bool eig_needs_global_v(int n);
// Orders above 128 (eig_large.hip): two-sided block Jacobi on 128 x 128 sub-problems (this kernel) + batched GEMMs.
int launch_eig_large(jstsp_ctx *ctx, int mode, int n, int batch, const float2 *Gpart, long long sGt, int nsplit,
                     long long sGs, const TrialParams *prm, const float *tau, float2 *Q, float *lam_out);
// Fast paths (eig2.hip): warm-started block Jacobi for n <= 64; tridiagonalisation + Sturm for lambda_max.
int launch_eig_fast(jstsp_ctx *ctx, int mode, int n, int batch, const float2 *Gpart, long long sGt, int nsplit,
                    long long sGs, const TrialParams *prm, const float *tau, float2 *Q, float *lam_out,
                    float2 *Uwarm, int warm, const uint32_t *skip_amax = nullptr);
// Orders 65..128 (eig3.hip): G in LDS, eigenvector basis in registers as a systolic array; SVT projector only.
int launch_eig128(jstsp_ctx *ctx, int n, int batch, const float2 *Gpart, long long sGt, int nsplit, long long sGs,
                  const TrialParams *prm, const float *tau, float2 *Q, float2 *Uwarm = nullptr, int warm = 0,
                  int max_sweeps = 16, float stop_level = 0.f);     // stop_level: see GramWS::eig_stop
int eig_fast_ne(int n);           // padded order (32 or 64) of the warm-start basis
// Warm-start record of the Lanczos lambda_max kernel (eig2.hip): per matrix the Ritz vector of the previous call.
struct LanczosWarm {
    float2 *x = nullptr;            // [count][ne] Ritz vectors
    int *state = nullptr;           // [count] 0: no vector yet, 1: x valid
    unsigned *mismatch = nullptr;   // [1] periodic cold verifications that disagreed with the warm-started value
    int ne = 0;                     // lanczos_ne(n)
    int call = 0;                   // call counter of the owning loop (every lanczos_verify-th call is verified)
};
int lanczos_ne(int n);            // padded order (64 or 128) of the warm-start vectors
int launch_lmax(jstsp_ctx *ctx, int n, int batch, const float2 *Gpart, long long sGt, int nsplit, long long sGs,
                float *lam_out, bool lanczos = false, const LanczosWarm *lw = nullptr, int first = 0);

// ---- fused element-wise / reduction kernels (admm.hip) -----------------------------------
int launch_form_z(jstsp_ctx *ctx, long long nm, int batch, const float2 *X, const float2 *V1,
                  const TrialParams *prm, float2 *Z);
int launch_update_x(jstsp_ctx *ctx, long long nm, int batch, float2 *X, float2 *V1,
                    const float2 *V2, const float2 *C, const float2 *Xs, const float2 *Y,
                    const float2 *subY, const float *invD, const TrialParams *prm, float2 *K);
int launch_update_c(jstsp_ctx *ctx, long long nm, int batch, const float2 *X, const float2 *Xs,
                    float2 *V2, float2 *C, const TrialParams *prm);
int launch_step_v(jstsp_ctx *ctx, int g, int batch, const float2 *Res, const float2 *RRes,
                  float2 *V, float2 *S, const int32_t *rank, int cnt, const TrialParams *prm,
                  double *ce3, int Imax, int it, float2 *RV = nullptr, int waves8 = 0,      // RV != nullptr: RV += alpha * RRes as well
                  float2 *Vlo = nullptr, float2 *RVlo = nullptr, int vlo_reset = 0);         // low-order parts: v and R v as two floats (admm.hip)
int launch_soft(jstsp_ctx *ctx, int g, int batch, const float2 *V, float2 *S, const int32_t *rank,
                int cnt, const TrialParams *prm);
int launch_inv_d(jstsp_ctx *ctx, long long nm, int batch, const float *Omega, float scale2rho,
                 const TrialParams *prm, float *invD);
int launch_rank_from_index(jstsp_ctx *ctx, int g, int batch, const int32_t *indx, int32_t *rank);
int launch_eye_minus(jstsp_ctx *ctx, int n, int count, const float2 *Q, float2 *P);
int launch_ce_ratio(jstsp_ctx *ctx, int batch, const float *lamV1, const float *lamV2,
                    const float *lamX, double *ce, int Imax, int it);

// ---- profiling helpers ---------------------------------------------------------------------
void prof_begin(jstsp_ctx *ctx, const char *name);
void prof_end(jstsp_ctx *ctx, const char *name);

}  // namespace jstsp
