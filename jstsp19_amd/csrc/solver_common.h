// Host-side helpers shared by the solver drivers: GEMM call shorthands, batched SVT,
// spectral norms, host<->device staging.
#pragma once
#include "common.h"
#include <vector>

namespace jstsp {

// Column-major matrix view of a batched operand: element (r, c) of problem t at
// p[t*st + r + ld*c]; st == 0 => one matrix shared by the whole batch.
struct Mat {
    const float2 *p; long long st; int ld;
};

// C = alpha * op(A) * op(B) + beta * D, all column-major; opX in {'N','C'} ('C' = conj. transpose)
int gemm(jstsp_ctx *ctx, char opA, char opB, int m, int n, int k, int batch, Mat A, Mat B, float2 *C,
         long long sCt, int ldc, float alpha = 1.f, const float2 *D = nullptr, long long sDt = 0,
         int ldd = 0, float beta = 0.f, int tag = GEMM_MISC, int splitk = 1, long long sCsplit = 0);

// Gram of a dictionary factor with float64 products and sums (gram64.hip): side 'L' G = X^H X (cols x cols), 'R' G = X X^H
// (rows x rows); X rows x cols column-major (ld = rows), sXt = 0: shared; G order n, ld = n, exactly Hermitian; Glo (optional,
// same layout): the part of the float64 sum that the fp32 G does not hold
int gram_f64(jstsp_ctx *ctx, char side, const float2 *X, long long sXt, int rows, int cols, int count, float2 *G, long long sGt,
             float2 *Glo = nullptr);

// The 64-term products of the gradient step on the f16 matrix pipe (hsmall.hip), N = Gr = 64:
//   Tc = sum of `parts` partial sums (P[t sPt + p sPp + n + 64 g]) [+ leading-column terms of a block-Toeplitz dictionary: Kf, Bdl],
//   Res = A^H Tc - RV (RV may be NULL), P1 = G_A Res, pmax[t] = max|P1| (atomicMax; may be NULL); Tc stored if non-NULL
bool grad_head_shape_ok(int N, int Gr, int G2);
int launch_grad_head(jstsp_ctx *ctx, int G2, int batch, const float2 *P, long long sPt, long long sPp, int parts, const float2 *Kf,
                     const float2 *Bdl, long long sBdl, const float2 *A, long long sA, const float2 *GA, long long sGA,
                     const float2 *RV, float2 *Tc, float2 *Res, float2 *P1, uint32_t *pmax, const float2 *RVlo = nullptr);
//   P1 = (G_hi + G_lo) X, G Hermitian 64 x 64 in two floats (sG = 0: shared), X and P1 64 x G2 per trial
int launch_left2(jstsp_ctx *ctx, int G2, int batch, const float2 *Ghi, const float2 *Glo, long long sG, const float2 *X, float2 *P1,
                 uint32_t *pmax);

// Descriptor only (the caller may attach a fused epilogue before launch_cgemm).
GemmDesc make_gemm(char opA, char opB, int m, int n, int k, int batch, Mat A, Mat B, float2 *C, long long sCt,
                   int ldc, float alpha = 1.f, const float2 *D = nullptr, long long sDt = 0, int ldd = 0,
                   float beta = 0.f, int splitk = 1, long long sCsplit = 0);

// Workspace of the Gram-form SVT / spectral norm of rows x cols matrices.
struct GramWS {
    int rows = 0, cols = 0, n = 0, nsplit = 1, batch = 0;
    bool left = true;          // true: G = Z Z^H (rows <= cols); false: G = Z^H Z
    float2 *Gpart = nullptr;   // batch * nsplit * n*n
    float2 *Q = nullptr;       // batch * n*n
    float2 *Vg = nullptr;      // eigenvectors in HBM when they do not fit in LDS
    float2 *Uwarm = nullptr;   // eigenvector basis of the previous call (warm start): NE x NE padded for n <= 64, n x n above
    float2 *Twarm = nullptr;   // n > 64: temporary of the warm-start transform G <- Uw^H (G Uw)
    mutable int warm = 0;      // 1 once Uwarm holds a basis
    // Orders 65..128, warm-started sequences only: no (further) Jacobi sweep is run once the Gram in the previous basis has all
    // relative off-diagonals |g_pq| / sqrt(g_pp g_qq) below this level (0: the end-of-sweep rule alone, which confirms convergence
    // with a sweep that starts below 1e-4).  An inexact inner solve for the fixed-point loops that can take it (mc_svt, mc_admm).
    float eig_stop = 0.f;
    mutable LanczosWarm lz;    // norm workspaces (need_q == false, n <= 128): warm-start record of the Lanczos lambda_max kernel
    static size_t bytes(int rows, int cols, int batch, bool need_q, int force_nsplit = 0);
    int alloc(Arena &a, int rows, int cols, int batch, bool need_q, int force_nsplit = 0);
};

// G partials of Z (split-K over the long dimension).
int gram_partials(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, long long sZt);
// Y = svt(Z, tau_t): tau from prm[t].tauY_rho, or tau[t] when tau != nullptr.  Y may alias nothing.
// `sequence` = true: successive calls see slowly varying inputs (an ADMM loop), so the
// eigenvector basis of the previous call warm-starts the Jacobi sweeps.
int svt_batched(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, const TrialParams *prm,
                const float *tau, float2 *Y, bool sequence = false);
// The two halves of svt_batched: Gram + eigen-decomposition -> projector Q; then Y = Z - Q Z.
// amax != nullptr (per-problem bound on max(|re|,|im|) of Z): Gram on the split-f16 path when rows <= 64
int svt_prepare(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, const TrialParams *prm, const float *tau,
                bool sequence, const uint32_t *amax = nullptr, bool allow_skip = false, const float2 *Z2 = nullptr,
                bool gram_done = false);   // gram_done: the partials of G are already in the workspace
                // Z2 (split-f16 Gram path only): the svt argument is Z - prm[t].irho * Z2, never stored
int svt_apply(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, float2 *Y);
// Gram partials of problems [t0, t0 + count) only (same workspace layout as gram_partials).
int gram_partials_range(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, long long sZt, int t0, int count,
                        const uint32_t *amax = nullptr, const TrialParams *skip_prm = nullptr, const float2 *Z2 = nullptr,
                        const TrialParams *zprm = nullptr, bool norm_only = false);
// lam[t] = lambda_max of the Gram partials already in the workspace, all w.batch problems
// The warm-start record of a norm workspace: all vectors invalid, the context's mismatch counter zeroed (on ctx->stream;
// once per solve, before the first lambda_max of its loop).  `call` of the record is set by the loop's owner.
int lanczos_warm_reset(jstsp_ctx *ctx, const GramWS &w);
int lmax_from_partials(jstsp_ctx *ctx, const GramWS &w, float *lam, bool lanczos = false);
int lmax_from_partials_range(jstsp_ctx *ctx, const GramWS &w, int first, int count, float *lam, bool lanczos);   // matrices [first, first + count)
// Make sure the context's side streams / events exist.
int ensure_side_streams(jstsp_ctx *ctx);
int ensure_bj_resources(jstsp_ctx *ctx);    // ctx->bj_stream / bj_ev (common.h)
bool ensure_cu_streams(jstsp_ctx *ctx);     // false: no masked streams on this runtime (callers use the side streams)
// Temporarily route the launch helpers (which use ctx->stream) to another stream.
struct StreamScope {
    jstsp_ctx *c; hipStream_t saved;
    StreamScope(jstsp_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { c->stream = s; }
    ~StreamScope() { c->stream = saved; }
};
// lam[t] = sigma_max(Z_t)^2   (lanczos: the per-iteration convergence-error curves; Householder + Sturm otherwise)
int sigma_max_sq(jstsp_ctx *ctx, const GramWS &w, const float2 *Z, float *lam, bool lanczos = false);

// Ginv[t] = G[t]^-1 for `count` Hermitian PD n x n matrices (hinv.hip); workspace from ctx->arena.
size_t hinv_bytes(int n, int count);
int hermitian_inverse(jstsp_ctx *ctx, int n, int count, const float2 *G, float2 *Ginv);

// pinv of small matrices in float64 (pinv.hip): P[t] (cols x rows) = pinv(A[t]) (rows x cols) when the matrix fits in LDS
bool pinv_fits(int rows, int cols);
int launch_pinv(jstsp_ctx *ctx, int rows, int cols, int count, const float2 *A, long long sAt, int lda, float2 *P,
                long long sPt, int ldp, float *rcond_out = nullptr);
// The context's conditioning record (common.h: jstsp_ctx::diag): allocate on first use / reset at the start of a call.
int ensure_diag(jstsp_ctx *ctx);
int diag_reset(jstsp_ctx *ctx);
// After the stream has been synchronised: JSTSP_E_ILLCOND when a Gram inverse of the call lost all its fp32 digits.
int diag_check_host(jstsp_ctx *ctx, const char *what);

// A proposed_algorithm solve whose device work has been enqueued but whose results are still in the context's workspace
// (proposed.hip): the halves of a pipelined JSTSP_HOST call.  Valid until the next solve on that context.
struct PendingSolve {
    bool active = false, fused = false, want_ce = false;
    int batch = 0;
    size_t g = 0, nm = 0;
    int Imax = 0;
    const float2 *dS = nullptr, *dY = nullptr;
    const double *dce = nullptr;
    const uint32_t *ovf = nullptr;
};
// The two phases for a caller that stages its own inputs (c64.hip): every array argument in device memory (tau_Y, tau_S, rho on
// the host as always).  enqueue: the whole solve without its final copies and flag read; flags: the trials whose predicted k
// scale overflowed in the fused pass (one stream synchronisation); resolve: such a run of trials again by the three-kernel
// iteration, outputs to device arrays.
int proposed_enqueue_device(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *subY, const float *Omega,
                            const jstsp_c32 *A, long long strideA, const jstsp_c32 *B, long long strideB, int Imax,
                            const double *tau_Y, const double *tau_S, const double *rho, int type, const int32_t *indx_S,
                            bool want_ce, PendingSolve *out);
int proposed_pending_flags(jstsp_ctx *ctx, const PendingSolve &p, std::vector<int> *ovf);
int proposed_resolve_device(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *subY, const float *Omega,
                            const jstsp_c32 *A, long long strideA, const jstsp_c32 *B, long long strideB, int Imax,
                            const double *tau_Y, const double *tau_S, const double *rho, int type, const int32_t *indx_S,
                            jstsp_c32 *S_dev, jstsp_c32 *Y_dev, double *ce_dev);

// Host -> device scalar block.
int upload(jstsp_ctx *ctx, void *dst, const void *src, size_t bytes);

// Staging of an array argument according to memspace: returns a device pointer (the caller's
// pointer for JSTSP_DEVICE, an arena copy for JSTSP_HOST).
template <class T> int stage_in(jstsp_ctx *ctx, const T *src, size_t n, int memspace, const T **out)
{
    if (memspace == JSTSP_DEVICE) { *out = src; return 0; }
    T *d = ctx->arena.get<T>(n);
    JSTSP_REQUIRE(d, JSTSP_E_NOMEM, "workspace exhausted while staging an input");
    JSTSP_HIP(hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *out = d;
    return 0;
}
template <class T> int stage_out(jstsp_ctx *ctx, T *dst, const T *dev, size_t n, int memspace)
{
    if (!dst) return 0;
    JSTSP_HIP(hipMemcpyAsync(dst, dev, n * sizeof(T),
                             memspace == JSTSP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             ctx->stream));
    return 0;
}
inline size_t rnd256(size_t b) { return (b + 255) & ~size_t(255); }

// Host-side staging of a block-Toeplitz dictionary (hostpack.hip): T = float2 or double2 (narrowed while copying).
// *gt_out = the block height when the compact route was taken (Bdev then receives the expanded array on ctx->stream), 0 otherwise
// (nothing uploaded).  Cdev: device room for the compact form, host_toeplitz_compact_elems() float2.
template <class T>
int host_toeplitz_stage(jstsp_ctx *ctx, const T *Bh, int G2, int M, int nB, float2 *Bdev, float2 *Cdev, size_t cdev_elems, int *gt_out);
size_t host_toeplitz_compact_elems(int G2, int M, int nB);

// ---- one pass over the dictionary per iteration (fused.hip) ---------------------------------
struct FusedWS {
    uint4 *Bf = nullptr; long long sBf = 0;       // tile images of B, uint4 per problem
    uint4 *ASp = nullptr; long long sAS = 0;      // B-operand fragments of (A S)^T, re-packed every iteration
    float2 *Ppart = nullptr;                      // [batch][parts][N x G2] partial sums of K B^H
    uint32_t *ovf = nullptr;                      // [batch]: raised for a trial when a k entry left the f16 range of its scale
    uint4 *Wqp = nullptr;                         // B-operand fragments of (I - Q)^T, 2048 uint4 per problem
    int parts = 0;
    // block-Toeplitz dictionary (fused.hip, "compact image"): B(ld Gt + g, m) == B(g, m - ld) bit for bit for m >= ld.
    // Then only block 0 (+ the ld leading columns of block ld) is stored, and the pass's refill reads it at shifted columns.
    uint4 *Ec = nullptr; long long sEc = 0;       // compact image, uint4 per problem; Bf stays NULL then
    int gt = 0, ecols = 0, ehalo = 0;             // block height (0: unstructured), chunks per image row, halo chunks before column 0
    // v2 (block height 64, fused_pass64_kernel): the LDS tile is the 39-column window of block 0 itself; the leading columns
    // m < ld of block ld (outside the Toeplitz part) are applied as fp32 corrections around the pass
    int v2 = 0;
    const float2 *Bsrc = nullptr; long long sBsrc = 0;    // the caller's dictionary
    float2 *Bdl = nullptr; long long sBdl = 0;            // its leading columns, row-major: [nB][G2][8]
    float2 *XsD = nullptr, *Kf = nullptr;                 // [batch][4][64 x 8]: (A S) Delta of this iteration in four partial sums; [batch][64 x 8]: k(:, 0..6) of the pass
};
struct FusedDesc {
    const uint4 *Bf; long long sBf;               // 0: one dictionary for the batch
    const uint32_t *bmax; int sbmax;
    const uint4 *ASp; long long sAS;
    const uint32_t *wmax;                         // [batch] max|A S|
    const uint32_t *kmax_prev;                    // [batch] max|k| of the previous iteration
    float2 *X, *V1, *V2;                          // N x M state, updated in place
    const float2 *subY, *Y; const float *invD;
    long long snm;
    const TrialParams *prm;
    float2 *Ppart;
    uint32_t *kmax_out, *xmax, *v1max, *zmax, *v2max, *ovf;
    int M, G2, batch, parts;
    // Y formed in the pass (instead of read from Y): fragments of I - Q, the svt argument Z of the next iteration and its
    // maximum, where to put the Z after that; Yout != NULL: also store Y (the last pass: Y is an output of the solver)
    const uint4 *Wqp; const float2 *Zin; const uint32_t *zmax_in; float2 *Zout, *Yout;
    int kback;                                    // headroom of the predicted k scale in bits (KBACK; JSTSP_FUSED_KBACK: tests)
    const uint4 *Ec; long long sEc; int gsh, ecols, ehalo;   // compact image of a block-Toeplitz dictionary (gsh = log2 Gt; 0: none)
    int v2; const float2 *XsD; float2 *Kf;                   // fused_pass64_kernel (see FusedWS)
    int inv_is_omega;                             // invD points at the caller's Omega (16-byte aligned): 1 / (Omega + 2 rho) is formed in the
                                                  // pass as two floats (common.h: admm_invd) instead of read as one rounded float
};
bool fused_shape_ok(int N, int M, int G2, int parts);
size_t fused_bytes(int M, int G2, int nB, int batch, int parts);
int fused_probe_toeplitz(jstsp_ctx *ctx, Arena &ar, const float2 *B, long long sBt, int G2, int M, int nB, int *gt);   // syncs
// G_B = B B^H (G2 x G2, column-major) of a block-Toeplitz dictionary from its first block row G0 = B(0:gt, :) B^H (gt x G2)
int toeplitz_gram_assemble(jstsp_ctx *ctx, const float2 *B, long long sBt, int G2, int M, int gt, int nB, const float2 *G0,
                           const float2 *G0lo, float2 *G, float2 *Glo);     // G0lo, Glo optional: low-order parts (float64 = hi + lo)
int fused_alloc(Arena &ar, FusedWS &f, int M, int G2, int nB, int batch, int parts, int gt, int v2);
int fused_pack_b(jstsp_ctx *ctx, FusedWS &f, const float2 *B, long long sBt, int G2, int M, int nB, const uint32_t *bmax);
int fused_pack_as(jstsp_ctx *ctx, const FusedWS &f, const float2 *W, long long sWt, int G2, int M, int batch, const uint32_t *wmax);
int launch_fused_pass(jstsp_ctx *ctx, const FusedDesc &d);
int fused_reduce(jstsp_ctx *ctx, const FusedWS &f, int G2, int M, int batch, float2 *Tc);
int fused_pack_wq(jstsp_ctx *ctx, const FusedWS &f, const float2 *Q, int batch);      // from the raw Q of svt_prepare

}  // namespace jstsp
