// jstsp_proposed_algorithm_c32 — the batched ADMM of
//   basic_system_functions/proposed_algorithm.m:1-73          (indx_S == NULL)
//   basic_system_functions/proposed_algorithm_angles.m:1-85   (indx_S != NULL)
// in structured form: the dense Kronecker operators K1, K2, R, K3 of the
// reference (:14-25, angles :37-43) are never formed —
//   K2*s = vec(A S B),  K2'*k = vec(A^H K B^H),  R*v = vec((A^H A) V (B B^H)),
//   iK1*b = b ./ (Omega + 2 rho),  K3*s = Omega_S .* S.
// All problems of the batch advance together; every step below is one kernel launch over
// the whole batch on the context's stream.
#include "solver_common.h"
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <algorithm>

// JSTSP_HOST_TRACE=1: wall-clock marks of the JSTSP_HOST path on stderr (ms since the first mark of the process)
static void host_trace(const char *what, int k = -1)
{
    static const bool on = [] { const char *e = jstsp::xp_getenv("JSTSP_HOST_TRACE"); return e && atoi(e) != 0; }();
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[jstsp host] %9.2f ms  %s%s\n", ms, what, k == 0 ? " (half 0)" : (k == 1 ? " (half 1)" : ""));
}

namespace jstsp {

struct ProposedWS {
    // state, N x M per problem
    float2 *X, *V1, *V2, *C, *Xs, *Y, *ZK, *Zb, *Zb2;
    float *invD;
    // Gr x G2 per problem
    float2 *V, *RV, *Res, *RRes, *S, *P1;
    float2 *Vlo = nullptr, *RVlo = nullptr;     // low-order parts of v and R v (compensated accumulation: allocated by the solve when used)
    // N x G2 per problem
    float2 *Tc, *W;
    float2 *GA, *GB;
    TrialParams *prm;
    int32_t *rank;
    double *ce;
    float *lam;            // 3 * batch
    GramWS gz, gn;         // SVT of Z; spectral norms of V1, V2, X
    // split-f16 path (hgemm.hip): B packed once per solve in both orientations, per-problem operand maxima
    bool h2 = false;
    HPack Bc, Bs;          // b(k = m, j = g) = conj(B) for K B^H ;  b(k = g, j = m) = B for (A S) B
    uint32_t *kmax = nullptr, *wmax = nullptr;
    uint32_t *nmax = nullptr, *zmax = nullptr;   // [X | V1 | V2] (3*batch, contiguous after kmax) and Znext maxima
    bool h2g = false;      // the two (G_A V) G_B applies of the gradient step on the same path
    HPack GBp;             // b(k, j) = G_B[k + G2 j]
    HPack Wp;              // the synthesis' a operand A S, re-packed every iteration (64 j-tiles would each split it)
    HPack Kp;              // K of K B^H in fragment order (one dictionary for the batch: hgemm_pair_kernel reads it for two trials at a time)
    uint32_t *pmax = nullptr;
};

// k-split of the fused three-Gram pass (hgram3_kernel: G_x, G_v1 and G_z from one read of X and V1): chunks of at most
// 1024 columns; 0 = that pass is not used for this shape (then every Gram workspace picks its own split)
static int gram3_nsplit(int N, int M, int G2, bool want_ce)
{
    // (measured at configs[1], chunks of 512 / 1024 / 2048 columns: 493 / 497 / 496 channel-estimates/s — more splits
    //  shorten the single-level fp32 chains but every consumer of the Gram sums the partials again)
    const int chunk = 1024;
    if (!(want_ce && N <= 64 && N <= M && use_hgemm(N, G2, M))) return 0;
    return std::max(1, std::min(32, (M + chunk - 1) / chunk));
}

static size_t proposed_bytes(int N, int M, int Gr, int G2, int batch, int nA, int nB, bool angles,
                             bool want_ce, int Imax)
{
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2, ng = (size_t)N * G2;
    size_t b = 0;
    b += rnd256(3 * batch * nm * sizeof(float2)) + 6 * rnd256(batch * nm * sizeof(float2));
    b += rnd256(batch * nm * sizeof(float));
    b += 6 * rnd256(batch * g * sizeof(float2));
    b += 2 * rnd256(batch * ng * sizeof(float2));
    b += rnd256((size_t)nA * Gr * Gr * sizeof(float2)) + rnd256((size_t)nB * G2 * G2 * sizeof(float2));
    b += rnd256(batch * sizeof(TrialParams));
    if (angles) b += rnd256(batch * g * sizeof(int32_t));
    b += rnd256((size_t)batch * 3 * Imax * sizeof(double));
    b += rnd256(3 * (size_t)batch * sizeof(float));
    const int ns3 = gram3_nsplit(N, M, G2, want_ce);
    b += GramWS::bytes(N, M, batch, true, ns3);
    if (want_ce) b += GramWS::bytes(N, M, 3 * batch, false, ns3);
    if (use_hgemm(N, G2, M))
        b += hgemm_pack_bytes(M, G2, nB) + hgemm_pack_bytes(G2, M, nB) + 2 * rnd256(8 * batch * sizeof(uint32_t)) +
             hgemm_pack_bytes(G2, N, batch);
    if (use_hgemm(N, G2, M) && nB == 1 && hgemm_pair_shape(N, G2, batch))
        b += rnd256((size_t)batch * 2 * (4 * ((M + 63) / 64)) * 256 * sizeof(uint4));
    if (use_hgemm(Gr, G2, G2)) b += hgemm_pack_bytes(G2, G2, nB) + rnd256(2 * batch * sizeof(uint32_t));
    return b;
}

static int proposed_alloc(Arena &a, ProposedWS &w, int N, int M, int Gr, int G2, int batch, int nA, int nB,
                          bool angles, bool want_ce, int Imax)
{
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2, ng = (size_t)N * G2;
    // X, V1, V2 in one block: the spectral norms of the three are one batched Gram over 3*batch matrices
    w.X = a.get<float2>(3 * batch * nm);
    w.V1 = w.X ? w.X + batch * nm : nullptr;
    w.V2 = w.X ? w.X + 2 * batch * nm : nullptr;
    w.C = a.get<float2>(batch * nm); w.Xs = a.get<float2>(batch * nm); w.Y = a.get<float2>(batch * nm);
    w.ZK = a.get<float2>(batch * nm);
    w.Zb = a.get<float2>(batch * nm);
    w.Zb2 = a.get<float2>(batch * nm);
    w.invD = a.get<float>(batch * nm);
    w.V = a.get<float2>(batch * g); w.RV = a.get<float2>(batch * g); w.Res = a.get<float2>(batch * g);
    w.RRes = a.get<float2>(batch * g); w.S = a.get<float2>(batch * g); w.P1 = a.get<float2>(batch * g);
    w.Tc = a.get<float2>(batch * ng); w.W = a.get<float2>(batch * ng);
    w.GA = a.get<float2>((size_t)nA * Gr * Gr); w.GB = a.get<float2>((size_t)nB * G2 * G2);
    w.prm = a.get<TrialParams>(batch);
    w.rank = angles ? a.get<int32_t>(batch * g) : nullptr;
    w.ce = a.get<double>((size_t)batch * 3 * Imax);
    w.lam = a.get<float>(3 * (size_t)batch);
    JSTSP_REQUIRE(w.X && w.V1 && w.V2 && w.C && w.Xs && w.Y && w.ZK && w.Zb && w.Zb2 && w.invD && w.V && w.RV && w.Res &&
                      w.RRes && w.S && w.P1 && w.Tc && w.W && w.GA && w.GB && w.prm && w.ce && w.lam &&
                      (!angles || w.rank),
                  JSTSP_E_NOMEM, "proposed_algorithm: workspace exhausted");
    const int ns3 = gram3_nsplit(N, M, G2, want_ce);
    JSTSP_TRY(w.gz.alloc(a, N, M, batch, true, ns3));
    if (want_ce) JSTSP_TRY(w.gn.alloc(a, N, M, 3 * batch, false, ns3));
    w.h2 = use_hgemm(N, G2, M);
    if (w.h2) {
        // operand maxima of one iteration, one block zeroed once per iteration: kmax | X | V1 | V2 | Znext | wmax | pmax x2.
        // TWO blocks, used by even / odd iterations: with JSTSP_OVERLAP=1 the side-stream Grams of iteration i still
        // read their maxima while the main stream starts iteration i+1, whose memset must not touch those words.
        w.kmax = a.get<uint32_t>(16 * (size_t)batch);
        JSTSP_REQUIRE(w.kmax, JSTSP_E_NOMEM, "proposed_algorithm: workspace exhausted");
        w.nmax = w.kmax + batch; w.zmax = w.kmax + 4 * (size_t)batch; w.wmax = w.kmax + 5 * (size_t)batch;
        w.pmax = w.kmax + 6 * (size_t)batch;
        w.Wp.KS = 4 * ((G2 + 63) / 64); w.Wp.JT = 2 * ((N + 63) / 64); w.Wp.count = batch;     // (an a pack needs whole 64-row
                                                                                            //  tiles only, not 128-wide ones)
        w.Wp.st = (long long)w.Wp.JT * w.Wp.KS * 256;
        w.Wp.data = a.get<uint4>((size_t)batch * w.Wp.st);
        w.Wp.bmax = w.wmax;
        JSTSP_REQUIRE(w.Wp.data, JSTSP_E_NOMEM, "proposed_algorithm: workspace exhausted");
        if (nB == 1 && hgemm_pair_shape(N, G2, batch)) {
            w.Kp.KS = 4 * ((M + 63) / 64); w.Kp.JT = 2; w.Kp.count = batch;
            w.Kp.st = (long long)w.Kp.JT * w.Kp.KS * 256;
            w.Kp.data = a.get<uint4>((size_t)batch * w.Kp.st);
            JSTSP_REQUIRE(w.Kp.data, JSTSP_E_NOMEM, "proposed_algorithm: workspace exhausted");
        }
    }
    w.h2g = use_hgemm(Gr, G2, G2);
    if (w.h2g && !w.pmax) {
        w.pmax = a.get<uint32_t>(2 * (size_t)batch);
        JSTSP_REQUIRE(w.pmax, JSTSP_E_NOMEM, "proposed_algorithm: workspace exhausted");
    }
    return 0;
}

}  // namespace jstsp

using namespace jstsp;

// A JSTSP_HOST solve whose device work has been enqueued but whose results are still in the context's workspace: the halves of
// a pipelined host call (jstsp_proposed_algorithm_c32 below).  proposed_finish copies them out, reads the overflow flags and
// re-solves flagged trials.
static bool tune_host_pipeline()
{
    load_tuning();
    return tune().host_pipeline != 0;
}

// One batched solve.  allow_fused = false: never the fused pass (the re-solve of trials whose predicted k scale
// overflowed in it).  overflowed != NULL: receives the indices of such trials (their outputs are not to be used); reading
// the per-trial flags costs ONE stream synchronisation at the end of a solve that used the fused pass.
static int proposed_impl(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch,
                         const jstsp_c32 *subY_, const float *Omega_,
                         const jstsp_c32 *A_, long long strideA, const jstsp_c32 *B_,
                         long long strideB, int Imax, const double *tau_Y,
                         const double *tau_S, const double *rho, int type,
                         const int32_t *indx_S_, jstsp_c32 *S_out, jstsp_c32 *Y_out,
                         double *ce_out, int memspace, bool allow_fused, std::vector<int> *overflowed, PendingSolve *defer = nullptr)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(subY_ && Omega_ && A_ && B_ && tau_Y && tau_S && rho && S_out, JSTSP_E_NULL,
                  "proposed_algorithm: NULL array argument");
    JSTSP_REQUIRE(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0 && Imax >= 0, JSTSP_E_SHAPE,
                  "proposed_algorithm: bad shape N=%d M=%d Gr=%d G2=%d batch=%d Imax=%d", N, M, Gr, G2, batch,
                  Imax);
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d",
                  memspace);
    JSTSP_REQUIRE(type == JSTSP_TYPE_APPROXIMATE || type == JSTSP_TYPE_STD, JSTSP_E_ARG, "bad type %d", type);
    JSTSP_REQUIRE(std::min(N, M) <= 2048, JSTSP_E_UNSUPPORTED,
                  "proposed_algorithm: min(N, M) = %d > 2048 (order of the SVT's Gram eigenproblem)",
                  std::min(N, M));
    JSTSP_REQUIRE(strideA == 0 || strideA >= (long long)N * Gr, JSTSP_E_SHAPE, "strideA too small");
    JSTSP_REQUIRE(strideB == 0 || strideB >= (long long)G2 * M, JSTSP_E_SHAPE, "strideB too small");
    const bool approx = type == JSTSP_TYPE_APPROXIMATE;
    JSTSP_REQUIRE(approx || (N >= Gr && M >= G2), JSTSP_E_UNSUPPORTED,
                  "proposed_algorithm 'std': K2 = kron(B.', A) must have full column rank (N >= Gr, M >= G2); the "
                  "under-determined U\\(L\\k) of the reference returns a basic, not least-squares, solution");
    JSTSP_ENTER(ctx);

    const bool angles = indx_S_ != nullptr;
    const bool want_ce = ce_out != nullptr;
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2, ng = (size_t)N * G2;
    const int nA = strideA ? batch : 1, nB = strideB ? batch : 1;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t szB = strideB ? (size_t)strideB * (batch - 1) + (size_t)G2 * M : (size_t)G2 * M;

    size_t need = proposed_bytes(N, M, Gr, G2, batch, nA, nB, angles, want_ce, std::max(Imax, 1));
    if (!approx)
        need += (pinv_fits(N, Gr) ? rnd256((size_t)nA * Gr * N * sizeof(float2))
                                  : rnd256((size_t)nA * Gr * Gr * sizeof(float2)) + hinv_bytes(Gr, nA)) +
                (pinv_fits(G2, M) ? rnd256((size_t)nB * M * G2 * sizeof(float2))
                                  : rnd256((size_t)nB * G2 * G2 * sizeof(float2)) + hinv_bytes(G2, nB));
    // one pass over the dictionary per iteration (fused.hip); JSTSP_FUSED_PARTS = column ranges per problem
    // (measured at configs[1], 8 / 4 ranges: 4.42 / 4.34 ms per iteration - fewer partial sums to write and add)
    // (when M / 32 is not a multiple of 4: 2 ranges, or 1)
    const Tuning &tn = tune();          // the switches of this call (common.h), parsed at JSTSP_ENTER
    const int ftiles = (M % 32 == 0) ? M / 32 : 0;
    const int fparts = tn.fused_parts > 0 ? tn.fused_parts : (ftiles % 4 == 0 ? 4 : (ftiles % 2 == 0 ? 2 : 1));
    const bool want_fused = allow_fused && tn.fused != 0 && approx && Imax > 1 && fused_shape_ok(N, M, G2, fparts);
    if (want_fused) need += fused_bytes(M, G2, nB, batch, fparts);
    need += 1024;                                                                 // probe flags of the block-Toeplitz test
    if (approx && tn.gram_refine && tn.rv_comp) need += 2 * rnd256((size_t)batch * g * sizeof(float2));     // low-order parts of v, R v
    if (approx && tn.gram_refine)      // low-order part of G_A, G_B's first block row (hi, lo)
        need += rnd256((size_t)nA * Gr * Gr * sizeof(float2)) + 2 * rnd256((size_t)nB * (G2 / 2 + 1) * G2 * sizeof(float2));
    // a JSTSP_HOST dictionary of some size, one per trial, contiguous: tested for the block-Toeplitz structure on the host while
    // it is staged, and uploaded as its first block + leading columns (hostpack.hip)
    const bool host_compact = memspace == JSTSP_HOST && tn.host_compact != 0 && tn.toeplitz != 0 && nB > 1 && G2 >= 32 &&
                              strideB == (long long)G2 * M && (szB * sizeof(float2) >= ((size_t)64 << 20) || tn.host_compact >= 2) &&
                              (long long)G2 * M < (1ll << 31);       // (JSTSP_HOST_COMPACT=2: at any size - the tests)
    const size_t compact_elems = host_compact ? host_toeplitz_compact_elems(G2, M, nB) : 0;
    if (memspace == JSTSP_HOST) {
        need += rnd256(batch * nm * sizeof(float2)) + rnd256(batch * nm * sizeof(float)) +
                rnd256(szA * sizeof(float2)) + rnd256(szB * sizeof(float2));
        if (angles) need += rnd256(batch * g * sizeof(int32_t));
        need += rnd256(compact_elems * sizeof(float2));
    }
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();

    const float2 *subY, *A, *B;
    const float *Omega;
    const int32_t *indx_S = nullptr;
    if (memspace == JSTSP_HOST) host_trace("stage: begin");
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(subY_), batch * nm, memspace, &subY));
    JSTSP_TRY(stage_in(ctx, Omega_, batch * nm, memspace, &Omega));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), szA, memspace, &A));
    if (memspace == JSTSP_HOST) host_trace("stage: subY, Omega, A issued");
    int known_gt = ctx->dict_block_hint;        // (a caller that expanded a block-Toeplitz dictionary itself: c64.hip)
    ctx->dict_block_hint = 0;
    if (host_compact) {
        float2 *Bd = ctx->arena.get<float2>(szB), *Cd = ctx->arena.get<float2>(compact_elems);
        JSTSP_REQUIRE(Bd && Cd, JSTSP_E_NOMEM, "workspace exhausted while staging an input");
        JSTSP_TRY(host_toeplitz_stage(ctx, reinterpret_cast<const float2 *>(B_), G2, M, nB, Bd, Cd, compact_elems, &known_gt));
        if (!known_gt) JSTSP_HIP(hipMemcpyAsync(Bd, B_, szB * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
        B = Bd;
    } else
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(B_), szB, memspace, &B));
    if (angles) JSTSP_TRY(stage_in(ctx, indx_S_, batch * g, memspace, &indx_S));
    if (memspace == JSTSP_HOST) host_trace("stage: dictionary issued");
    if (memspace == JSTSP_DEVICE)
        JSTSP_REQUIRE(((uintptr_t)subY % 16 == 0) && ((uintptr_t)Omega % 8 == 0), JSTSP_E_ARG,
                      "device arrays must be 16-byte aligned");

    ProposedWS w;
    JSTSP_TRY(proposed_alloc(ctx->arena, w, N, M, Gr, G2, batch, nA, nB, angles, want_ce, std::max(Imax, 1)));

    // ---- per-problem scalars ------------------------------------------------------------------
    {
        std::vector<TrialParams> hp(batch);
        for (int t = 0; t < batch; ++t) hp[t] = make_trial_params(rho[t], tau_Y[t], tau_S[t]);
        JSTSP_TRY(upload(ctx, w.prm, hp.data(), batch * sizeof(TrialParams)));
    }

    // ---- setup: zero state (:8-12), iK1 (:14-20), Grams of the dictionary factors (:25) --------
    hipStream_t st = ctx->stream;
    JSTSP_HIP(hipMemsetAsync(w.X, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.V1, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.V2, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.C, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.Xs, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.Y, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.V, 0, batch * g * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.S, 0, batch * g * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(w.ce, 0, (size_t)batch * 3 * std::max(Imax, 1) * sizeof(double), st));   // ce(:,3) stays 0 for 'std' (:6)
    JSTSP_TRY(launch_inv_d(ctx, (long long)nm, batch, Omega, 2.f, w.prm, w.invD));
    // (the fused pass forms 1 / (Omega + 2 rho) itself, as two floats, when it can read Omega with its 16-byte loads)
    const bool omega_direct = ((uintptr_t)Omega % 16 == 0) && (nm % 4 == 0) && tn.inv_two_float != 0;
    if (angles) JSTSP_TRY(launch_rank_from_index(ctx, (int)g, batch, indx_S, w.rank));

    const Mat Am{A, strideA, N}, Bm{B, strideB, G2};
    // JSTSP_TOEPLITZ (fused.hip): 0 - the dictionary is taken as unstructured; 1 - a block-Toeplitz one gets the compact HBM image
    // of the fused pass (bit-identical results); 2 (default) - also the window kernel for block height 64 (fp32-equivalent, not
    // bit-identical)
    const int toep_env = tn.toeplitz;
    int toep_gt = 0;
    bool toep_probed = false;
    const Mat GAm{w.GA, strideA ? (long long)Gr * Gr : 0, Gr}, GBm{w.GB, strideB ? (long long)G2 * G2 : 0, G2};
    // G_A = A^H A (Gr x Gr), G_B = B B^H (G2 x G2):  R = K2'*K2 = G_B^T (x) G_A                      (:25)
    // These two are OPERATORS of the iteration: `R*v` (:47) stands next to `K2'*k`, which is computed through A and B
    // themselves, in every one of the Imax gradient steps - an error of a Gram is a constant bias of the gradient, not
    // rounding noise.  Measured (round 4, tools/precision_study.py and tools/parity_fixture_check.py against 2560 float64
    // solves): with G_A from a 64-term fp32 chain and G_B from the split-f16 product, rms |dNMSE| 3.9e-7, max 1.95e-6; the
    // fp32 storage of every array of the iteration together contributes 0.9e-7, a G_B held to 22 bits 1.2e-7.  So, for
    // 'approximate' (JSTSP_GRAM_REFINE=0: the round-3 products):
    //  * both Grams are formed in float64 from the fp32 inputs (G_A: gram64.hip, kept as TWO floats hi + lo; G_B: the fp32-MFMA
    //    product with fp64 master accumulators - of the first block row only when the dictionary is block-Toeplitz, the rest
    //    assembled in float64 - rounded once to fp32);
    //  * the iterations use G_A,hi and G_B as its 22-bit split-f16 pack for `R*res` (which only sets the step length); whenever
    //    R v itself is recomputed from v (every JSTSP_RV_REFRESH-th iteration) it is (G_A,hi + G_A,lo) V, then times G_B as the
    //    fp32-MFMA product with fp64 master accumulators.  Measured on 640 fixture trials, rms |dNMSE| / channel-estimates/s:
    //    plain refresh 2.21e-7 / 841, + fp64-master second factor 1.90e-7 / 823, + G_A,lo 1.76e-7 / 820; a low-order part of
    //    G_B on top changed nothing (1.76e-7 / 808) and is not kept.
    const bool refine = approx && tn.gram_refine != 0;
    // the 64-term products of the gradient step on the f16 pipe, fused into one launch (hsmall.hip)
    const bool use_head = refine && tn.grad_head != 0 && grad_head_shape_ok(N, Gr, G2);      // (bit 0: Res / P1; bit 1: first factor of R v)
    // JSTSP_RV_COMP=1 (opt-in): v and R v as two floats each (admm.hip: step_v_kernel): the recurrence R v += alpha R res then
    // tracks R times the v that was actually accumulated to about 48 bits, as the float64 reference's does; the low-order part of
    // a recomputed R v comes out of the fp64-master product (C_lo), the gradient subtracts both parts (cgemm: D_lo).  Measured
    // (2560 / 1280 fixture trials + the bench batch): with the default recomputation every 4th iteration max
    // |dNMSE| 7.8e-7 instead of 8.7e-7 for -1.2 % (two more arrays through the step kernel); with NO recomputation at all
    // (JSTSP_RV_REFRESH=1000) rms 2.04e-7, max 8.6e-7 over the 2560 sweep trials but 1.2e-6 on one trial of the bench batch, at
    // 845 instead of 810 channel-estimates/s; recomputing only now and then (every 16th, or at 0, 4, 8, 16, 32, 64) is WORSE than
    // never (1.5e-6, 2.0e-6): an exact R v is inconsistent with the operator the recurrence's split-f16 products realise, and
    // every recomputation is a kick of that size.
    const bool comp = refine && tn.rv_comp != 0 && Imax > 0;
    float2 *rv_lo_out = nullptr;
    if (comp) {
        w.Vlo = ctx->arena.get<float2>((size_t)batch * g);
        w.RVlo = ctx->arena.get<float2>((size_t)batch * g);
        JSTSP_REQUIRE(w.Vlo && w.RVlo, JSTSP_E_NOMEM, "proposed_algorithm: workspace exhausted (two-float accumulation)");
    }
    float2 *GAlo = nullptr;
    if (refine) {
        GAlo = ctx->arena.get<float2>((size_t)nA * Gr * Gr);
        JSTSP_REQUIRE(GAlo, JSTSP_E_NOMEM, "proposed_algorithm: workspace exhausted (Gram refinement)");
        JSTSP_TRY(gram_f64(ctx, 'L', A, strideA, N, Gr, nA, w.GA, (long long)Gr * Gr, GAlo));
    } else
    JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, Gr, N, nA, Am, Am, w.GA, (long long)Gr * Gr, Gr));
    // the block-Toeplitz probe (fused.hip) serves the Gram as well as the pass
    if (toep_env != 0 && known_gt && strideB) {        // found on the host while staging (exact by construction): no device probe
        toep_gt = known_gt;
        toep_probed = true;
    } else
    if (toep_env != 0 && (long long)G2 * M < (1ll << 31) && refine) {
        JSTSP_TRY(fused_probe_toeplitz(ctx, ctx->arena, B, strideB, G2, M, nB, &toep_gt));
        toep_probed = true;
    }
    if (w.h2) {
        // pack the dictionary first: G_B = B B^H is itself "a = B, b = conj(B)^T" on the split-f16 path
        JSTSP_TRY(hgemm_pack(ctx, w.Bc, ctx->arena, B, strideB, G2, 1, 1, M, G2, nB, (long long)G2 * M));
        // (the synthesis orientation w.Bs is packed at its first use: with the fused pass that is the last iteration of a
        //  three-output call and never in a two-output call - pack_bs below)
    }
    if (refine) {
        // (the assembly reads B columns below L = G2 / Gt and above M - L: it needs M >= L - a constant dictionary with fewer columns
        //  than delay blocks passes the probe and takes the full product)
        if (toep_gt && toep_env >= 2 && M >= G2 / toep_gt) {      // (JSTSP_TOEPLITZ=1 stays bit-identical to the unstructured path: full product there)
            // block (ld, ld') of G_B from block (0, ld' - ld) of the first block row and at most 3 (L - 1) products of leading /
            // trailing columns (fused.hip: toeplitz_gram_kernel), all in float64: 1 / L of the product
            float2 *G0 = ctx->arena.get<float2>((size_t)nB * toep_gt * G2), *G0lo = ctx->arena.get<float2>((size_t)nB * toep_gt * G2);
            JSTSP_REQUIRE(G0 && G0lo, JSTSP_E_NOMEM, "workspace exhausted (G_B block row)");
            GemmDesc dg = make_gemm('N', 'C', toep_gt, G2, M, nB, Bm, Bm, G0, (long long)toep_gt * G2, toep_gt);
            dg.force_m64 = 1; dg.C_lo = G0lo;
            JSTSP_TRY(launch_cgemm(ctx, dg, GEMM_MISC));
            JSTSP_TRY(toeplitz_gram_assemble(ctx, B, strideB, G2, M, toep_gt, nB, G0, G0lo, w.GB, nullptr));
            ctx->last_dict_block = toep_gt;        // (the structure has been used, whether or not the fused pass takes this shape)
        } else {
            GemmDesc dg = make_gemm('N', 'C', G2, G2, M, nB, Bm, Bm, w.GB, (long long)G2 * G2, G2);
            dg.force_m64 = 1; dg.herm_upper = 1;
            JSTSP_TRY(launch_cgemm(ctx, dg, GEMM_MISC));
            JSTSP_TRY(hermitian_fill_lower(ctx, w.GB, (long long)G2 * G2, G2, nB));
        }
    } else if (w.h2) {
        {
            // (G_B is Hermitian: the tiles below its diagonal - 12 of 32 at G2 = 512 - are not computed but mirrored)
            HGemmDesc hb{B, strideB, G2, w.Bc.bmax, w.Bc.data, w.Bc.st, w.Bc.bmax, 1, w.Bc.KS, w.Bc.JT, w.GB,
                         (long long)G2 * G2, G2, G2, G2, M, nB, EPI_NONE, nullptr, nullptr, nullptr};
            hb.herm_upper = 1;
            JSTSP_TRY(launch_hgemm(ctx, hb, nullptr));
            JSTSP_TRY(hermitian_fill_lower(ctx, w.GB, (long long)G2 * G2, G2, nB));
        }
    } else
    JSTSP_TRY(gemm(ctx, 'N', 'C', G2, G2, M, nB, Bm, Bm, w.GB, (long long)G2 * G2, G2));

    if (w.h2g && approx)
        JSTSP_TRY(hgemm_pack(ctx, w.GBp, ctx->arena, w.GB, strideB ? (long long)G2 * G2 : 0, 1, G2, 0, G2, G2, nB,
                             (long long)G2 * G2));
    // 'std': v = U\(L\k) (:29,:53) is the least-squares solution K2^+ k = vec(pinv(A) K pinv(B)) for K2 = kron(B.', A)
    // of full column rank.  Factors that fit the in-LDS float64 kernel get a true SVD-based pinv (pinv.hip: every shape
    // of the reference's drivers); larger ones the fp32 Gram inverse G^-1 (hinv.hip), kept in GA / GB, with its
    // conditioning recorded (JSTSP_E_ILLCOND / jstsp_last_conditioning).
    float2 *PA = nullptr, *PB = nullptr;          // pinv(A): Gr x N per problem;  pinv(B): M x G2 per problem
    if (!approx) {
        JSTSP_TRY(diag_reset(ctx));
        if (pinv_fits(N, Gr)) {
            PA = ctx->arena.get<float2>((size_t)nA * Gr * N);
            JSTSP_REQUIRE(PA, JSTSP_E_NOMEM, "proposed_algorithm 'std': workspace exhausted");
            JSTSP_TRY(launch_pinv(ctx, N, Gr, nA, A, strideA, N, PA, (long long)Gr * N, Gr));
        }
        if (pinv_fits(G2, M)) {
            PB = ctx->arena.get<float2>((size_t)nB * M * G2);
            JSTSP_REQUIRE(PB, JSTSP_E_NOMEM, "proposed_algorithm 'std': workspace exhausted");
            JSTSP_TRY(launch_pinv(ctx, G2, M, nB, B, strideB, G2, PB, (long long)M * G2, M));
        }
        float2 *GAi = PA ? nullptr : ctx->arena.get<float2>((size_t)nA * Gr * Gr);
        float2 *GBi = PB ? nullptr : ctx->arena.get<float2>((size_t)nB * G2 * G2);
        JSTSP_REQUIRE((PA || GAi) && (PB || GBi), JSTSP_E_NOMEM, "proposed_algorithm 'std': workspace exhausted");
        const size_t mark = ctx->arena.off;
        if (!PA) {
            JSTSP_TRY(hermitian_inverse(ctx, Gr, nA, w.GA, GAi));
            ctx->arena.off = mark;
            JSTSP_HIP(hipMemcpyAsync(w.GA, GAi, (size_t)nA * Gr * Gr * sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
        if (!PB) {
            JSTSP_TRY(hermitian_inverse(ctx, G2, nB, w.GB, GBi));
            ctx->arena.off = mark;
            JSTSP_HIP(hipMemcpyAsync(w.GB, GBi, (size_t)nB * G2 * G2 * sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
    }
    const long long snm = (long long)nm, sg = (long long)g, sng = (long long)ng;
    // Streams.  The critical path of an iteration is MFMA-bound
    //   [Y = Z - QZ, X/K/V1 update] -> K B^H -> Gram applies -> step -> A S B -> [C/V2 update]
    // and stays on the context's stream `sm`.  Two chains hang off it and run concurrently on side
    // streams, fed by events:
    //   s1: as soon as X, V1 of iteration i exist (after update_x) the NEXT iteration's SVT input
    //       Z = X - V1/rho, its Gram and the Jacobi eigen-decomposition (memory / latency bound);
    //   s2: the spectral norms of convergence_error(i,1:2): Gram of [X | V1] after update_x, Gram of
    //       V2 after update_c, then lambda_max of all three.
    // Buffer hazards are closed by events: update_x(i+1) waits for s1 (needs Q) and for s2's Gram of
    // [X | V1] (reads what update_x overwrites); update_c(i+1) waits for s2's Gram of V2.
    // The per-iteration memset of the operand maxima is double-buffered by iteration parity for the same reason.
    // Default: everything on the context's stream (same kernels, same arithmetic, identical
    // results).  JSTSP_OVERLAP=1 enables the side streams: measured +3 % channel-estimates/s at
    // BASELINE configs[1], but co-running kernels stretch each other (the K B^H launch goes from
    // 2.66 to 4.26 ms), which muddles per-kernel accounting — off until the side chains are
    // lighter than the MFMA-bound Grams they currently contain.
    // R v (`R*v` of :47) is recomputed from v every 4th iteration and carried by R v += alpha R res (the `R*res` of :48
    // is computed anyway) in between.  Measured at configs[1], 6 trials against the float64 oracle, refresh period
    // 1 / 4 / 8 / never: 514 / 529 / 533 / 536 channel-estimates/s, max |dNMSE| 2.4e-7 / 2.7e-7 / 2.7e-7 / 4.4e-7,
    // max |dS|/max|S| 2.5e-6 / 2.8e-6 / 3.2e-6 / 4.6e-6.  JSTSP_RV_REFRESH=1 recomputes every iteration.
    const int rv_refresh = std::max(1, tn.rv_refresh);
    auto refresh_at = [&](int it) -> bool { return it < tn.rv_always || it % rv_refresh == 0; };      // R v recomputed at iteration `it`?
    // With the fused pass the work between two passes is three short independent chains (Gram + eigen-decomposition of
    // the next Z | partial sums -> gradient step -> A S | spectral norms): there the side streams are on by default
    // (4.35 -> 4.21 ms per iteration at configs[1]).
    const bool overlap = tn.overlap >= 0 ? tn.overlap != 0 : want_fused;
    uint32_t *const kmax0 = w.kmax;
    JSTSP_TRY(ensure_side_streams(ctx));
    hipStream_t sm = ctx->stream, s1 = overlap ? ctx->side[0] : sm, s2 = overlap ? ctx->side[1] : sm;
    hipEvent_t ev_x = ctx->ev[0], ev_svt = ctx->ev[1], ev_gxv = ctx->ev[2], ev_c = ctx->ev[3], ev_gv2 = ctx->ev[4],
               ev_ce = ctx->ev[5], ev_lxv = ctx->ev[6];
    // iteration 0's SVT preparation on the main stream (X = V1 = 0)
    if (Imax > 0) {
        JSTSP_TRY(launch_form_z(ctx, snm, batch, w.X, w.V1, w.prm, w.Zb));
        JSTSP_TRY(svt_prepare(ctx, w.gz, w.Zb, w.prm, nullptr, true));
    }
    const bool fz = w.gz.left;                  // fused epilogues (need the Z - Q Z orientation)
    // opt-in: problems whose threshold is below the fp32 resolution of Z skip the Gram + eigen-decomposition (Y = Z)
    const bool svt_skip = tn.svt_skip != 0;
    const bool hmax = w.h2 && fz && N <= 64;    // the epilogues also deliver max|X|, |V1|, |V2|, |Znext|: split-f16 Grams
    // The svt argument Z = X - V1/rho is never stored on that path: the Gram kernel and the (I - Q) Z product form it
    // from X and V1 on the fly (both were written by the kernel before and are re-read while still close), which saves
    // one N x M array write per iteration in the X/K/V1 epilogue.
    // Only with convergence_error, where the Gram pass over X and V1 exists anyway and delivers G_z with it
    // (hgram3_kernel); a Z-only Gram from two sources costs more than the saved write.
    const bool zfly = hmax && want_ce && gram3_nsplit(N, M, G2, want_ce) > 0 && w.gz.nsplit == w.gn.nsplit;
    float2 *Zbuf[2] = {w.Zb, w.Zb2};            // svt argument of iteration it lives in Zbuf[it & 1]
    // Fused pass (fused.hip): after the gradient step of iteration it, ONE kernel forms Xs = A S B, the V2 / X / V1 / k
    // updates of :61-65 and of the next iteration's :38-43, and the first factor K B^H of the next :47 - the dictionary
    // is read once per iteration instead of twice.  The next iteration then starts at the gradient step.
    // Y = (I - Q) Z is formed inside the pass
    // with convergence_error: G_z comes from the three-Gram pass over X, V1 (zfly); without: from the Z the pass stores
    // (round 3: the opt-in short-cut JSTSP_SVT_SKIP=1 no longer switches the pass off - a skipped trial's Q = 0 becomes
    //  I - Q = I in the pass's fragments)
    const bool fusedp = want_fused && hmax && (want_ce ? zfly : true);
    const bool fusedy = fusedp;
    FusedWS fw;
    if (fusedp) {
        // a block-Toeplitz dictionary (what the reference's drivers build) is kept as its first block only: probed exactly,
        // results bit-identical either way (fused.hip); JSTSP_TOEPLITZ=0: always the full tile image
        // JSTSP_TOEPLITZ=2 (default): block height 64 with Y formed in the pass takes the window kernel (fused_pass64_kernel:
        // 20-KiB LDS tile, element-wise operands prefetched into LDS; the leading columns as fp32 corrections - not
        // bit-identical, fp32-equivalent); 1: the compact HBM image only (bit-identical to 0)
        if (toep_env != 0 && !toep_probed) JSTSP_TRY(fused_probe_toeplitz(ctx, ctx->arena, B, strideB, G2, M, nB, &toep_gt));
        ctx->last_dict_block = toep_gt;
        JSTSP_TRY(fused_alloc(ctx->arena, fw, M, G2, nB, batch, fparts, toep_gt, toep_env >= 2 && fusedy));
        JSTSP_TRY(fused_pack_b(ctx, fw, B, strideB, G2, M, nB, w.Bc.bmax));
        JSTSP_HIP(hipMemsetAsync(fw.ovf, 0, (size_t)batch * sizeof(uint32_t), sm));
    }
    // headroom (bits) of the k scale the pass predicts from the previous iteration's maximum; JSTSP_FUSED_KBACK is a
    // test hook: a negative value makes every pass overflow, which must end in the per-trial re-solve below
    const int fused_kback = tn.fused_kback;
    bool passed = false;                        // X, V1, k-partials of this iteration came from the previous pass
    // Order inside the window between two passes (measured in round 3).  Three chains start when a pass ends:
    // the gradient step (critical: the next pass waits for its A S), the three-Gram pass -> eigen-decomposition (the next pass
    // waits for I - Q), and the norms of convergence_error (Gram of V2 -> lambda_max; only the Gram must be done before the
    // next pass overwrites V2).  The Gram pass starts with the window, the eigen-decomposition behind the G_B apply of the
    // gradient step (a resident Jacobi workgroup leaves that apply no room on its CU), and beside a Jacobi the step kernel
    // runs with eight waves.  The alternatives that were measured and lost (Gram of V2 one window late with double-buffered
    // norm partials, the whole side chain behind the step's head, other gate positions, stream priorities) are gone from
    // the code; their numbers are in DESIGN.md.
    hipEvent_t ev_q1 = ctx->ev[7];
    // convergence_error(:,1:2): lambda_max of three Grams per trial and iteration, each warm-started from its own Ritz vector
    // of the previous iteration (eig2.hip); the record starts empty
    if (want_ce) JSTSP_TRY(lanczos_warm_reset(ctx, w.gn));
    for (int it = 0; it < Imax; ++it) {
        if (want_ce) w.gn.lz.call = it;
        float2 *Zc = fz ? Zbuf[it & 1] : w.Zb, *Zn = fz ? Zbuf[(it + 1) & 1] : w.Zb;
        // every operand maximum of this iteration starts from zero (one memset instead of four)
        // (block it & 1; every consumer of the same block from iteration it-2 has been waited for by the main stream
        //  during iteration it-1: ev_svt, ev_gxv, ev_gv2)
        if (w.h2) {
            const size_t boff = (size_t)(it & 1) * 8 * (size_t)batch;
            w.kmax = kmax0 + boff; w.nmax = w.kmax + batch; w.zmax = w.kmax + 4 * (size_t)batch;
            w.wmax = w.kmax + 5 * (size_t)batch; w.pmax = w.kmax + 6 * (size_t)batch;
            if (!passed) JSTSP_HIP(hipMemsetAsync(w.kmax, 0, 8 * (size_t)batch * sizeof(uint32_t), sm));
        } else if (w.h2g) JSTSP_HIP(hipMemsetAsync(w.pmax, 0, 2 * (size_t)batch * sizeof(uint32_t), sm));
        int apply_no = 0;
        // -- sub 1: Y = svt(X - V1/rho, tau_Y/rho) = Z - Q Z                                 (:35)
        if (it > 0) JSTSP_HIP(hipStreamWaitEvent(sm, ev_svt, 0));
        if (it > 0 && want_ce) JSTSP_HIP(hipStreamWaitEvent(sm, ev_gxv, 0));
        if (passed) {
            // sub 1, sub 2, k and the V1 update of this iteration were applied by the previous iteration's pass
        } else if (fz) {
            // Y = Z - Q Z with the X / K / V1 updates (and the next Z) applied to the tile in registers (:35-43,:64)
            // Y = Z - Q Z as (I - Q) Z: the beta-term form reads the Z tile a second time from HBM (PMC: +0.5 GB)
            JSTSP_TRY(launch_eye_minus(ctx, N, batch, w.gz.Q, w.gz.Q));
            GemmDesc dq = make_gemm('N', 'N', N, M, N, batch, Mat{w.gz.Q, (long long)N * N, N}, Mat{Zc, snm, N}, w.Y, snm, N);
            dq.epi = EPI_UPDATE_X; dq.prm = w.prm;
            dq.e_rw0 = w.V1; dq.e_w1 = w.X; dq.e_w2 = w.ZK;
            dq.e_r0 = w.V2; dq.e_r2 = w.Xs; dq.e_r3 = subY; dq.e_f0 = w.invD;
            dq.e_w3 = (it + 1 < Imax && (!zfly || fusedy)) ? Zn : nullptr;      // (the fused pass reads the stored Z)
            if (zfly && it > 0) { dq.B = w.X; dq.B2 = w.V1; }      // Z = X - V1/rho in the panel loader (it == 0: Z = 0 in Zc)
            dq.epi_store_c = (it + 1 == Imax);          // Y itself is only an output of the last iteration
            if (w.h2) {                                 // max|K| for the split-f16 correlation, from the same epilogue
                dq.amax_out = w.kmax;
                if (N <= 64) { dq.amax_x = w.nmax; dq.amax_v1 = w.nmax + batch; dq.amax_z = w.zmax; }
            }
            JSTSP_TRY(launch_cgemm(ctx, dq, GEMM_MISC));
        } else {
            JSTSP_TRY(svt_apply(ctx, w.gz, Zc, w.Y));
            // -- sub 2 + k of sub 3 + V1 dual update                                        (:38-43,:64)
            JSTSP_TRY(launch_update_x(ctx, snm, batch, w.X, w.V1, w.V2, w.C, w.Xs, w.Y, subY, w.invD, w.prm, w.ZK));
        }
        JSTSP_HIP(hipEventRecord(ev_x, sm));
        // s1: next iteration's Z, Gram, eigen-decomposition.  stage 0: all of it; 1: up to the Gram pass; 2: from the
        // eigen-decomposition on (JSTSP_SVT_ORDER: where in the window the two halves are issued, see below)
        auto issue_s1 = [&](int stage) -> int {
            if (stage != 2) JSTSP_HIP(hipStreamWaitEvent(s1, ev_x, 0));
            StreamScope sc(ctx, s1);
            if (stage != 2 && !fz) JSTSP_TRY(launch_form_z(ctx, snm, batch, w.X, w.V1, w.prm, Zn));   // fused: written by the epilogue
            if (zfly) {
              if (stage != 2) {
                // (the spectral norms of X, V1 of the previous iteration read the G_x, G_v1 partials this pass overwrites)
                if (fusedp && it > 0) JSTSP_HIP(hipStreamWaitEvent(s1, ev_lxv, 0));
                // (Until the library was compiled without packed-fp32 instructions - see build.py - this pass also had to
                // wait for the previous iteration's lambda_max kernels: the Lanczos kernel, whose complex arithmetic hipcc had
                // turned into v_pk_fma_f32 chains, returned different Ritz values when MFMA-heavy waves shared its SIMDs.)
                // one pass over X and V1: G_x, G_v1 (convergence_error) and G_z of Z = X - V1/rho (next svt)
                JSTSP_TRY(launch_hgram3(ctx, w.X, w.V1, snm, N, M, batch, w.gz.nsplit, w.nmax, w.nmax + batch, w.zmax,
                                        w.prm, w.gz.Gpart, w.gn.Gpart, w.gn.Gpart + (size_t)batch * N * N * w.gn.nsplit));
                JSTSP_HIP(hipEventRecord(ev_gxv, s1));
                if (fusedp) {       // lambda_max of G_x, G_v1 now (beside the pass), not after it with G_v2: the next Gram
                                    // pass then never waits for them
                    JSTSP_HIP(hipStreamWaitEvent(s2, ev_gxv, 0));
                    StreamScope sc2(ctx, s2);
                    JSTSP_TRY(lmax_from_partials_range(ctx, w.gn, 0, 2 * batch, w.lam, true));
                    JSTSP_HIP(hipEventRecord(ev_lxv, s2));
                }
              }
              if (stage == 1) return 0;
                JSTSP_TRY(svt_prepare(ctx, w.gz, w.X, w.prm, nullptr, true, w.zmax, svt_skip, nullptr, true));
                // the pass at the end of this iteration forms Y = (I - Q) Z itself: fragments of I - Q
                if (fusedy) JSTSP_TRY(fused_pack_wq(ctx, fw, w.gz.Q, batch));
            } else {
                JSTSP_TRY(svt_prepare(ctx, w.gz, Zn, w.prm, nullptr, true, hmax ? w.zmax : nullptr, svt_skip));
                if (fusedy) JSTSP_TRY(fused_pack_wq(ctx, fw, w.gz.Q, batch));
            }
            JSTSP_HIP(hipEventRecord(ev_svt, s1));
            return 0;
        };
        // behind a pass (with convergence_error): the Gram pass starts with the window, the eigen-decomposition behind the
        // gradient step's bandwidth-heavy head (profiles/r03_fused_iteration_timeline.txt); otherwise the whole side chain now
        const bool svt_split = passed && fusedp && zfly && it + 1 < Imax;
        if (it + 1 < Imax) JSTSP_TRY(issue_s1(svt_split ? 1 : 0));
        if (want_ce && !(zfly && it + 1 < Imax)) {              // s2: Gram of [X | V1]
            JSTSP_HIP(hipStreamWaitEvent(s2, ev_x, 0));
            StreamScope sc(ctx, s2);
            JSTSP_TRY(gram_partials_range(ctx, w.gn, w.X, snm, 0, 2 * batch, hmax ? w.nmax : nullptr, nullptr, nullptr, nullptr, true));      // (norms only: high f16 plane)
            JSTSP_HIP(hipEventRecord(ev_gxv, s2));
        }
        // -- sub 3: res = K2'*k - R*v                                                        (:47)
        //    Tc = K B^H  (N x G2), then Res = A^H Tc - G_A V G_B
        // (measured and dropped: 1. ONE fp32-FMA kernel for the sum of the partial sums, Res = A^H Tc - R v and G_A Res - three
        //  launches of the chain between two passes - 3.96 -> 4.42 ms per iteration: 64 KiB of A and G_A into LDS per workgroup
        //  and VALU products lose against the MFMA GEMM even at k = 64;  2. the last column range of a problem adding the
        //  partial sums inside the pass: the device-scope fence it needs writes the L2 back, 3.95 -> 4.36 ms)
        if (passed) {
            if (!use_head) JSTSP_TRY(fused_reduce(ctx, fw, G2, M, batch, w.Tc));      // (hsmall.hip sums the partial sums itself)
        } else if (PB) {       // 'std' with a float64 pinv of B:  Tc = K pinv(B)
            JSTSP_TRY(gemm(ctx, 'N', 'N', N, G2, M, batch, Mat{w.ZK, snm, N}, Mat{PB, strideB ? (long long)M * G2 : 0, M},
                           w.Tc, sng, N, 1.f, nullptr, 0, 0, 0.f, GEMM_CORRELATE));
        } else if (w.h2) {
            if (!fz) JSTSP_TRY(hgemm_absmax(ctx, w.ZK, snm, snm, batch, w.kmax));
            HGemmDesc hc{w.ZK, snm, N, w.kmax, w.Bc.data, strideB ? w.Bc.st : 0, w.Bc.bmax, strideB ? 1 : 0, w.Bc.KS,
                         w.Bc.JT, w.Tc, sng, N, N, G2, M, batch, EPI_NONE, nullptr, nullptr, nullptr};
            if (w.Kp.data && !strideB && w.Kp.KS == w.Bc.KS) {
                // one dictionary for the batch: k(n, m) in fragment order once (1 GiB read + written at configs[4] against 96 GiB
                // that the 1024 workgroups of the per-trial kernel split themselves), then two trials per workgroup
                w.Kp.bmax = w.kmax;
                JSTSP_TRY(hgemm_repack(ctx, w.Kp, w.ZK, snm, N, 1, 0, M, N, w.kmax));
                hc.Ap = w.Kp.data; hc.sApt = w.Kp.st; hc.aKS = w.Kp.KS;
            }
            JSTSP_TRY(launch_hgemm(ctx, hc, "correlate"));
        } else
        JSTSP_TRY(gemm(ctx, 'N', 'C', N, G2, M, batch, Mat{w.ZK, snm, N}, Bm, w.Tc, sng, N, 1.f, nullptr, 0, 0,
                       0.f, GEMM_CORRELATE));
        auto fused_reduce_if = [&](bool p) -> int { return p ? fused_reduce(ctx, fw, G2, M, batch, w.Tc) : 0; };
        const long long cnt_ll = std::min<long long>(10 + 5ll * (it + 1), (long long)g);
        // (G_A X) G_B: the G2 x G2 factor is packed once per solve; max|G_A X| comes from the first product's epilogue.
        // exact (R v itself is being formed from v): with the low-order parts of both Grams (see the setup above), the first factor
        // on the f16 pipe with G_A = hi + lo (hsmall.hip), the second as the fp32-MFMA product with fp64 master accumulators against
        // G_B,hi plus the split-f16 product against G_B,lo
        const Mat GAl{GAlo, strideA ? (long long)Gr * Gr : 0, Gr};
        auto second_factor = [&](float2 *out, uint32_t *pm, bool exact) -> int {
            if (exact || !w.h2g) {
                GemmDesc dr = make_gemm('N', 'N', Gr, G2, G2, batch, Mat{w.P1, sg, Gr}, GBm, out, sg, Gr);
                dr.force_m64 = exact ? 1 : 0;
                if (exact) dr.C_lo = rv_lo_out;         // (R v itself: what the fp32 result leaves of the float64 sums)
                return launch_cgemm(ctx, dr, GEMM_MISC);
            }
            HGemmDesc hg{w.P1, sg, Gr, pm, w.GBp.data, strideB ? w.GBp.st : 0, w.GBp.bmax, strideB ? 1 : 0,
                         w.GBp.KS, w.GBp.JT, out, sg, Gr, Gr, G2, G2, batch, EPI_NONE, nullptr, nullptr, nullptr};
            return launch_hgemm(ctx, hg, nullptr);
        };
        auto apply_R = [&](const float2 *Xin, float2 *out, bool exact) -> int {
            uint32_t *pm = w.pmax ? w.pmax + (size_t)(apply_no++ & 1) * batch : nullptr;      // two applies per iteration, one slot each
            if (use_head && exact && (tn.grad_head & 2)) {
                JSTSP_TRY(launch_left2(ctx, G2, batch, w.GA, GAlo, strideA ? (long long)Gr * Gr : 0, Xin, w.P1, pm));
            } else {
                const bool alo = exact;
                if (alo) JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, Gr, batch, GAl, Mat{Xin, sg, Gr}, w.P1, sg, Gr));      // P1 = G_A,lo X
                GemmDesc dp = make_gemm('N', 'N', Gr, G2, Gr, batch, GAm, Mat{Xin, sg, Gr}, w.P1, sg, Gr, 1.f, alo ? w.P1 : nullptr, sg,
                                        Gr, alo ? 1.f : 0.f);
                dp.amax_out = w.h2g ? pm : nullptr;
                JSTSP_TRY(launch_cgemm(ctx, dp, GEMM_MISC));
            }
            return second_factor(out, pm, exact);
        };
        if (approx) {
            // R v: recomputed from v every `rv_refresh` iterations, carried by R v += alpha R res in between (both are
            // `R*v` of :47; the recurrence alone drifts in fp32)
            const bool refreshed = refresh_at(it);
            if (refreshed) {
                rv_lo_out = comp ? w.RVlo : nullptr;
                JSTSP_TRY(apply_R(w.V, w.RV, refine));
                rv_lo_out = nullptr;
            }
            if (use_head && !(tn.grad_head & 1)) JSTSP_TRY(fused_reduce_if(passed));
            if (use_head && (tn.grad_head & 1)) {
                // Res = A^H Tc - R v and P1 = G_A Res in one kernel on the f16 pipe, straight from the pass's partial sums
                uint32_t *pm = w.pmax ? w.pmax + (size_t)(apply_no++ & 1) * batch : nullptr;
                if (passed)
                    JSTSP_TRY(launch_grad_head(ctx, G2, batch, fw.Ppart, (long long)fw.parts * sng, sng, fw.parts, fw.v2 ? fw.Kf : nullptr,
                                               fw.Bdl, fw.sBdl, A, strideA, w.GA, strideA ? (long long)Gr * Gr : 0, w.RV, nullptr, w.Res,
                                               w.P1, pm, comp ? w.RVlo : nullptr));
                else
                    JSTSP_TRY(launch_grad_head(ctx, G2, batch, w.Tc, sng, 0, 1, nullptr, nullptr, 0, A, strideA, w.GA,
                                               strideA ? (long long)Gr * Gr : 0, w.RV, nullptr, w.Res, w.P1, pm, comp ? w.RVlo : nullptr));
                //    R*res for alpha = res'*res / (res'*R*res)                                (:48)
                JSTSP_TRY(second_factor(w.RRes, pm, false));
            } else {
                GemmDesc dres = make_gemm('C', 'N', Gr, G2, N, batch, Am, Mat{w.Tc, sng, N}, w.Res, sg, Gr, 1.f, w.RV, sg, Gr, -1.f);
                dres.D_lo = comp ? w.RVlo : nullptr;
                JSTSP_TRY(launch_cgemm(ctx, dres, GEMM_MISC));
                //    R*res for alpha = res'*res / (res'*R*res)                                (:48)
                JSTSP_TRY(apply_R(w.Res, w.RRes, false));
            }
            if (svt_split) {
                JSTSP_HIP(hipEventRecord(ev_q1, sm));
                JSTSP_HIP(hipStreamWaitEvent(s1, ev_q1, 0));
                JSTSP_TRY(issue_s1(2));
            }
            //    v += alpha res; ce(i,3); s = soft(v) (.* Omega_S)                            (:49-56, angles :36,:68)
            JSTSP_TRY(launch_step_v(ctx, (int)g, batch, w.Res, w.RRes, w.V, w.S, w.rank, (int)cnt_ll, w.prm, w.ce,
                                    Imax, it, (rv_refresh > 1 || comp) ? w.RV : nullptr, svt_split, comp ? w.Vlo : nullptr,
                                    comp ? w.RVlo : nullptr, refreshed ? 1 : 0));
        } else {
            //    v = U\(L\k) = pinv(A) K pinv(B)   [ = G_A^-1 (A^H Tc) G_B^-1 on the Gram route: GA / GB hold the inverses ]  (:53)
            float2 *left = PB ? w.V : w.P1;        // result of the A side; the B side (if any) finishes into V
            if (PA)
                JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, N, batch, Mat{PA, strideA ? (long long)Gr * N : 0, Gr},
                               Mat{w.Tc, sng, N}, left, sg, Gr));
            else {
                JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, N, batch, Am, Mat{w.Tc, sng, N}, w.Res, sg, Gr));
                JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, Gr, batch, GAm, Mat{w.Res, sg, Gr}, left, sg, Gr));
            }
            if (!PB) JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, G2, batch, Mat{w.P1, sg, Gr}, GBm, w.V, sg, Gr));
            JSTSP_TRY(launch_soft(ctx, (int)g, batch, w.V, w.S, w.rank, (int)cnt_ll, w.prm));     // (:56)
        }
        // -- Xs = A S B                                                                      (:58)
        // (the last iteration of a two-output call on the fused path: Xs only feeds X, V2 and convergence_error, none of which
        //  is returned - S is final, Y was stored by the last pass)
        if (fusedp && !want_ce && it + 1 == Imax) break;
        if (w.h2) {
            GemmDesc dw = make_gemm('N', 'N', N, G2, Gr, batch, Am, Mat{w.S, sg, Gr}, w.W, sng, N);
            dw.amax_out = w.wmax;                       // max|A S| for the split-f16 synthesis
            JSTSP_TRY(launch_cgemm(ctx, dw, GEMM_MISC));
        } else
        JSTSP_TRY(gemm(ctx, 'N', 'N', N, G2, Gr, batch, Am, Mat{w.S, sg, Gr}, w.W, sng, N));
        if (it > 0 && want_ce) JSTSP_HIP(hipStreamWaitEvent(sm, ev_gv2, 0));
        passed = false;
        if (fusedp && it + 1 < Imax) {
            // the pass writes the operand maxima of iteration it + 1 (k, X, V1, next Z): zero that block now
            uint32_t *nx = kmax0 + (size_t)((it + 1) & 1) * 8 * (size_t)batch;
            JSTSP_HIP(hipMemsetAsync(nx, 0, 8 * (size_t)batch * sizeof(uint32_t), sm));
            JSTSP_TRY(fused_pack_as(ctx, fw, w.W, sng, G2, M, batch, w.wmax));
            JSTSP_HIP(hipStreamWaitEvent(sm, ev_svt, 0));          // Y of the next iteration (side stream s1)
            FusedDesc fd{fw.Bf, strideB ? fw.sBf : 0, w.Bc.bmax, strideB ? 1 : 0, fw.ASp, fw.sAS, w.wmax, w.kmax,
                         w.X, w.V1, w.V2, subY, w.Y, omega_direct ? Omega : w.invD, snm, w.prm, fw.Ppart,
                         nx, nx + batch, nx + 2 * (size_t)batch, nx + 4 * (size_t)batch, w.nmax + 2 * (size_t)batch, fw.ovf,
                         M, G2, batch, fparts,
                         fusedy ? fw.Wqp : nullptr, Zbuf[(it + 1) & 1], w.zmax,
                         (fw.v2 && zfly) ? nullptr : Zbuf[it & 1],      // (window kernel: Z comes from the staged X, V1)
                         (fusedy && it + 2 == Imax) ? w.Y : nullptr, fused_kback,
                         fw.Ec, strideB ? fw.sEc : 0, fw.gt ? 31 - __builtin_clz((unsigned)fw.gt) : 0, fw.ecols, fw.ehalo,
                         fw.v2, fw.XsD, fw.Kf, omega_direct ? 1 : 0};
            JSTSP_TRY(launch_fused_pass(ctx, fd));
            passed = true;
        } else
        if (w.h2) {
            // a(i, k = g) = W[i + N g] in fragment order, once per iteration instead of once per j-tile
            JSTSP_TRY(hgemm_repack(ctx, w.Wp, w.W, sng, N, 1, 0, G2, N, w.wmax));
            if (!w.Bs.data)
                JSTSP_TRY(hgemm_pack(ctx, w.Bs, ctx->arena, B, strideB, 1, G2, 0, G2, M, nB, (long long)G2 * M, w.Bc.bmax));
            HGemmDesc hs{w.W, sng, N, w.wmax, w.Bs.data, strideB ? w.Bs.st : 0, w.Bs.bmax, strideB ? 1 : 0, w.Bs.KS,
                         w.Bs.JT, w.Xs, snm, N, N, M, G2, batch, fz ? EPI_UPDATE_C : EPI_NONE, w.prm, w.X, w.V2,
                         hmax ? w.nmax + 2 * (size_t)batch : nullptr, w.Wp.data, w.Wp.st, w.Wp.KS};
            JSTSP_TRY(launch_hgemm(ctx, hs, "synthesize"));
            if (!fz) JSTSP_TRY(launch_update_c(ctx, snm, batch, w.X, w.Xs, w.V2, w.C, w.prm));
        } else if (fz) {
            // Xs = W B with sub 4 + the V2 dual update applied in the epilogue              (:58,:61,:65)
            GemmDesc ds = make_gemm('N', 'N', N, M, G2, batch, Mat{w.W, sng, N}, Bm, w.Xs, snm, N);
            ds.epi = EPI_UPDATE_C; ds.prm = w.prm;
            ds.e_r0 = w.X; ds.e_rw0 = w.V2;
            JSTSP_TRY(launch_cgemm(ctx, ds, GEMM_SYNTH));
        } else {
            JSTSP_TRY(gemm(ctx, 'N', 'N', N, M, G2, batch, Mat{w.W, sng, N}, Bm, w.Xs, snm, N, 1.f, nullptr, 0, 0,
                           0.f, GEMM_SYNTH));
            // -- sub 4 + V2 dual update                                                      (:61,:65)
            JSTSP_TRY(launch_update_c(ctx, snm, batch, w.X, w.Xs, w.V2, w.C, w.prm));
        }
        // -- convergence_error(i,1:2) = norm(V1)^2/norm(X)^2, norm(V2)^2/norm(X)^2           (:67,:69)
        if (want_ce) {
            JSTSP_HIP(hipEventRecord(ev_c, sm));
            JSTSP_HIP(hipStreamWaitEvent(s2, ev_c, 0));
            if (zfly) JSTSP_HIP(hipStreamWaitEvent(s2, ev_gxv, 0));      // G_x, G_v1 came from the side stream s1
            StreamScope sc(ctx, s2);
            JSTSP_TRY(gram_partials_range(ctx, w.gn, w.X, snm, 2 * batch, batch, hmax ? w.nmax : nullptr, nullptr, nullptr, nullptr, true));
            JSTSP_HIP(hipEventRecord(ev_gv2, s2));
            if (fusedp && zfly && it + 1 < Imax) JSTSP_TRY(lmax_from_partials_range(ctx, w.gn, 2 * batch, batch, w.lam, true));
            else JSTSP_TRY(lmax_from_partials(ctx, w.gn, w.lam, true));
            JSTSP_TRY(launch_ce_ratio(ctx, batch, w.lam + batch, w.lam + 2 * batch, w.lam, w.ce, Imax, it));
            JSTSP_HIP(hipEventRecord(ev_ce, s2));
        }
    }
    if (want_ce && Imax > 0) JSTSP_HIP(hipStreamWaitEvent(sm, ev_ce, 0));
    if (defer) {        // results stay in the workspace; copies, flags and recovery happen in proposed_finish
        defer->active = true; defer->fused = fusedp; defer->want_ce = want_ce && Imax > 0; defer->batch = batch; defer->g = g; defer->nm = nm;
        defer->Imax = Imax; defer->dS = w.S; defer->dY = w.Y; defer->dce = w.ce; defer->ovf = fusedp ? fw.ovf : nullptr;
        return 0;
    }

    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(S_out), w.S, batch * g, memspace));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(Y_out), w.Y, batch * nm, memspace));
    if (want_ce && Imax > 0) JSTSP_TRY(stage_out(ctx, ce_out, w.ce, (size_t)batch * 3 * Imax, memspace));
    if (fusedp && overflowed) {
        // Trials in which an entry of k left the f16 range of the scale predicted for it (growth beyond 2^(2 + kback) from
        // one iteration to the next): their results are wrong from that pass on.  The caller re-solves exactly those.
        std::vector<uint32_t> flags(batch);
        JSTSP_HIP(hipMemcpyAsync(flags.data(), fw.ovf, (size_t)batch * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        JSTSP_HIP(hipStreamSynchronize(st));
        for (int t = 0; t < batch; ++t)
            if (flags[t]) overflowed->push_back(t);
    }
    if (memspace == JSTSP_HOST) {
        JSTSP_HIP(hipStreamSynchronize(st));
        if (!approx) JSTSP_TRY(diag_check_host(ctx, "proposed_algorithm 'std'"));
    }
    return 0;
}

// Recovery: runs of consecutive flagged trials are solved again by the three-kernel iteration (every operand scale
// there is the exact maximum of data that already exists: it cannot overflow), straight into the caller's arrays -
// per-trial arrays are contiguous with the trial index slowest, so a sub-batch is a pointer offset in either memspace.
static int resolve_overflowed(jstsp_ctx *ctx, const std::vector<int> &ovf, int N, int M, int Gr, int G2, const jstsp_c32 *subY,
                              const float *Omega, const jstsp_c32 *A, long long strideA, const jstsp_c32 *B, long long strideB,
                              int Imax, const double *tau_Y, const double *tau_S, const double *rho, int type,
                              const int32_t *indx_S, jstsp_c32 *S_out, jstsp_c32 *Y_out, double *ce_out, int memspace, int *count)
{
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2;
    for (size_t i = 0; i < ovf.size();) {
        size_t j = i + 1;
        while (j < ovf.size() && ovf[j] == ovf[j - 1] + 1) ++j;
        const int t0 = ovf[i], cnt = (int)(j - i);
        JSTSP_TRY(proposed_impl(ctx, N, M, Gr, G2, cnt, subY + t0 * nm, Omega + t0 * nm, A + (size_t)t0 * strideA, strideA,
                                B + (size_t)t0 * strideB, strideB, Imax, tau_Y + t0, tau_S + t0, rho + t0, type,
                                indx_S ? indx_S + t0 * g : nullptr, S_out + t0 * g, Y_out ? Y_out + t0 * nm : nullptr,
                                ce_out ? ce_out + (size_t)t0 * 3 * Imax : nullptr, memspace, false, nullptr));
        *count += cnt;
        i = j;
    }
    return 0;
}

// Second phase of a deferred JSTSP_HOST solve: device -> host copies of the outputs, the overflow flags, one synchronisation.
static int proposed_finish(jstsp_ctx *ctx, const PendingSolve &p, jstsp_c32 *S_out, jstsp_c32 *Y_out, double *ce_out, std::vector<int> *ovf)
{
    JSTSP_ENTER(ctx);
    host_trace("finish: begin");
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(S_out), p.dS, p.batch * p.g, JSTSP_HOST));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(Y_out), p.dY, p.batch * p.nm, JSTSP_HOST));
    if (p.want_ce) JSTSP_TRY(stage_out(ctx, ce_out, p.dce, (size_t)p.batch * 3 * p.Imax, JSTSP_HOST));
    std::vector<uint32_t> flags(p.ovf ? p.batch : 0);
    if (p.ovf) JSTSP_HIP(hipMemcpyAsync(flags.data(), p.ovf, (size_t)p.batch * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    host_trace("finish: copies issued");
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    host_trace("finish: synchronised");
    for (int t = 0; t < (int)flags.size(); ++t)
        if (flags[t]) ovf->push_back(t);
    return 0;
}

extern "C" int jstsp_proposed_algorithm_c32(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch,
                                            const jstsp_c32 *subY, const float *Omega,
                                            const jstsp_c32 *A, long long strideA, const jstsp_c32 *B,
                                            long long strideB, int Imax, const double *tau_Y,
                                            const double *tau_S, const double *rho, int type,
                                            const int32_t *indx_S, jstsp_c32 *S_out, jstsp_c32 *Y_out,
                                            double *ce_out, int memspace)
{
    std::vector<int> ovf;
    if (ctx) { ctx->fused_fallbacks = 0; ctx->last_dict_block = 0; }
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2;
    // JSTSP_HOST with a large batch: the two halves of the batch on two contexts of the device, so that the upload of the second
    // half (4.75 GiB per 256 trials at BASELINE configs[1], 57 GB/s) runs while the first half is being solved, and the download
    // of the first half while the second is.  A trial's result does not depend on what else is in its batch (tests), so the
    // outputs are those of the one-call form.  Measured at configs[1] (tools/host_path_rate.py): 256 trials 544 -> 594
    // channel-estimates/s, 128 trials 514 -> 540, 64 trials 466 -> 391 (a 32-trial solve no longer fills the chip): from 128 on.
    const size_t in_bytes = (size_t)batch * (nm * 12 + (strideB ? (size_t)G2 * M * 8 : 0));
    const bool pipeline = ctx && memspace == JSTSP_HOST && type == JSTSP_TYPE_APPROXIMATE && batch >= 128 && Imax > 1 && Y_out &&
                          in_bytes >= ((size_t)256 << 20) && subY && Omega && A && B && tau_Y && tau_S && rho && S_out &&
                          tune_host_pipeline();
    if (pipeline) {
        if (!ctx->helper) JSTSP_TRY(jstsp_create(ctx->device, &ctx->helper));
        jstsp_ctx *cx[2] = {ctx, ctx->helper};
        const int h = ((batch / 2 + 7) / 8) * 8, cnt[2] = {h, batch - h}, t0[2] = {0, h};
        PendingSolve pend[2];
        // (a failure inside the pipeline - typically JSTSP_E_NOMEM for the helper's workspace - must not return while the other
        //  half still reads the caller's host arrays or writes its outputs: both streams are drained first)
        auto drained = [&](int rc) {
            if (rc) for (int k = 0; k < 2; ++k) { DeviceScope ds(cx[k]->device); (void)hipStreamSynchronize(cx[k]->stream); }
            return rc;
        };
#define JSTSP_TRY_PIPE(expr) do { int rc_ = drained(expr); if (rc_ != 0) return rc_; } while (0)
        for (int k = 0; k < 2; ++k) {
            host_trace("enqueue", k);
            cx[k]->fused_fallbacks = 0; cx[k]->last_dict_block = 0;
            JSTSP_TRY_PIPE(proposed_impl(cx[k], N, M, Gr, G2, cnt[k], subY + t0[k] * nm, Omega + t0[k] * nm, A + (size_t)t0[k] * strideA, strideA,
                                    B + (size_t)t0[k] * strideB, strideB, Imax, tau_Y + t0[k], tau_S + t0[k], rho + t0[k], type,
                                    indx_S ? indx_S + t0[k] * g : nullptr, S_out + t0[k] * g, Y_out + t0[k] * nm,
                                    ce_out ? ce_out + (size_t)t0[k] * 3 * Imax : nullptr, JSTSP_HOST, true, nullptr, &pend[k]));
        }
        host_trace("both halves enqueued");
        int fallbacks = 0;
        for (int k = 0; k < 2; ++k) {
            std::vector<int> o;
            JSTSP_TRY_PIPE(proposed_finish(cx[k], pend[k], S_out + t0[k] * g, Y_out + t0[k] * nm, ce_out ? ce_out + (size_t)t0[k] * 3 * Imax : nullptr, &o));
            JSTSP_TRY_PIPE(resolve_overflowed(cx[k], o, N, M, Gr, G2, subY + t0[k] * nm, Omega + t0[k] * nm, A + (size_t)t0[k] * strideA, strideA,
                                         B + (size_t)t0[k] * strideB, strideB, Imax, tau_Y + t0[k], tau_S + t0[k], rho + t0[k], type,
                                         indx_S ? indx_S + t0[k] * g : nullptr, S_out + t0[k] * g, Y_out + t0[k] * nm,
                                         ce_out ? ce_out + (size_t)t0[k] * 3 * Imax : nullptr, JSTSP_HOST, &fallbacks));
        }
#undef JSTSP_TRY_PIPE
        ctx->fused_fallbacks = fallbacks;
        ctx->last_dict_block = (cx[0]->last_dict_block == cx[1]->last_dict_block) ? cx[0]->last_dict_block : 0;
        return 0;
    }
    JSTSP_TRY(proposed_impl(ctx, N, M, Gr, G2, batch, subY, Omega, A, strideA, B, strideB, Imax, tau_Y, tau_S, rho, type,
                            indx_S, S_out, Y_out, ce_out, memspace, true, &ovf));
    int fallbacks = 0;
    JSTSP_TRY(resolve_overflowed(ctx, ovf, N, M, Gr, G2, subY, Omega, A, strideA, B, strideB, Imax, tau_Y, tau_S, rho, type, indx_S, S_out,
                                 Y_out, ce_out, memspace, &fallbacks));
    ctx->fused_fallbacks += fallbacks;
    return 0;
}

// ---- the two-phase form for device arrays (include/jstsp.h) ---------------------------------------------------------------------
struct jstsp_pending {
    int N, M, Gr, G2, batch, Imax, type;
    const jstsp_c32 *subY, *A, *B;
    const float *Omega;
    long long strideA, strideB;
    const int32_t *indx_S;
    jstsp_c32 *S_out, *Y_out;
    double *ce_out;
    std::vector<double> tau_Y, tau_S, rho;
    uint32_t *flags = nullptr;          // pinned host copy of the per-trial overflow flags (NULL: the solve did not use the fused pass)
    hipEvent_t done = nullptr;
};

extern "C" int jstsp_proposed_algorithm_begin_c32(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *subY,
                                                  const float *Omega, const jstsp_c32 *A, long long strideA, const jstsp_c32 *B,
                                                  long long strideB, int Imax, const double *tau_Y, const double *tau_S,
                                                  const double *rho, int type, const int32_t *indx_S, jstsp_c32 *S_out,
                                                  jstsp_c32 *Y_out, double *ce_out, jstsp_pending **pending)
{
    JSTSP_REQUIRE(ctx && pending, JSTSP_E_NULL, "proposed_algorithm_begin: NULL context or handle pointer");
    *pending = nullptr;
    ctx->fused_fallbacks = 0; ctx->last_dict_block = 0;
    PendingSolve ps;
    JSTSP_TRY(proposed_impl(ctx, N, M, Gr, G2, batch, subY, Omega, A, strideA, B, strideB, Imax, tau_Y, tau_S, rho, type, indx_S, S_out,
                            Y_out, ce_out, JSTSP_DEVICE, true, nullptr, &ps));
    JSTSP_ENTER(ctx);
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(S_out), ps.dS, ps.batch * ps.g, JSTSP_DEVICE));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(Y_out), ps.dY, ps.batch * ps.nm, JSTSP_DEVICE));
    if (ps.want_ce) JSTSP_TRY(stage_out(ctx, ce_out, ps.dce, (size_t)ps.batch * 3 * ps.Imax, JSTSP_DEVICE));
    jstsp_pending *p = new (std::nothrow) jstsp_pending();
    JSTSP_REQUIRE(p, JSTSP_E_NOMEM, "proposed_algorithm_begin: out of host memory");
    p->N = N; p->M = M; p->Gr = Gr; p->G2 = G2; p->batch = batch; p->Imax = Imax; p->type = type;
    p->subY = subY; p->A = A; p->B = B; p->Omega = Omega; p->strideA = strideA; p->strideB = strideB; p->indx_S = indx_S;
    p->S_out = S_out; p->Y_out = Y_out; p->ce_out = ce_out;
    p->tau_Y.assign(tau_Y, tau_Y + batch); p->tau_S.assign(tau_S, tau_S + batch); p->rho.assign(rho, rho + batch);
    hipError_t e = hipSuccess;          // (positive status = the HIP error code, as everywhere in this ABI)
    if (ps.ovf) {
        e = hipHostMalloc((void **)&p->flags, (size_t)batch * sizeof(uint32_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMemcpyAsync(p->flags, ps.ovf, (size_t)batch * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(p->done, ctx->stream);
    const int rc = (int)e;
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);
        if (p->flags) (void)hipHostFree(p->flags);
        if (p->done) (void)hipEventDestroy(p->done);
        delete p;
        set_error("proposed_algorithm_begin: could not set up the completion record: %s", hipGetErrorString(e));
        return rc;
    }
    *pending = p;
    return 0;
}

extern "C" int jstsp_proposed_algorithm_end(jstsp_ctx *ctx, jstsp_pending *p, int *fallbacks)
{
    JSTSP_REQUIRE(ctx && p, JSTSP_E_NULL, "proposed_algorithm_end: NULL context or handle");
    JSTSP_ENTER(ctx);
    int rc = 0, count = 0;
    const hipError_t e = hipEventSynchronize(p->done);
    if (e != hipSuccess) { set_error("proposed_algorithm_end: waiting for the solve failed: %s", hipGetErrorString(e)); rc = (int)e; }
    std::vector<int> ovf;
    if (!rc && p->flags)
        for (int t = 0; t < p->batch; ++t)
            if (p->flags[t]) ovf.push_back(t);
    if (!rc && !ovf.empty())
        rc = resolve_overflowed(ctx, ovf, p->N, p->M, p->Gr, p->G2, p->subY, p->Omega, p->A, p->strideA, p->B, p->strideB, p->Imax,
                                p->tau_Y.data(), p->tau_S.data(), p->rho.data(), p->type, p->indx_S, p->S_out, p->Y_out, p->ce_out,
                                JSTSP_DEVICE, &count);
    ctx->fused_fallbacks = count;
    if (fallbacks) *fallbacks = count;
    if (p->flags) (void)hipHostFree(p->flags);
    (void)hipEventDestroy(p->done);
    delete p;
    return rc;
}

namespace jstsp {
// (solver_common.h: the phases of a pipelined host call for c64.hip, which stages - and narrows - its inputs itself)
int proposed_enqueue_device(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *subY, const float *Omega,
                            const jstsp_c32 *A, long long strideA, const jstsp_c32 *B, long long strideB, int Imax,
                            const double *tau_Y, const double *tau_S, const double *rho, int type, const int32_t *indx_S,
                            bool want_ce, PendingSolve *out)
{
    static jstsp_c32 sink_c;            // (a deferred solve never writes its output arguments; they only say what is wanted)
    static double sink_d;
    if (ctx) { ctx->fused_fallbacks = 0; ctx->last_dict_block = 0; }
    return proposed_impl(ctx, N, M, Gr, G2, batch, subY, Omega, A, strideA, B, strideB, Imax, tau_Y, tau_S, rho, type, indx_S, &sink_c,
                         &sink_c, want_ce ? &sink_d : nullptr, JSTSP_DEVICE, true, nullptr, out);
}
int proposed_pending_flags(jstsp_ctx *ctx, const PendingSolve &p, std::vector<int> *ovf)
{
    JSTSP_ENTER(ctx);
    std::vector<uint32_t> flags(p.ovf ? p.batch : 0);
    if (p.ovf) JSTSP_HIP(hipMemcpyAsync(flags.data(), p.ovf, (size_t)p.batch * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    host_trace("finish: copies issued");
    JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    host_trace("finish: synchronised");
    for (int t = 0; t < (int)flags.size(); ++t)
        if (flags[t]) ovf->push_back(t);
    return 0;
}
int proposed_resolve_device(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *subY, const float *Omega,
                            const jstsp_c32 *A, long long strideA, const jstsp_c32 *B, long long strideB, int Imax,
                            const double *tau_Y, const double *tau_S, const double *rho, int type, const int32_t *indx_S,
                            jstsp_c32 *S_dev, jstsp_c32 *Y_dev, double *ce_dev)
{
    return proposed_impl(ctx, N, M, Gr, G2, batch, subY, Omega, A, strideA, B, strideB, Imax, tau_Y, tau_S, rho, type, indx_S, S_dev,
                         Y_dev, ce_dev, JSTSP_DEVICE, false, nullptr);
}
}  // namespace jstsp

/* Trials of the last jstsp_proposed_algorithm_* call on this context that were solved a second time by the three-kernel
 * iteration because the fused pass's predicted operand scale overflowed for them (0 in all but pathological inputs). */
extern "C" int jstsp_last_fused_fallbacks(jstsp_ctx *ctx, int *count)
{
    JSTSP_REQUIRE(ctx && count, JSTSP_E_NULL, "last_fused_fallbacks: NULL argument");
    *count = ctx->fused_fallbacks;
    return 0;
}
/* Block height Gt of the block-Toeplitz structure the last jstsp_proposed_algorithm_* call found in its dictionary
 * (B(ld Gt + g, m) == B(g, m - ld) bit for bit; the pass then streams the first block only), 0 if it found none. */
extern "C" int jstsp_last_dictionary_block(jstsp_ctx *ctx, int *gt)
{
    JSTSP_REQUIRE(ctx && gt, JSTSP_E_NULL, "last_dictionary_block: NULL argument");
    *gt = ctx->last_dict_block;
    return 0;
}
