// Kernel-level entry points (correlate / synthesize), svt, the matrix-completion baselines
// mc_svt / mc_admm and the spectral-norm NMSE of the drivers.
#include "solver_common.h"
#include <algorithm>

namespace jstsp {

__global__ __launch_bounds__(256) void diff_kernel(long long n, const float2 *a, const float2 *b, float2 *o)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        o[i] = make_float2(a[i].x - b[i].x, a[i].y - b[i].y);
}

// nmse = min(1, num/den)   (plot_errorVSsnr.m:138-141; NaN stays NaN: `NaN > 1` is false)
__global__ void ratio_cap_kernel(int batch, const float *num, const float *den, double *out, int cap,
                                 long long ostride, long long ooff)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < batch) {
        double e = (double)num[t] / (double)den[t];
        if (cap && e > 1.0) e = 1.0;
        out[(long long)t * ostride + ooff] = e;
    }
}

// mc_svt.m:9   Y = Y + rho (OH - Omega .* X)
__global__ __launch_bounds__(256) void mc_svt_update_kernel(long long nm, float2 *Y, const float2 *OH,
                                                            const float *Omega, const float2 *X,
                                                            const TrialParams *prm)
{
    const int t = blockIdx.y;
    const TrialParams p = prm[t];               // (rho as two floats: common.h)
    const long long base = (long long)t * nm, stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nm; i += stride) {
        const float om = Omega[base + i];
        const float2 x = X[base + i], oh = OH[base + i];
        float2 y = Y[base + i];
        const float dx = oh.x - om * x.x, dy = oh.y - om * x.y;
        y.x = fmaf(p.rho, dx, y.x) + p.rho_lo * dx;
        y.y = fmaf(p.rho, dy, y.y) + p.rho_lo * dy;
        Y[base + i] = y;
    }
}

// mc_admm.m:24-26   Y = (OH + Z + rho X) ./ (Omega + rho);  Z = Z + rho (X - Y);  Zn = Y - Z/rho (next svt arg)
__global__ __launch_bounds__(256) void mc_admm_update_kernel(long long nm, float2 *Y, float2 *Z,
                                                             const float2 *OH, const float *invD,
                                                             const float2 *X, const TrialParams *prm,
                                                             float2 *Zn)
{
    const int t = blockIdx.y;
    const TrialParams p = prm[t];               // (rho, 1/rho as two floats each: common.h)
    const long long base = (long long)t * nm, stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nm; i += stride) {
        const float id = invD[base + i];
        const float2 x = X[base + i], oh = OH[base + i];
        float2 z = Z[base + i];
        const float2 y = make_float2(((oh.x + z.x) + mul2(p.rho, p.rho_lo, x.x)) * id, ((oh.y + z.y) + mul2(p.rho, p.rho_lo, x.y)) * id);
        z.x = admm_v1(p, z.x, x.x, y.x);        // Z + rho (X - Y)
        z.y = admm_v1(p, z.y, x.y, y.y);
        Y[base + i] = y;
        Z[base + i] = z;
        Zn[base + i] = make_float2(admm_z(p, y.x, z.x), admm_z(p, y.y, z.y));
    }
}

// rate[t] = sum_i log2(1 + lam_i / (R (noise_var + num/den)))   (plot_rateVSframelength.m:81: log2 det(I + Zbar Zbar'/(R (..))))
__global__ __launch_bounds__(256) void rate_kernel(int batch, int n, int R, const float *lam, const float *num,
                                                   const float *den, double noise_var, double *rate)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= batch) return;
    const double e = (double)num[t] / (double)den[t];
    const double c = 1.0 / ((double)R * (noise_var + e));
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += log2(1.0 + fmax((double)lam[(long long)t * n + i], 0.0) * c);
    rate[t] = acc;
}

// ---- equilibration of the public correlate / synthesize on the split-f16 path -----------------------------------
// The split x = h + l of hgemm.hip is exact to 2^-23 of the per-PROBLEM maximum.  A row of K (or of B) far below that
// maximum would keep fewer digits than an fp32 product gives it, although the output row it produces depends on nothing
// else.  Rows / columns along the indices that are NOT contracted are therefore scaled by exact powers of two to a
// common magnitude before the split and scaled back afterwards: every output row and column then carries fp32 relative
// accuracy with respect to its own operands.  (Along the contracted index the accuracy is that of the sum's largest terms,
// as in any fp32 product whose addends differ in magnitude.)

// e[t][i] = exponent of max(|re|, |im|) over row i (axis 0) or column i (axis 1) of the rows x cols matrix X[t]; 0 for a zero line
__global__ __launch_bounds__(256) void axis_exp_kernel(int rows, int cols, const float2 *X, long long sXt, int axis, int32_t *e)
{
    const int t = blockIdx.y, tid = threadIdx.x;
    const float2 *x = X + (long long)t * sXt;
    __shared__ float red[256];
    if (axis == 0) {
        const int r = blockIdx.x * 64 + (tid & 63), ph = tid >> 6;
        float m = 0.f;
        if (r < rows)
            for (int c = ph; c < cols; c += 4) {
                const float2 v = x[r + (long long)rows * c];
                m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
            }
        red[tid] = m;
        __syncthreads();
        if (ph == 0 && r < rows) {
            m = fmaxf(fmaxf(red[tid], red[tid + 64]), fmaxf(red[tid + 128], red[tid + 192]));
            e[(long long)t * rows + r] = (m > 0.f && m < INFINITY) ? ilogbf(m) : 0;
        }
    } else {
        const int c = blockIdx.x * 4 + (tid >> 6), l = tid & 63;
        float m = 0.f;
        if (c < cols)
            for (int r = l; r < rows; r += 64) {
                const float2 v = x[r + (long long)rows * c];
                m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (l == 0 && c < cols) e[(long long)t * cols + c] = (m > 0.f && m < INFINITY) ? ilogbf(m) : 0;
    }
}

// out[t][i, j] = in[t][i, j] * 2^(sign * (er[t][i] + ec[t][j]))   (er / ec may be NULL; strides 0 = shared by the batch)
__global__ __launch_bounds__(256) void scale2_kernel(int rows, int cols, const float2 *in, long long sIn, float2 *out, long long sOut,
                                                     const int32_t *er, long long ser, const int32_t *ec, long long sec, int sign)
{
    const int t = blockIdx.y;
    const long long n = (long long)rows * cols, stride = (long long)gridDim.x * 256;
    const float2 *x = in + (long long)t * sIn;
    float2 *o = out + (long long)t * sOut;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i % rows), c = (int)(i / rows);
        const int ex = sign * ((er ? er[(long long)t * ser + r] : 0) + (ec ? ec[(long long)t * sec + c] : 0));
        const float2 v = x[i];
        o[i] = make_float2(ldexpf(v.x, ex), ldexpf(v.y, ex));
    }
}

static int axis_exp(jstsp_ctx *ctx, int rows, int cols, const float2 *X, long long sXt, int count, int axis, int32_t *e)
{
    const dim3 g(axis == 0 ? (rows + 63) / 64 : (cols + 3) / 4, count);
    hipLaunchKernelGGL(axis_exp_kernel, g, dim3(256), 0, ctx->stream, rows, cols, X, sXt, axis, e);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

static dim3 grid2(long long n, int batch);
static int scale2(jstsp_ctx *ctx, int rows, int cols, int count, const float2 *in, long long sIn, float2 *out, long long sOut,
                  const int32_t *er, long long ser, const int32_t *ec, long long sec, int sign)
{
    hipLaunchKernelGGL(scale2_kernel, grid2((long long)rows * cols, count), dim3(256), 0, ctx->stream, rows, cols, in, sIn, out,
                       sOut, er, ser, ec, sec, sign);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

static dim3 grid2(long long n, int batch)
{
    long long blocks = std::max<long long>(1, std::min<long long>((n + 255) / 256, (4096 + batch - 1) / batch));
    return dim3((unsigned)blocks, (unsigned)batch);
}

static int check_common(jstsp_ctx *ctx, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    return 0;
}

static int upload_tau_rho(jstsp_ctx *ctx, int batch, const double *tau, const double *rho, TrialParams *prm)
{
    std::vector<TrialParams> hp(batch);
    for (int t = 0; t < batch; ++t) {
        hp[t] = make_trial_params(rho ? rho[t] : 1.0, tau[t], 0.0);
    }
    return upload(ctx, prm, hp.data(), batch * sizeof(TrialParams));
}

}  // namespace jstsp

using namespace jstsp;

// mc_svt / mc_admm are fixed-point loops whose svt argument moves little from one iteration to the next: the eigen-decomposition of
// iteration i starts from the basis of iteration i - 1 and (orders 65..128, GramWS::eig_stop) runs no further sweep once the Gram
// in that basis has relative off-diagonals below this level - an inexact inner solve.  Measured at configs[2] (128 x 128, 20
// iterations, 16 trials against the float64 oracle): max error of X 5.18e-5 / 9.15e-5 (mc_svt / mc_admm) with the level at 0,
// 5.15e-5 / 9.16e-5 at 1e-4 (the default), 5.13e-5 / 9.18e-5 at 1e-3, 5.5e-5 / 5.5e-4 at 1e-2; mc_svt x 20 at 1024 trials
// 0.112 / 0.066 / 0.051 / 0.050 s.  JSTSP_MC_EIG_STOP overrides (0: every call converged to the end-of-sweep rule).
static float mc_eig_stop()
{
    const char *e = getenv("JSTSP_MC_EIG_STOP");
    return e ? (float)atof(e) : 1e-4f;
}

// One packed b operand for the batch and a shape hgemm_pair_kernel takes: the a operand (rows x Kd per problem, column-major, leading
// dimension `rows`) in fragment order as well - the two-trials-per-workgroup kernel reads packed operands only (hgemm.hip)
static size_t pair_apack_bytes(int Kd, int batch) { return rnd256((size_t)batch * 2 * (4 * ((Kd + 63) / 64)) * 256 * sizeof(uint4)); }
static int pair_apack(jstsp_ctx *ctx, const float2 *Aop, long long sAt, int rows, int Kd, int batch, const uint32_t *amax, const HPack &bp,
                      HGemmDesc &hd)
{
    HPack ap;
    ap.KS = 4 * ((Kd + 63) / 64); ap.JT = 2; ap.count = batch;
    ap.st = (long long)ap.JT * ap.KS * 256;
    if (ap.KS != bp.KS) return 0;
    ap.data = ctx->arena.get<uint4>((size_t)batch * ap.st);
    JSTSP_REQUIRE(ap.data, JSTSP_E_NOMEM, "workspace exhausted (packed a operand)");
    JSTSP_TRY(hgemm_repack(ctx, ap, Aop, sAt, rows, 1, 0, Kd, rows, amax));
    hd.Ap = ap.data; hd.sApt = ap.st; hd.aKS = ap.KS;
    return 0;
}

extern "C" {

int jstsp_correlate_c32(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *K_,
                        const jstsp_c32 *A_, long long strideA, const jstsp_c32 *B_, long long strideB,
                        jstsp_c32 *out, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(K_ && A_ && B_ && out, JSTSP_E_NULL, "correlate: NULL array argument");
    JSTSP_REQUIRE(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0, JSTSP_E_SHAPE, "correlate: bad shape");
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2, ng = (size_t)N * G2;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t szB = strideB ? (size_t)strideB * (batch - 1) + (size_t)G2 * M : (size_t)G2 * M;
    size_t need = rnd256(batch * ng * sizeof(float2)) + rnd256(batch * g * sizeof(float2));
    const bool h2 = use_hgemm(N, G2, M);
    const int nB = strideB ? batch : 1;
    if (h2) need += hgemm_pack_bytes(M, G2, nB) + rnd256(batch * sizeof(uint32_t)) + rnd256(batch * nm * sizeof(float2)) +
                    rnd256((size_t)nB * G2 * M * sizeof(float2)) + rnd256((size_t)batch * N * sizeof(int32_t)) +
                    rnd256((size_t)nB * G2 * sizeof(int32_t));
    const bool pair = h2 && !strideB && hgemm_pair_shape(N, G2, batch);
    if (pair) need += pair_apack_bytes(M, batch);
    if (memspace == JSTSP_HOST)
        need += rnd256(batch * nm * sizeof(float2)) + rnd256(szA * sizeof(float2)) + rnd256(szB * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *K, *A, *B;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(K_), batch * nm, memspace, &K));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), szA, memspace, &A));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(B_), szB, memspace, &B));
    float2 *Tc = ctx->arena.get<float2>(batch * ng);
    float2 *O = ctx->arena.get<float2>(batch * g);
    JSTSP_REQUIRE(Tc && O, JSTSP_E_NOMEM, "correlate: workspace exhausted");
    // A^H (K B^H): the cheaper association (N*M*G2 + Gr*N*G2 MACs)
    if (h2) {
        // rows of K (index n) and rows of B (index g) are not contracted: equilibrate them (see axis_exp_kernel)
        int32_t *eK = ctx->arena.get<int32_t>((size_t)batch * N), *eB = ctx->arena.get<int32_t>((size_t)nB * G2);
        float2 *Ks = ctx->arena.get<float2>(batch * nm), *Bs = ctx->arena.get<float2>((size_t)nB * G2 * M);
        JSTSP_REQUIRE(eK && eB && Ks && Bs, JSTSP_E_NOMEM, "correlate: workspace exhausted");
        const long long sBs = strideB ? (long long)G2 * M : 0;
        JSTSP_TRY(axis_exp(ctx, N, M, K, (long long)nm, batch, 0, eK));
        JSTSP_TRY(scale2(ctx, N, M, batch, K, (long long)nm, Ks, (long long)nm, eK, N, nullptr, 0, -1));
        JSTSP_TRY(axis_exp(ctx, G2, M, B, strideB, nB, 0, eB));
        JSTSP_TRY(scale2(ctx, G2, M, nB, B, strideB, Bs, sBs, eB, G2, nullptr, 0, -1));
        // b(k = m, j = g) = conj(B[g + G2 m]) packed once; a = K
        HPack pk;
        JSTSP_TRY(hgemm_pack(ctx, pk, ctx->arena, Bs, sBs, G2, 1, 1, M, G2, nB, (long long)G2 * M));
        uint32_t *amax = ctx->arena.get<uint32_t>(batch);
        JSTSP_REQUIRE(amax, JSTSP_E_NOMEM, "correlate: workspace exhausted");
        JSTSP_TRY(hgemm_absmax(ctx, Ks, (long long)nm, (long long)nm, batch, amax));
        HGemmDesc hd{Ks, (long long)nm, N, amax, pk.data, strideB ? pk.st : 0, pk.bmax, strideB ? 1 : 0, pk.KS, pk.JT,
                     Tc, (long long)ng, N, N, G2, M, batch, EPI_NONE, nullptr, nullptr, nullptr};
        if (pair) JSTSP_TRY(pair_apack(ctx, Ks, (long long)nm, N, M, batch, amax, pk, hd));
        JSTSP_TRY(launch_hgemm(ctx, hd, "correlate"));
        JSTSP_TRY(scale2(ctx, N, G2, batch, Tc, (long long)ng, Tc, (long long)ng, eK, N, eB, strideB ? G2 : 0, +1));
    } else
    JSTSP_TRY(gemm(ctx, 'N', 'C', N, G2, M, batch, Mat{K, (long long)nm, N}, Mat{B, strideB, G2}, Tc,
                   (long long)ng, N, 1.f, nullptr, 0, 0, 0.f, GEMM_CORRELATE));
    JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, N, batch, Mat{A, strideA, N}, Mat{Tc, (long long)ng, N}, O,
                   (long long)g, Gr));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out), O, batch * g, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_synthesize_c32(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *S_,
                         const jstsp_c32 *A_, long long strideA, const jstsp_c32 *B_, long long strideB,
                         jstsp_c32 *out, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(S_ && A_ && B_ && out, JSTSP_E_NULL, "synthesize: NULL array argument");
    JSTSP_REQUIRE(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0, JSTSP_E_SHAPE, "synthesize: bad shape");
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2, ng = (size_t)N * G2;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t szB = strideB ? (size_t)strideB * (batch - 1) + (size_t)G2 * M : (size_t)G2 * M;
    size_t need = rnd256(batch * ng * sizeof(float2)) + rnd256(batch * nm * sizeof(float2));
    const bool h2 = use_hgemm(N, M, G2);
    const int nB = strideB ? batch : 1;
    if (h2) need += hgemm_pack_bytes(G2, M, nB) + rnd256(batch * sizeof(uint32_t)) + rnd256((size_t)nB * G2 * M * sizeof(float2)) +
                    rnd256((size_t)batch * N * sizeof(int32_t)) + rnd256((size_t)nB * M * sizeof(int32_t));
    const bool pair = h2 && !strideB && hgemm_pair_shape(N, M, batch);
    if (pair) need += pair_apack_bytes(G2, batch);
    if (memspace == JSTSP_HOST)
        need += rnd256(batch * g * sizeof(float2)) + rnd256(szA * sizeof(float2)) + rnd256(szB * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *S, *A, *B;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(S_), batch * g, memspace, &S));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), szA, memspace, &A));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(B_), szB, memspace, &B));
    float2 *W = ctx->arena.get<float2>(batch * ng);
    float2 *O = ctx->arena.get<float2>(batch * nm);
    JSTSP_REQUIRE(W && O, JSTSP_E_NOMEM, "synthesize: workspace exhausted");
    JSTSP_TRY(gemm(ctx, 'N', 'N', N, G2, Gr, batch, Mat{A, strideA, N}, Mat{S, (long long)g, Gr}, W,
                   (long long)ng, N));
    if (h2) {
        // rows of A S (index n) and columns of B (index m) are not contracted: equilibrate them (see axis_exp_kernel)
        int32_t *eW = ctx->arena.get<int32_t>((size_t)batch * N), *eB = ctx->arena.get<int32_t>((size_t)nB * M);
        float2 *Bs = ctx->arena.get<float2>((size_t)nB * G2 * M);
        JSTSP_REQUIRE(eW && eB && Bs, JSTSP_E_NOMEM, "synthesize: workspace exhausted");
        const long long sBs = strideB ? (long long)G2 * M : 0;
        JSTSP_TRY(axis_exp(ctx, N, G2, W, (long long)ng, batch, 0, eW));
        JSTSP_TRY(scale2(ctx, N, G2, batch, W, (long long)ng, W, (long long)ng, eW, N, nullptr, 0, -1));
        JSTSP_TRY(axis_exp(ctx, G2, M, B, strideB, nB, 1, eB));
        JSTSP_TRY(scale2(ctx, G2, M, nB, B, strideB, Bs, sBs, nullptr, 0, eB, M, -1));
        // b(k = g, j = m) = B[g + G2 m] packed once; a = A S
        HPack pk;
        JSTSP_TRY(hgemm_pack(ctx, pk, ctx->arena, Bs, sBs, 1, G2, 0, G2, M, nB, (long long)G2 * M));
        uint32_t *amax = ctx->arena.get<uint32_t>(batch);
        JSTSP_REQUIRE(amax, JSTSP_E_NOMEM, "synthesize: workspace exhausted");
        JSTSP_TRY(hgemm_absmax(ctx, W, (long long)ng, (long long)ng, batch, amax));
        HGemmDesc hd{W, (long long)ng, N, amax, pk.data, strideB ? pk.st : 0, pk.bmax, strideB ? 1 : 0, pk.KS, pk.JT,
                     O, (long long)nm, N, N, M, G2, batch, EPI_NONE, nullptr, nullptr, nullptr};
        if (pair) JSTSP_TRY(pair_apack(ctx, W, (long long)ng, N, G2, batch, amax, pk, hd));
        JSTSP_TRY(launch_hgemm(ctx, hd, "synthesize"));
        JSTSP_TRY(scale2(ctx, N, M, batch, O, (long long)nm, O, (long long)nm, eW, N, eB, strideB ? M : 0, +1));
    } else
    JSTSP_TRY(gemm(ctx, 'N', 'N', N, M, G2, batch, Mat{W, (long long)ng, N}, Mat{B, strideB, G2}, O,
                   (long long)nm, N, 1.f, nullptr, 0, 0, 0.f, GEMM_SYNTH));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out), O, batch * nm, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_ls_c32(jstsp_ctx *ctx, int N, int M, int Gr, int G2, int batch, const jstsp_c32 *Y_, const jstsp_c32 *A_,
                 long long strideA, const jstsp_c32 *B_, long long strideB, jstsp_c32 *S_out, int memspace)
{
    // S_ls = pinv(A)*Y*pinv(B) (plot_errorVSsnr.m:83).  Per factor:
    //   fits the in-LDS float64 kernel (pinv.hip; every shape of the reference's drivers): SVD-based pinv as MATLAB's;
    //   larger: pinv(A) = (A^H A)^-1 A^H (N >= Gr), pinv(B) = B^H (B B^H)^-1 (M >= G2) through the fp32 Gram inverse
    //           (hinv.hip), conditioning recorded / checked.
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(Y_ && A_ && B_ && S_out, JSTSP_E_NULL, "ls: NULL array argument");
    JSTSP_REQUIRE(N > 0 && M > 0 && Gr > 0 && G2 > 0 && batch > 0, JSTSP_E_SHAPE, "ls: bad shape");
    const bool pA = pinv_fits(N, Gr), pB = pinv_fits(G2, M);
    JSTSP_REQUIRE((pA || N >= Gr) && (pB || M >= G2), JSTSP_E_UNSUPPORTED,
                  "ls: a factor too large for the float64 pinv kernel must have full rank (A: N >= Gr, B: M >= G2)");
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2, ng = (size_t)N * G2;
    const int nA = strideA ? batch : 1, nB = strideB ? batch : 1;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t szB = strideB ? (size_t)strideB * (batch - 1) + (size_t)G2 * M : (size_t)G2 * M;
    size_t need = 2 * rnd256(batch * ng * sizeof(float2)) + 3 * rnd256(batch * g * sizeof(float2));
    need += pA ? rnd256((size_t)nA * Gr * N * sizeof(float2))
               : 2 * rnd256((size_t)nA * Gr * Gr * sizeof(float2)) + hinv_bytes(Gr, nA);
    need += pB ? rnd256((size_t)nB * M * G2 * sizeof(float2))
               : 2 * rnd256((size_t)nB * G2 * G2 * sizeof(float2)) + hinv_bytes(G2, nB);
    if (memspace == JSTSP_HOST)
        need += rnd256(batch * nm * sizeof(float2)) + rnd256(szA * sizeof(float2)) + rnd256(szB * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    Arena &a = ctx->arena;
    const float2 *Y, *A, *B;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Y_), batch * nm, memspace, &Y));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), szA, memspace, &A));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(B_), szB, memspace, &B));
    JSTSP_TRY(diag_reset(ctx));
    float2 *Tc = a.get<float2>(batch * ng), *Tc2 = a.get<float2>(batch * ng), *R1 = a.get<float2>(batch * g),
           *S = a.get<float2>(batch * g);
    JSTSP_REQUIRE(Tc && Tc2 && R1 && S, JSTSP_E_NOMEM, "ls: workspace exhausted");
    const Mat Am{A, strideA, N}, Bm{B, strideB, G2};
    const long long sg = (long long)g, sng = (long long)ng;
    // ---- B side: Tc = Y pinv(B)   (N x G2)
    if (pB) {
        float2 *PB = a.get<float2>((size_t)nB * M * G2);
        JSTSP_REQUIRE(PB, JSTSP_E_NOMEM, "ls: workspace exhausted");
        JSTSP_TRY(launch_pinv(ctx, G2, M, nB, B, strideB, G2, PB, (long long)M * G2, M));
        JSTSP_TRY(gemm(ctx, 'N', 'N', N, G2, M, batch, Mat{Y, (long long)nm, N}, Mat{PB, strideB ? (long long)M * G2 : 0, M},
                       Tc, sng, N, 1.f, nullptr, 0, 0, 0.f, GEMM_CORRELATE));
    } else {
        float2 *GB = a.get<float2>((size_t)nB * G2 * G2), *GBi = a.get<float2>((size_t)nB * G2 * G2);
        JSTSP_REQUIRE(GB && GBi, JSTSP_E_NOMEM, "ls: workspace exhausted");
        JSTSP_TRY(gemm(ctx, 'N', 'C', G2, G2, M, nB, Bm, Bm, GB, (long long)G2 * G2, G2));
        const size_t mark = a.off;
        JSTSP_TRY(hermitian_inverse(ctx, G2, nB, GB, GBi));
        a.off = mark;
        JSTSP_TRY(gemm(ctx, 'N', 'C', N, G2, M, batch, Mat{Y, (long long)nm, N}, Bm, Tc2, sng, N, 1.f, nullptr, 0, 0, 0.f,
                       GEMM_CORRELATE));
        JSTSP_TRY(gemm(ctx, 'N', 'N', N, G2, G2, batch, Mat{Tc2, sng, N}, Mat{GBi, strideB ? (long long)G2 * G2 : 0, G2},
                       Tc, sng, N));
    }
    // ---- A side: S = pinv(A) Tc   (Gr x G2)
    if (pA) {
        float2 *PA = a.get<float2>((size_t)nA * Gr * N);
        JSTSP_REQUIRE(PA, JSTSP_E_NOMEM, "ls: workspace exhausted");
        JSTSP_TRY(launch_pinv(ctx, N, Gr, nA, A, strideA, N, PA, (long long)Gr * N, Gr));
        JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, N, batch, Mat{PA, strideA ? (long long)Gr * N : 0, Gr}, Mat{Tc, sng, N}, S, sg,
                       Gr));
    } else {
        float2 *GA = a.get<float2>((size_t)nA * Gr * Gr), *GAi = a.get<float2>((size_t)nA * Gr * Gr);
        JSTSP_REQUIRE(GA && GAi, JSTSP_E_NOMEM, "ls: workspace exhausted");
        JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, Gr, N, nA, Am, Am, GA, (long long)Gr * Gr, Gr));
        const size_t mark = a.off;
        JSTSP_TRY(hermitian_inverse(ctx, Gr, nA, GA, GAi));
        a.off = mark;
        JSTSP_TRY(gemm(ctx, 'C', 'N', Gr, G2, N, batch, Am, Mat{Tc, sng, N}, R1, sg, Gr));
        JSTSP_TRY(gemm(ctx, 'N', 'N', Gr, G2, Gr, batch, Mat{GAi, strideA ? (long long)Gr * Gr : 0, Gr}, Mat{R1, sg, Gr}, S,
                       sg, Gr));
    }
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(S_out), S, batch * g, memspace));
    if (memspace == JSTSP_HOST) {
        JSTSP_HIP(hipStreamSynchronize(ctx->stream));
        JSTSP_TRY(diag_check_host(ctx, "ls"));
    }
    return 0;
}

int jstsp_pinv_c32(jstsp_ctx *ctx, int rows, int cols, int batch, const jstsp_c32 *A_, jstsp_c32 *P_, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(A_ && P_, JSTSP_E_NULL, "pinv: NULL array argument");
    JSTSP_REQUIRE(rows > 0 && cols > 0 && batch > 0, JSTSP_E_SHAPE, "pinv: bad shape");
    JSTSP_REQUIRE(pinv_fits(rows, cols), JSTSP_E_UNSUPPORTED,
                  "pinv: %d x %d does not fit the in-LDS float64 kernel ((max*min + min^2) * 16 B <= 156 KiB)", rows, cols);
    const size_t n = (size_t)rows * cols;
    size_t need = rnd256(batch * n * sizeof(float2));
    if (memspace == JSTSP_HOST) need += rnd256(batch * n * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *A;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), batch * n, memspace, &A));
    float2 *P = memspace == JSTSP_DEVICE ? reinterpret_cast<float2 *>(P_) : ctx->arena.get<float2>(batch * n);
    JSTSP_REQUIRE(P, JSTSP_E_NOMEM, "pinv: workspace exhausted");
    JSTSP_TRY(diag_reset(ctx));
    JSTSP_TRY(launch_pinv(ctx, rows, cols, batch, A, (long long)n, rows, P, (long long)n, cols));
    if (memspace == JSTSP_HOST) {
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(P_), P, batch * n, memspace));
        JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int jstsp_svt_c32(jstsp_ctx *ctx, int Mr, int Mt, int batch, const jstsp_c32 *Y_, const double *tau,
                  jstsp_c32 *X_, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(Y_ && tau && X_, JSTSP_E_NULL, "svt: NULL argument");
    JSTSP_REQUIRE(Mr > 0 && Mt > 0 && batch > 0, JSTSP_E_SHAPE, "svt: bad shape");
    JSTSP_REQUIRE(std::min(Mr, Mt) <= 2048, JSTSP_E_UNSUPPORTED, "svt: min(Mr, Mt) = %d > 2048",
                  std::min(Mr, Mt));
    const size_t nm = (size_t)Mr * Mt;
    size_t need = GramWS::bytes(Mr, Mt, batch, true) + rnd256(batch * sizeof(TrialParams)) +
                  rnd256(batch * nm * sizeof(float2));
    if (memspace == JSTSP_HOST) need += rnd256(batch * nm * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *Y;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Y_), batch * nm, memspace, &Y));
    TrialParams *prm = ctx->arena.get<TrialParams>(batch);
    float2 *X = ctx->arena.get<float2>(batch * nm);
    JSTSP_REQUIRE(prm && X, JSTSP_E_NOMEM, "svt: workspace exhausted");
    GramWS w;
    JSTSP_TRY(w.alloc(ctx->arena, Mr, Mt, batch, true));
    JSTSP_TRY(upload_tau_rho(ctx, batch, tau, nullptr, prm));
    JSTSP_TRY(svt_batched(ctx, w, Y, prm, nullptr, X));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(X_), X, batch * nm, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_nmse_spectral_c32(jstsp_ctx *ctx, int R, int C, int batch, const jstsp_c32 *S_,
                            const jstsp_c32 *Zbar_, double *nmse, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(S_ && Zbar_ && nmse, JSTSP_E_NULL, "nmse: NULL argument");
    JSTSP_REQUIRE(R > 0 && C > 0 && batch > 0, JSTSP_E_SHAPE, "nmse: bad shape");
    JSTSP_REQUIRE(std::min(R, C) <= 2048, JSTSP_E_UNSUPPORTED, "nmse: min(R, C) = %d > 2048", std::min(R, C));
    const size_t n = (size_t)R * C;
    size_t need = GramWS::bytes(R, C, batch, false) + rnd256(batch * n * sizeof(float2)) +
                  2 * rnd256(batch * sizeof(float)) + rnd256(batch * sizeof(double));
    if (memspace == JSTSP_HOST) need += 2 * rnd256(batch * n * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *S, *Zb;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(S_), batch * n, memspace, &S));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Zbar_), batch * n, memspace, &Zb));
    float2 *D = ctx->arena.get<float2>(batch * n);
    float *num = ctx->arena.get<float>(batch), *den = ctx->arena.get<float>(batch);
    double *o = ctx->arena.get<double>(batch);
    JSTSP_REQUIRE(D && num && den && o, JSTSP_E_NOMEM, "nmse: workspace exhausted");
    GramWS w;
    JSTSP_TRY(w.alloc(ctx->arena, R, C, batch, false));
    const long long tot = (long long)batch * n;
    hipLaunchKernelGGL(diff_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 4096)), dim3(256), 0,
                       ctx->stream, tot, S, Zb, D);
    JSTSP_TRY(sigma_max_sq(ctx, w, D, num));
    JSTSP_TRY(sigma_max_sq(ctx, w, Zb, den));
    hipLaunchKernelGGL(ratio_cap_kernel, dim3((batch + 255) / 256), dim3(256), 0, ctx->stream, batch, num, den,
                       o, 1, 1ll, 0ll);
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, nmse, o, (size_t)batch, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_lambda_max_sequence_c32(jstsp_ctx *ctx, int n, int batch, int steps, const jstsp_c32 *G_, float *lam, int memspace)
{
    // lambda_max of a SEQUENCE of batches of Hermitian matrices, matrix t of step s warm-started from matrix t of step s - 1:
    // the kernel behind convergence_error(:,1:2) (proposed_algorithm.m:67,69) as the ADMM loops drive it, on the caller's matrices
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(G_ && lam, JSTSP_E_NULL, "lambda_max_sequence: NULL argument");
    JSTSP_REQUIRE(n > 0 && n <= 128 && batch > 0 && steps > 0, JSTSP_E_SHAPE, "lambda_max_sequence: 1 <= n <= 128, batch, steps > 0");
    const size_t nn = (size_t)n * n, tot = (size_t)steps * batch;
    size_t need = rnd256((size_t)batch * lanczos_ne(n) * sizeof(float2)) + rnd256((size_t)batch * sizeof(int)) + rnd256(tot * sizeof(float));
    if (memspace == JSTSP_HOST) need += rnd256(tot * nn * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *G;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(G_), tot * nn, memspace, &G));
    GramWS w;                                   // (only the warm-start record and the shape are used)
    w.n = n; w.batch = batch; w.nsplit = 1;
    w.lz.ne = lanczos_ne(n);
    w.lz.x = ctx->arena.get<float2>((size_t)batch * w.lz.ne);
    w.lz.state = ctx->arena.get<int>((size_t)batch);
    float *o = ctx->arena.get<float>(tot);
    JSTSP_REQUIRE(w.lz.x && w.lz.state && o, JSTSP_E_NOMEM, "lambda_max_sequence: workspace exhausted");
    JSTSP_TRY(lanczos_warm_reset(ctx, w));
    for (int s = 0; s < steps; ++s) {
        w.lz.call = s;
        JSTSP_TRY(launch_lmax(ctx, n, batch, G + (size_t)s * batch * nn, (long long)nn, 1, 0, o + (size_t)s * batch, true,
                              w.lz.mismatch ? &w.lz : nullptr, 0));
    }
    JSTSP_TRY(stage_out(ctx, lam, o, tot, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_rate_c32(jstsp_ctx *ctx, int R, int C, int batch, const jstsp_c32 *S_, const jstsp_c32 *Zbar_,
                   double noise_var, double *rate, int memspace)
{
    // log2(real(det(eye(Nr) + 1/Nr*Zbar*Zbar'*1/(noise_var + norm(Zbar-S)^2/norm(Zbar)^2))))  (plot_rateVSframelength.m:81)
    // = sum_i log2(1 + lambda_i(Zbar Zbar') / (Nr (noise_var + e))): the eigenvalues of the Gram of the smaller side
    // (det(I + c Z Z') = det(I + c Z' Z)); the NMSE e is NOT capped in that script.
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(S_ && Zbar_ && rate, JSTSP_E_NULL, "rate: NULL argument");
    JSTSP_REQUIRE(R > 0 && C > 0 && batch > 0, JSTSP_E_SHAPE, "rate: bad shape");
    JSTSP_REQUIRE(std::min(R, C) <= 128, JSTSP_E_UNSUPPORTED, "rate: min(R, C) = %d > 128", std::min(R, C));
    const size_t n = (size_t)R * C;
    const int ng = std::min(R, C), ne = (ng + 1) & ~1;
    size_t need = GramWS::bytes(R, C, batch, false) + rnd256(batch * n * sizeof(float2)) +
                  2 * rnd256(batch * sizeof(float)) + rnd256(batch * sizeof(double)) +
                  rnd256((size_t)batch * ng * ng * sizeof(float2)) + rnd256((size_t)batch * ng * sizeof(float)) +
                  rnd256((size_t)batch * ne * ne * sizeof(float2));
    if (memspace == JSTSP_HOST) need += 2 * rnd256(batch * n * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *S, *Zb;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(S_), batch * n, memspace, &S));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Zbar_), batch * n, memspace, &Zb));
    float2 *D = ctx->arena.get<float2>(batch * n);
    float *num = ctx->arena.get<float>(batch), *den = ctx->arena.get<float>(batch);
    double *o = ctx->arena.get<double>(batch);
    float2 *U = ctx->arena.get<float2>((size_t)batch * ng * ng);
    float *lam = ctx->arena.get<float>((size_t)batch * ng);
    float2 *Vg = eig_needs_global_v(ng) ? ctx->arena.get<float2>((size_t)batch * ne * ne) : nullptr;
    JSTSP_REQUIRE(D && num && den && o && U && lam && (!eig_needs_global_v(ng) || Vg), JSTSP_E_NOMEM, "rate: workspace exhausted");
    GramWS w;
    JSTSP_TRY(w.alloc(ctx->arena, R, C, batch, false));
    const long long tot = (long long)batch * n;
    hipLaunchKernelGGL(diff_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 4096)), dim3(256), 0,
                       ctx->stream, tot, S, Zb, D);
    JSTSP_TRY(sigma_max_sq(ctx, w, D, num));
    JSTSP_TRY(sigma_max_sq(ctx, w, Zb, den));                     // leaves the Gram partials of Zbar in the workspace
    const long long sG = (long long)w.n * w.n;
    JSTSP_TRY(launch_eig(ctx, EIG_VECS, w.n, batch, w.Gpart, sG * w.nsplit, w.nsplit, sG, nullptr, nullptr, U, lam, Vg));
    hipLaunchKernelGGL(rate_kernel, dim3((batch + 255) / 256), dim3(256), 0, ctx->stream, batch, w.n, R, lam, num, den,
                       noise_var, o);
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, rate, o, (size_t)batch, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_mc_svt_c32(jstsp_ctx *ctx, int Mr, int Mt, int batch, const jstsp_c32 *OH_, const float *Omega_,
                     int Imax, const double *tau, const double *rho, jstsp_c32 *X_out, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(OH_ && Omega_ && tau && rho && X_out, JSTSP_E_NULL, "mc_svt: NULL argument");
    JSTSP_REQUIRE(Mr > 0 && Mt > 0 && batch > 0 && Imax >= 0, JSTSP_E_SHAPE, "mc_svt: bad shape");
    JSTSP_REQUIRE(std::min(Mr, Mt) <= 2048, JSTSP_E_UNSUPPORTED, "mc_svt: min(Mr, Mt) = %d > 2048",
                  std::min(Mr, Mt));
    const size_t nm = (size_t)Mr * Mt;
    size_t need = GramWS::bytes(Mr, Mt, batch, true) + rnd256(batch * sizeof(TrialParams)) +
                  2 * rnd256(batch * nm * sizeof(float2));
    if (memspace == JSTSP_HOST) need += rnd256(batch * nm * sizeof(float2)) + rnd256(batch * nm * sizeof(float));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *OH;
    const float *Omega;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(OH_), batch * nm, memspace, &OH));
    JSTSP_TRY(stage_in(ctx, Omega_, batch * nm, memspace, &Omega));
    TrialParams *prm = ctx->arena.get<TrialParams>(batch);
    float2 *Y = ctx->arena.get<float2>(batch * nm), *X = ctx->arena.get<float2>(batch * nm);
    JSTSP_REQUIRE(prm && Y && X, JSTSP_E_NOMEM, "mc_svt: workspace exhausted");
    GramWS w;
    JSTSP_TRY(w.alloc(ctx->arena, Mr, Mt, batch, true));
    w.eig_stop = mc_eig_stop();
    JSTSP_TRY(upload_tau_rho(ctx, batch, tau, rho, prm));
    JSTSP_HIP(hipMemsetAsync(Y, 0, batch * nm * sizeof(float2), ctx->stream));     // mc_svt.m:5
    JSTSP_HIP(hipMemsetAsync(X, 0, batch * nm * sizeof(float2), ctx->stream));
    for (int it = 0; it < Imax; ++it) {                                              // :7
        JSTSP_TRY(svt_batched(ctx, w, Y, prm, nullptr, X, true));                    // :8
        hipLaunchKernelGGL(mc_svt_update_kernel, grid2((long long)nm, batch), dim3(256), 0, ctx->stream,
                           (long long)nm, Y, OH, Omega, X, prm);                     // :9
    }
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(X_out), X, batch * nm, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int jstsp_mc_admm_c32(jstsp_ctx *ctx, int Mr, int Mt, int batch, const jstsp_c32 *Htrue_,
                      const jstsp_c32 *OH_, const float *Omega_, int Imax, const double *tau,
                      const double *rho, jstsp_c32 *X_out, double *ce_out, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(OH_ && Omega_ && tau && rho && X_out, JSTSP_E_NULL, "mc_admm: NULL argument");
    JSTSP_REQUIRE(!ce_out || Htrue_, JSTSP_E_NULL, "mc_admm: convergence_error needs Htrue");
    JSTSP_REQUIRE(Mr > 0 && Mt > 0 && batch > 0 && Imax >= 0, JSTSP_E_SHAPE, "mc_admm: bad shape");
    JSTSP_REQUIRE(std::min(Mr, Mt) <= 2048, JSTSP_E_UNSUPPORTED, "mc_admm: min(Mr, Mt) = %d > 2048",
                  std::min(Mr, Mt));
    const size_t nm = (size_t)Mr * Mt;
    const bool want_ce = ce_out != nullptr;
    size_t need = GramWS::bytes(Mr, Mt, batch, true) + rnd256(batch * sizeof(TrialParams)) +
                  5 * rnd256(batch * nm * sizeof(float2)) + rnd256(batch * nm * sizeof(float)) +
                  2 * rnd256(batch * sizeof(float)) + rnd256((size_t)batch * std::max(Imax, 1) * sizeof(double));
    if (want_ce) need += GramWS::bytes(Mr, Mt, batch, false);
    if (memspace == JSTSP_HOST) need += 2 * rnd256(batch * nm * sizeof(float2)) + rnd256(batch * nm * sizeof(float));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *OH, *Htrue = nullptr;
    const float *Omega;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(OH_), batch * nm, memspace, &OH));
    JSTSP_TRY(stage_in(ctx, Omega_, batch * nm, memspace, &Omega));
    if (want_ce) JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Htrue_), batch * nm, memspace, &Htrue));
    TrialParams *prm = ctx->arena.get<TrialParams>(batch);
    float2 *X = ctx->arena.get<float2>(batch * nm), *Y = ctx->arena.get<float2>(batch * nm),
           *Z = ctx->arena.get<float2>(batch * nm), *Zn = ctx->arena.get<float2>(batch * nm),
           *D = ctx->arena.get<float2>(batch * nm);
    float *invD = ctx->arena.get<float>(batch * nm);
    float *num = ctx->arena.get<float>(batch), *den = ctx->arena.get<float>(batch);
    double *ce = ctx->arena.get<double>((size_t)batch * std::max(Imax, 1));
    JSTSP_REQUIRE(prm && X && Y && Z && Zn && D && invD && num && den && ce, JSTSP_E_NOMEM,
                  "mc_admm: workspace exhausted");
    GramWS w, wn;
    JSTSP_TRY(w.alloc(ctx->arena, Mr, Mt, batch, true));
    w.eig_stop = mc_eig_stop();
    if (want_ce) JSTSP_TRY(wn.alloc(ctx->arena, Mr, Mt, batch, false));
    JSTSP_TRY(upload_tau_rho(ctx, batch, tau, rho, prm));
    hipStream_t st = ctx->stream;
    JSTSP_HIP(hipMemsetAsync(X, 0, batch * nm * sizeof(float2), st));                // mc_admm.m:6-8
    JSTSP_HIP(hipMemsetAsync(Y, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(Z, 0, batch * nm * sizeof(float2), st));
    JSTSP_HIP(hipMemsetAsync(Zn, 0, batch * nm * sizeof(float2), st));
    JSTSP_TRY(launch_inv_d(ctx, (long long)nm, batch, Omega, 1.f, prm, invD));        // :11-17  A = diag(Omega) + rho I
    if (want_ce) JSTSP_TRY(sigma_max_sq(ctx, wn, Htrue, den));
    if (want_ce) JSTSP_TRY(lanczos_warm_reset(ctx, wn));       // the error curve's lambda_max, warm-started from iteration to iteration
    const long long tot = (long long)batch * nm;
    for (int it = 0; it < Imax; ++it) {                                               // :20
        JSTSP_TRY(svt_batched(ctx, w, Zn, prm, nullptr, X, true));                    // :22  X = svt(Y - Z/rho, tau/rho)
        hipLaunchKernelGGL(mc_admm_update_kernel, grid2((long long)nm, batch), dim3(256), 0, st, (long long)nm,
                           Y, Z, OH, invD, X, prm, Zn);                               // :24-26
        if (want_ce) {                                                                // :28
            hipLaunchKernelGGL(diff_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 4096)),
                               dim3(256), 0, st, tot, X, Htrue, D);
            JSTSP_TRY(sigma_max_sq(ctx, wn, D, num, true));
            hipLaunchKernelGGL(ratio_cap_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, batch, num, den,
                               ce, 0, (long long)Imax, (long long)it);
        }
    }
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(X_out), X, batch * nm, memspace));
    if (want_ce && Imax > 0) JSTSP_TRY(stage_out(ctx, ce_out, ce, (size_t)batch * Imax, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"

// Res = A' * Tc - RV,  P1 = GA * Res   (Gr x G2 per problem; N = Gr = 64, G2 a multiple of 64): the gradient step's 64-term
// products `K2'*k - R*v` (second factor) and the first factor of `R*res` (proposed_algorithm.m:47-48) as the solver runs them
// (csrc/hsmall.hip: three-way split-f16 operands, float64 final sums).  Tc: N x G2 x batch; A: N x Gr (strideA 0 = shared); GA: Gr x Gr
// Hermitian (strideG 0 = shared); RV: Gr x G2 x batch or NULL.
extern "C" int jstsp_gradient_head_c32(jstsp_ctx *ctx, int N, int Gr, int G2, int batch, const jstsp_c32 *Tc_, const jstsp_c32 *A_,
                                       long long strideA, const jstsp_c32 *GA_, long long strideG, const jstsp_c32 *RV_,
                                       jstsp_c32 *Res_out, jstsp_c32 *P1_out, int memspace)
{
    JSTSP_TRY(check_common(ctx, memspace));
    JSTSP_ENTER(ctx);
    JSTSP_REQUIRE(Tc_ && A_ && GA_ && Res_out && P1_out, JSTSP_E_NULL, "gradient_head: NULL array argument");
    JSTSP_REQUIRE(batch > 0 && grad_head_shape_ok(N, Gr, G2), JSTSP_E_UNSUPPORTED,
                  "gradient_head: N = %d, Gr = %d, G2 = %d (needs N = Gr = 64, G2 a multiple of 64)", N, Gr, G2);
    const size_t ng = (size_t)N * G2, g = (size_t)Gr * G2;
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t szG = strideG ? (size_t)strideG * (batch - 1) + (size_t)Gr * Gr : (size_t)Gr * Gr;
    size_t need = 2 * rnd256(batch * g * sizeof(float2));
    if (memspace == JSTSP_HOST)
        need += rnd256(batch * ng * sizeof(float2)) + rnd256(szA * sizeof(float2)) + rnd256(szG * sizeof(float2)) + rnd256(batch * g * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    const float2 *Tc, *A, *GA, *RV = nullptr;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Tc_), batch * ng, memspace, &Tc));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), szA, memspace, &A));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(GA_), szG, memspace, &GA));
    if (RV_) JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(RV_), batch * g, memspace, &RV));
    float2 *Res = ctx->arena.get<float2>(batch * g), *P1 = ctx->arena.get<float2>(batch * g);
    JSTSP_REQUIRE(Res && P1, JSTSP_E_NOMEM, "gradient_head: workspace exhausted");
    JSTSP_TRY(launch_grad_head(ctx, G2, batch, Tc, (long long)ng, 0, 1, nullptr, nullptr, 0, A, strideA, GA, strideG, RV, nullptr, Res, P1,
                               nullptr));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(Res_out), Res, batch * g, memspace));
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(P1_out), P1, batch * g, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
