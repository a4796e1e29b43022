// jstsp_mmv_omp_c32 — joint (simultaneous / MMV) orthogonal matching pursuit, the "OMP with MMV" baseline of the
// drivers and the second stage of their TSSR recipe:
//     s_omp_solver = spx.pursuit.joint.OrthogonalMatchingPursuit(A, numOfnz);
//     S_omp_mmv    = s_omp_solver.solve(Y_hbf_nr*pinv(B));            plot_errorVSsnr.m:116-117
//     S_tssr       = ... .solve(Y_svt*pinv(B));                        plot_errorVSsnr.m:158-162 (commented recipe)
// sparse-plex is NOT in the reference tree and no version is pinned (README.md:9): PARITY UNPINNED.  What is
// implemented is the published algorithm (Tropp, Gilbert, Strauss, "Algorithms for simultaneous sparse approximation",
// 2006; Chen & Huo 2006): all columns of Y share one support,
//     repeat K times:  g* = argmax_g || A(:,g)^H R ||_p  (p = 2 Chen-Huo, p = 1 Tropp's S-OMP; first index on ties),
//                      support += g*;  Z(support,:) = least squares of Y on A(:,support);  R = Y - A(:,support) Z,
// stopping early when every atom is in the support, when the new atom is numerically dependent on the chosen ones,
// or when ||R||_F <= 1e-6 ||Y||_F (with the drivers' numOfnz = 100 >= 32 atoms of a square A the result is the LS
// estimate pinv(A)*Y, which is what errorVSsnr_angles.fig shows: the LS and MMV-OMP curves coincide).
//
// One workgroup per problem; the least squares is carried incrementally by modified Gram-Schmidt of the selected atoms
// (float64 dot products), R is updated in place, Z comes from one back-substitution per column at the end.  The
// problems of this path are small (A 32 x 32 ... 128 x 128, tens to hundreds of columns): a latency-bound baseline,
// not a hot kernel.
#include "solver_common.h"
#include <algorithm>
#include <cstring>

namespace jstsp {

namespace {

__device__ __forceinline__ double wave_sum_d(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// workspace per problem (global): R N x S | Q N x K | Rt K x K | T K x S
__global__ __launch_bounds__(256) void mmv_omp_kernel(int N, int Gr, int S, int K, int pnorm, const float2 *A,
                                                      long long strideA, const float2 *Y, float2 *Rws, float2 *Qws,
                                                      float2 *Rtws, float2 *Tws, float2 *Z, int32_t *index_out,
                                                      int32_t *count_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *part = reinterpret_cast<float *>(smem_raw);            // [256] partial norms
    float *score = part + 256;                                    // [Gr]
    int *taken = reinterpret_cast<int *>(score + Gr);             // [Gr]
    double *red = reinterpret_cast<double *>(taken + Gr + (Gr & 1));   // [8]
    __shared__ int s_best, s_stop;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float2 *a = A + (long long)t * strideA;
    const float2 *y = Y + (long long)t * N * S;
    float2 *R = Rws + (long long)t * N * S, *Q = Qws + (long long)t * N * K, *Rt = Rtws + (long long)t * K * K,
           *T = Tws + (long long)t * K * S;
    float2 *z = Z + (long long)t * Gr * S;
    int32_t *io = index_out + (long long)t * K;

    auto block_sum = [&](double v) {
        v = wave_sum_d(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };

    double y2 = 0.0;
    for (int e = tid; e < N * S; e += 256) {
        const float2 v = y[e];
        R[e] = v;
        y2 += (double)v.x * v.x + (double)v.y * v.y;
    }
    for (int e = tid; e < Gr * S; e += 256) z[e] = make_float2(0.f, 0.f);
    for (int e = tid; e < K * K; e += 256) Rt[e] = make_float2(0.f, 0.f);
    for (int g = tid; g < Gr; g += 256) taken[g] = 0;
    for (int k = tid; k < K; k += 256) io[k] = 0;
    y2 = block_sum(y2);
    const int kmax = min(K, min(N, Gr));
    int k = 0;
    // deterministic split of the (atom, column) correlations: thread -> atom g = tid % gp, column group tid / gp
    int gp = 1;
    while (gp < Gr && gp < 256) gp <<= 1;                          // power of two >= Gr (<= 256)
    const int ncg = 256 / gp;                                     // column groups
    for (; k < kmax; ++k) {
        // ---- scores: || A(:,g)^H R ||_p over the S columns
        for (int g0 = 0; g0 < Gr; g0 += gp) {
            const int g = g0 + tid % gp, cg = tid / gp;
            float acc = 0.f;
            if (g < Gr && cg < ncg)
                for (int s = cg; s < S; s += ncg) {
                    float cx = 0.f, cy = 0.f;
                    const float2 *ag = a + (long long)N * g, *rs = R + (long long)N * s;
                    for (int i = 0; i < N; ++i) {
                        const float2 u = ag[i], v = rs[i];
                        cx += u.x * v.x + u.y * v.y;              // conj(a) r
                        cy += u.x * v.y - u.y * v.x;
                    }
                    acc += (pnorm == 1) ? sqrtf(cx * cx + cy * cy) : (cx * cx + cy * cy);
                }
            part[tid] = acc;
            __syncthreads();
            if (cg == 0 && g < Gr) {
                float sc = 0.f;
                for (int c = 0; c < ncg; ++c) sc += part[c * gp + (tid % gp)];     // fixed order: reproducible
                score[g] = taken[g] ? -1.f : sc;
            }
            __syncthreads();
        }
        if (tid == 0) {
            float best = -1.f;
            int bi = -1;
            for (int g = 0; g < Gr; ++g) {
                const float sc = score[g];
                if (sc == sc && sc > best) { best = sc; bi = g; }  // first index on ties
            }
            s_best = bi;
        }
        __syncthreads();
        const int gsel = s_best;
        if (gsel < 0) break;
        // ---- q_k = atom orthogonalised against q_0 .. q_{k-1} (modified Gram-Schmidt, two passes)
        float2 *q = Q + (long long)N * k;
        double n0 = 0.0;
        for (int i = tid; i < N; i += 256) {
            const float2 v = a[(long long)N * gsel + i];
            q[i] = v;
            n0 += (double)v.x * v.x + (double)v.y * v.y;
        }
        n0 = block_sum(n0);
        for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < k; ++j) {
                const float2 *qj = Q + (long long)N * j;
                double dx = 0.0, dy = 0.0;
                for (int i = tid; i < N; i += 256) {
                    const float2 u = qj[i], v = q[i];
                    dx += (double)u.x * v.x + (double)u.y * v.y;
                    dy += (double)u.x * v.y - (double)u.y * v.x;
                }
                dx = block_sum(dx);
                dy = block_sum(dy);
                for (int i = tid; i < N; i += 256) {
                    const float2 u = qj[i];
                    float2 v = q[i];
                    v.x -= (float)(dx * u.x - dy * u.y);
                    v.y -= (float)(dx * u.y + dy * u.x);
                    q[i] = v;
                }
                if (tid == 0) {
                    float2 r = Rt[j + (long long)K * k];
                    r.x += (float)dx; r.y += (float)dy;
                    Rt[j + (long long)K * k] = r;
                }
                __syncthreads();
            }
        double n1 = 0.0;
        for (int i = tid; i < N; i += 256) {
            const float2 v = q[i];
            n1 += (double)v.x * v.x + (double)v.y * v.y;
        }
        n1 = block_sum(n1);
        if (!(n1 > 1e-10 * n0) || !(n0 > 0.0)) break;             // atom numerically inside the span of the support
        const float inv = (float)(1.0 / sqrt(n1));
        for (int i = tid; i < N; i += 256) { float2 v = q[i]; v.x *= inv; v.y *= inv; q[i] = v; }
        if (tid == 0) { Rt[k + (long long)K * k] = make_float2((float)sqrt(n1), 0.f); io[k] = gsel + 1; taken[gsel] = 1; }
        __syncthreads();
        // ---- T(k,:) = q_k^H R;  R -= q_k T(k,:);  ||R||_F^2
        double r2 = 0.0;
        for (int s = tid; s < S; s += 256) {
            float2 *rs = R + (long long)N * s;
            double cx = 0.0, cy = 0.0;
            for (int i = 0; i < N; ++i) {
                const float2 u = q[i], v = rs[i];
                cx += (double)u.x * v.x + (double)u.y * v.y;
                cy += (double)u.x * v.y - (double)u.y * v.x;
            }
            const float fx = (float)cx, fy = (float)cy;
            T[k + (long long)K * s] = make_float2(fx, fy);
            for (int i = 0; i < N; ++i) {
                const float2 u = q[i];
                float2 v = rs[i];
                v.x -= fx * u.x - fy * u.y;
                v.y -= fx * u.y + fy * u.x;
                rs[i] = v;
                r2 += (double)v.x * v.x + (double)v.y * v.y;
            }
        }
        r2 = block_sum(r2);
        if (tid == 0) s_stop = (r2 <= 1e-12 * y2);
        __syncthreads();
        if (s_stop) { ++k; break; }
    }
    const int nsel = k;
    if (tid == 0) count_out[t] = nsel;
    __syncthreads();
    // ---- Z(support,:) = Rt^-1 T (back-substitution, one column per thread)
    for (int s = tid; s < S; s += 256) {
        for (int r = nsel - 1; r >= 0; --r) {
            float2 acc = T[r + (long long)K * s];
            for (int c = r + 1; c < nsel; ++c) {
                const float2 u = Rt[r + (long long)K * c];
                const float2 v = z[(io[c] - 1) + (long long)Gr * s];
                acc.x -= u.x * v.x - u.y * v.y;
                acc.y -= u.x * v.y + u.y * v.x;
            }
            const float d = Rt[r + (long long)K * r].x;
            z[(io[r] - 1) + (long long)Gr * s] = make_float2(acc.x / d, acc.y / d);
        }
    }
}

}  // namespace

}  // namespace jstsp

using namespace jstsp;

extern "C" int jstsp_mmv_omp_c32(jstsp_ctx *ctx, int N, int Gr, int S, int batch, const jstsp_c32 *A_, long long strideA,
                                 const jstsp_c32 *Y_, int K, int pnorm, jstsp_c32 *Z_out, int32_t *index_out,
                                 int32_t *count_out, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(A_ && Y_ && Z_out, JSTSP_E_NULL, "mmv_omp: NULL array argument");
    JSTSP_REQUIRE(N > 0 && Gr > 0 && S > 0 && batch > 0 && K > 0, JSTSP_E_SHAPE, "mmv_omp: bad shape");
    JSTSP_REQUIRE(Gr <= 4096, JSTSP_E_UNSUPPORTED, "mmv_omp: Gr = %d > 4096", Gr);
    JSTSP_REQUIRE(pnorm == 1 || pnorm == 2, JSTSP_E_ARG, "mmv_omp: pnorm must be 1 or 2");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_REQUIRE(strideA == 0 || strideA >= (long long)N * Gr, JSTSP_E_SHAPE, "strideA too small");
    JSTSP_ENTER(ctx);
    const int Kc = std::min(K, std::min(N, Gr));                   // at most min(N, Gr) independent atoms
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)N * Gr : (size_t)N * Gr;
    const size_t ns = (size_t)N * S, gs = (size_t)Gr * S;
    size_t need = rnd256(batch * ns * sizeof(float2)) + rnd256((size_t)batch * N * Kc * sizeof(float2)) +
                  rnd256((size_t)batch * Kc * Kc * sizeof(float2)) + rnd256((size_t)batch * Kc * S * sizeof(float2)) +
                  rnd256(batch * gs * sizeof(float2)) + rnd256((size_t)batch * Kc * sizeof(int32_t)) +
                  rnd256((size_t)batch * sizeof(int32_t));
    if (memspace == JSTSP_HOST) need += rnd256(szA * sizeof(float2)) + rnd256(batch * ns * sizeof(float2));
    JSTSP_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    Arena &ar = ctx->arena;
    const float2 *A, *Y;
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(A_), szA, memspace, &A));
    JSTSP_TRY(stage_in(ctx, reinterpret_cast<const float2 *>(Y_), batch * ns, memspace, &Y));
    float2 *R = ar.get<float2>(batch * ns), *Q = ar.get<float2>((size_t)batch * N * Kc),
           *Rt = ar.get<float2>((size_t)batch * Kc * Kc), *T = ar.get<float2>((size_t)batch * Kc * S),
           *Z = ar.get<float2>(batch * gs);
    int32_t *io = ar.get<int32_t>((size_t)batch * Kc), *cnt = ar.get<int32_t>(batch);
    JSTSP_REQUIRE(R && Q && Rt && T && Z && io && cnt, JSTSP_E_NOMEM, "mmv_omp: workspace exhausted");
    const size_t sh = 256 * sizeof(float) + (size_t)Gr * (sizeof(float) + sizeof(int)) + 16 + 8 * sizeof(double);
    JSTSP_HIP(hipFuncSetAttribute((const void *)mmv_omp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(mmv_omp_kernel, dim3(batch), dim3(256), sh, ctx->stream, N, Gr, S, Kc, pnorm, A, strideA, Y, R, Q,
                       Rt, T, Z, io, cnt);
    JSTSP_HIP(hipGetLastError());
    JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(Z_out), Z, batch * gs, memspace));
    if (index_out) {
        // the caller's array has K entries per problem; entries beyond the count are 0
        if (Kc == K) JSTSP_TRY(stage_out(ctx, index_out, io, (size_t)batch * K, memspace));
        else {
            if (memspace == JSTSP_DEVICE) JSTSP_HIP(hipMemsetAsync(index_out, 0, (size_t)batch * K * sizeof(int32_t), ctx->stream));
            else memset(index_out, 0, (size_t)batch * K * sizeof(int32_t));
            JSTSP_HIP(hipMemcpy2DAsync(index_out, (size_t)K * sizeof(int32_t), io, (size_t)Kc * sizeof(int32_t),
                                       (size_t)Kc * sizeof(int32_t), batch,
                                       memspace == JSTSP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    if (count_out) JSTSP_TRY(stage_out(ctx, count_out, cnt, (size_t)batch, memspace));
    if (memspace == JSTSP_HOST) JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
