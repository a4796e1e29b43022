// SVT projector of Gram matrices of order 65..128 (the svt of benchmark_algorithms/svt.m:5-10 on 128 x 128 inputs:
// BASELINE configs[2]) — parallel-order two-sided Jacobi, one 1024-thread workgroup per matrix.
//
// Two 128 x 128 complex fp32 matrices (G and the eigenvector basis U, 128 KiB each) do not fit in the 160 KiB of LDS
// together; the general kernel (eig.hip) keeps U in HBM and spends 22 us per round on its round trips.  Here
//   * G lives in LDS (128 x 129 float2) and is rotated in place by index, four 2x2 blocks per thread and round,
//     exactly as jacobi2_kernel (eig2.hip) does for orders <= 64;
//   * U lives in REGISTERS, as a systolic array mapped to lanes: row r of U is spread over 8 consecutive lanes, each
//     holding 8 of the 64 column PAIR-SLOTS of the round-robin ordering (16 complex registers).  The pair in slot P is
//     always (top[P], bottom[P]); after every round the columns move one slot around the circle — top slots shift
//     left, bottom slots right, the ends wrap — which is a register rename inside a lane and one DPP row shift between
//     neighbouring lanes.  No LDS or HBM traffic for U at all.
// A sweep is 127 rounds, after which every column is back in its initial slot.  Q = U diag(q) U^H is one in-LDS MFMA
// product at the end (U is written over G, which is no longer needed).
#include "common.h"
#include <cstdlib>

namespace jstsp {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NE = 128, LD = NE + 1, H = NE / 2, NT = 1024;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }  // conj(a) b

// column held by the top / bottom slot P in round s (circle method, n = 128): pair 0 is (127, s)
__device__ __forceinline__ int col_top(int P, int s) { if (P == 0) return NE - 1; int a = s + P; return a >= NE - 1 ? a - (NE - 1) : a; }
__device__ __forceinline__ int col_bot(int P, int s) { int b = s - P; return b < 0 ? b + (NE - 1) : b; }

template <int CTRL> __device__ __forceinline__ float dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

__global__ __launch_bounds__(NT) void jacobi128_kernel(int n, const float2 *Gpart, long long sGt, int nsplit, long long sGs,
                                                        const TrialParams *prm, const float *tau, float2 *Q, float conv_tol,
                                                        int max_sweeps, int *sweep_stat, float2 *Uwarm, int warm, float pre_tol)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2 *G = reinterpret_cast<float2 *>(smem_raw);           // [NE][LD] column-major; later U [NE][NE]
    float *rot = reinterpret_cast<float *>(G + NE * LD);        // [H][4]: c, wx, wy, bits (p | q << 8 | top_is_p << 16)
    float *red = rot + 4 * H;                                   // [24]
    float *qv = red + 24;                                       // [NE]
    const int t = blockIdx.x, tid = threadIdx.x;
    // (latency-bound: when the block Jacobi of eig_large.hip runs its panel products beside this kernel, their waves must not
    //  take issue slots from it)
    __builtin_amdgcn_s_setprio(3);

    for (int e = tid; e < NE * NE; e += NT) {
        const int i = e % NE, j = e / NE;
        float2 g = make_float2(0.f, 0.f);
        if (i < n && j < n) {
            const float2 *src = Gpart + (long long)t * sGt + i + (long long)n * j;
            for (int s = 0; s < nsplit; ++s) {
                const float2 v = src[(long long)s * sGs];
                g.x += v.x; g.y += v.y;
            }
        }
        G[i + LD * j] = g;
    }
    __syncthreads();
    // all-zero Gram = all-zero svt argument: output 0, i.e. Q = I (guard of svt.m:7-12, see eig2.hip)
    if (tid == 0) red[2] = 0.f;
    __syncthreads();
    for (int i = tid; i < n; i += NT)
        if (G[i + LD * i].x != 0.f) red[2] = 1.f;
    __syncthreads();
    if (red[2] == 0.f) {
        if (Q) for (int e = tid; e < n * n; e += NT) Q[(size_t)t * n * n + e] = make_float2((e % n == e / n) ? 1.f : 0.f, 0.f);
        if (Uwarm && !warm)                                     // the caller treats the basis as valid from now on
            for (int e = tid; e < n * n; e += NT)
                Uwarm[(size_t)t * n * n + e] = make_float2((e % n == e / n) ? 1.f : 0.f, 0.f);
        return;
    }
    for (int e = tid; e < NE * NE; e += NT) {                   // Hermitian part
        const int i = e % NE, j = e / NE;
        if (i < j) {
            const float2 u = G[i + LD * j], l = G[j + LD * i];
            const float2 a = make_float2(0.5f * (u.x + l.x), 0.5f * (u.y - l.y));
            G[i + LD * j] = a;
            G[j + LD * i] = make_float2(a.x, -a.y);
        } else if (i == j) G[i + LD * i].y = 0.f;
    }
    __syncthreads();
    {
        float m = 0.f;
        for (int i = tid; i < NE; i += NT) m = fmaxf(m, fabsf(G[i + LD * i].x));
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = m;
        __syncthreads();
        if (tid == 0) { float mm = red[4]; for (int wv = 1; wv < NT / 64; ++wv) mm = fmaxf(mm, red[4 + wv]); red[1] = mm; }
        __syncthreads();
    }
    const float dmax = red[1];

    // ---- U in the slot layout: thread (row ur, lane group uj) holds slots 8 uj .. 8 uj + 7.  Identity, or (warm
    //      start: the caller has already replaced G by Uw^H G Uw) the n x n basis of the previous call
    const int ur = tid >> 3, uj = tid & 7;
    float2 *Uw = Uwarm ? Uwarm + (size_t)t * n * n : nullptr;
    float2 top[8], bot[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int P = 8 * uj + k;
        const int ct = col_top(P, 0), cb = col_bot(P, 0);
        top[k] = (warm && Uw && ur < n && ct < n) ? Uw[ur + (size_t)n * ct] : make_float2(ur == ct ? 1.f : 0.f, 0.f);
        bot[k] = (warm && Uw && ur < n && cb < n) ? Uw[ur + (size_t)n * cb] : make_float2(ur == cb ? 1.f : 0.f, 0.f);
    }

    // this thread's blocks (a <= b) of the G update: item k of the 2080 upper-triangle slot pairs, k = tid, tid + 1024, tid + 2048
    int wa[3], wb[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int k = tid + NT * it;
        wa[it] = -1; wb[it] = -1;
        if (k < H * (H + 1) / 2) {
            // column b holds b + 1 items (a = 0..b): the largest b with b (b + 1) / 2 <= k
            int b = (int)((sqrtf(8.f * (float)k + 1.f) - 1.f) * 0.5f);
            while (b * (b + 1) / 2 > k) --b;
            while ((b + 1) * (b + 2) / 2 <= k) ++b;
            wa[it] = k - b * (b + 1) / 2; wb[it] = b;
        }
    }
    int sweeps_done = 0;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        // The rule at the end of a sweep knows the level the sweep STARTED from, so it confirms convergence with one sweep more
        // than the result needs (a sweep that starts below 1e-4 leaves 1e-8).  One pass over the off-diagonal entries (a
        // hundredth of a sweep) measures the matrix AS IT STANDS: below pre_tol no further sweep is run (GramWS::eig_stop: the
        // warm-started calls of mc_svt / mc_admm, one sweep instead of two).
        if (pre_tol > 0.f) {
            if (tid == 0) red[0] = 0.f;
            __syncthreads();
            float cw = 0.f;
            for (int e = tid; e < NE * NE; e += NT) {
                const int p = e % NE, q = e / NE;
                if (p >= q) continue;
                const float a = G[p + LD * p].x, dd = G[q + LD * q].x;
                const float2 bq = G[p + LD * q];
                const float ab = sqrtf(bq.x * bq.x + bq.y * bq.y);
                const float scale = sqrtf(fabsf(a) * fabsf(dd));
                if (ab > 0.f && ab > 1e-8f * scale) cw = fmaxf(cw, ab / fmaxf(scale, 1e-3f * dmax));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cw = fmaxf(cw, __shfl_xor(cw, o));
            if ((tid & 63) == 0) atomicMax(reinterpret_cast<int *>(&red[0]), __float_as_int(cw));
            __syncthreads();
            const float w0 = red[0];
            __syncthreads();
            if (w0 < pre_tol) break;
        }
        ++sweeps_done;
        if (tid == 0) red[0] = 0.f;
        float worst = 0.f;
        for (int s = 0; s < NE - 1; ++s) {
            // -- rotation of each of the H disjoint pairs of this round
            if (tid < H) {
                const int a0 = col_top(tid, s), b0 = col_bot(tid, s);
                const int p = min(a0, b0), q = max(a0, b0);
                const float a = G[p + LD * p].x, dd = G[q + LD * q].x;
                const float2 bq = G[p + LD * q];
                const float ab = sqrtf(bq.x * bq.x + bq.y * bq.y);
                float c = 1.f, wx = 0.f, wy = 0.f;
                const float scale = sqrtf(fabsf(a) * fabsf(dd));
                if (ab > 0.f && ab > 1e-8f * scale) {
                    worst = fmaxf(worst, ab / fmaxf(scale, 1e-3f * dmax));
                    const float zeta = (dd - a) / (2.f * ab);
                    const float tt = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                    c = 1.f / sqrtf(1.f + tt * tt);
                    const float sn = tt * c;
                    wx = sn * bq.x / ab;
                    wy = sn * bq.y / ab;
                }
                rot[4 * tid + 0] = c; rot[4 * tid + 1] = wx; rot[4 * tid + 2] = wy;
                rot[4 * tid + 3] = __int_as_float(p | (q << 8) | ((a0 < b0 ? 1 : 0) << 16));
            }
            __syncthreads();
            // -- G <- J^H G J on 2x2 blocks (pair a rows, pair b columns).  G stays Hermitian, so only the blocks with a <= b are
            //    computed (2080 of 4096: two or three per thread, their slot pairs fixed for the whole kernel: wa[], wb[]) and
            //    block (b, a) is written as the conjugate transpose of block (a, b) - half the rotation arithmetic of a round
            //    (the kernel is bound by VALU issue, HISTORY.md round 5) and G exactly Hermitian at every step.
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int a = wa[it], b = wb[it];
                if (a < 0) continue;
                const float4 ra = *reinterpret_cast<const float4 *>(&rot[4 * a]);
                const float4 rb = *reinterpret_cast<const float4 *>(&rot[4 * b]);
                const int pa = __float_as_int(ra.w) & 0xff, qa = (__float_as_int(ra.w) >> 8) & 0xff;
                const int pb = __float_as_int(rb.w) & 0xff, qb = (__float_as_int(rb.w) >> 8) & 0xff;
                const float ca = ra.x, cb = rb.x;
                const float2 wa_ = make_float2(ra.y, ra.z), wb_ = make_float2(rb.y, rb.z);
                const bool ida = (ra.y == 0.f && ra.z == 0.f), idb = (rb.y == 0.f && rb.z == 0.f);
                if (ida && idb) continue;
                float2 gpp = G[pa + LD * pb], gpq = G[pa + LD * qb];
                float2 gqp = G[qa + LD * pb], gqq = G[qa + LD * qb];
                // right: [x_p, x_q] -> [c x_p - conj(w) x_q, w x_p + c x_q]   (columns pb, qb)
                float2 t0 = cmulc(wb_, gpq), t1 = cmul(wb_, gpp);
                float2 n_pp = make_float2(cb * gpp.x - t0.x, cb * gpp.y - t0.y);
                float2 n_pq = make_float2(t1.x + cb * gpq.x, t1.y + cb * gpq.y);
                t0 = cmulc(wb_, gqq); t1 = cmul(wb_, gqp);
                float2 n_qp = make_float2(cb * gqp.x - t0.x, cb * gqp.y - t0.y);
                float2 n_qq = make_float2(t1.x + cb * gqq.x, t1.y + cb * gqq.y);
                // left: [y_p; y_q] -> [c y_p - w y_q; conj(w) y_p + c y_q]     (rows pa, qa)
                t0 = cmul(wa_, n_qp); t1 = cmulc(wa_, n_pp);
                gpp = make_float2(ca * n_pp.x - t0.x, ca * n_pp.y - t0.y);
                gqp = make_float2(t1.x + ca * n_qp.x, t1.y + ca * n_qp.y);
                t0 = cmul(wa_, n_qq); t1 = cmulc(wa_, n_pq);
                gpq = make_float2(ca * n_pq.x - t0.x, ca * n_pq.y - t0.y);
                gqq = make_float2(t1.x + ca * n_qq.x, t1.y + ca * n_qq.y);
                if (a == b) {           // the annihilated block: exact zeros off the diagonal, real diagonal
                    gpq = make_float2(0.f, 0.f); gqp = make_float2(0.f, 0.f);
                    gpp.y = 0.f; gqq.y = 0.f;
                } else {                // block (b, a) = (block (a, b))^H
                    G[pb + LD * pa] = make_float2(gpp.x, -gpp.y); G[qb + LD * pa] = make_float2(gpq.x, -gpq.y);
                    G[pb + LD * qa] = make_float2(gqp.x, -gqp.y); G[qb + LD * qa] = make_float2(gqq.x, -gqq.y);
                }
                G[pa + LD * pb] = gpp; G[pa + LD * qb] = gpq;
                G[qa + LD * pb] = gqp; G[qa + LD * qb] = gqq;
            }
            // -- U <- U J on this thread's eight pair-slots (registers)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 r4 = *reinterpret_cast<const float4 *>(&rot[4 * (8 * uj + k)]);
                if (r4.y == 0.f && r4.z == 0.f) continue;
                const bool top_is_p = (__float_as_int(r4.w) >> 16) & 1;
                const float c = r4.x;
                const float2 w = make_float2(r4.y, r4.z);
                const float2 xp = top_is_p ? top[k] : bot[k], xq = top_is_p ? bot[k] : top[k];
                const float2 t0 = cmulc(w, xq), t1 = cmul(w, xp);
                const float2 np_ = make_float2(c * xp.x - t0.x, c * xp.y - t0.y);
                const float2 nq_ = make_float2(t1.x + c * xq.x, t1.y + c * xq.y);
                top[k] = top_is_p ? np_ : nq_;
                bot[k] = top_is_p ? nq_ : np_;
            }
            // -- the columns move one slot around the circle: top slots shift left (slot 0 is fixed), bottom slots
            //    shift right, top slot 1 drops to bottom slot 0 and bottom slot 63 rises to top slot 63
            {
                const float2 t_from_next = make_float2(dpp<0x101>(top[0].x), dpp<0x101>(top[0].y));   // row_shl:1: lane + 1
                const float2 b_from_prev = make_float2(dpp<0x111>(bot[7].x), dpp<0x111>(bot[7].y));   // row_shr:1: lane - 1
                const float2 old_top1 = top[1], old_bot7 = bot[7], old_top0 = top[0];
#pragma unroll
                for (int k = 7; k > 0; --k) bot[k] = bot[k - 1];
                bot[0] = (uj == 0) ? old_top1 : b_from_prev;
#pragma unroll
                for (int k = 0; k < 7; ++k) top[k] = top[k + 1];
                top[7] = (uj == 7) ? old_bot7 : t_from_next;
                if (uj == 0) top[0] = old_top0;              // slot 0 never moves
            }
            __syncthreads();
        }
        if (tid < H) atomicMax(reinterpret_cast<int *>(&red[0]), __float_as_int(worst));
        __syncthreads();
        const float w = red[0];
        __syncthreads();
        if (w < conv_tol) break;
    }
    if (sweep_stat && tid == 0) atomicAdd(sweep_stat, sweeps_done);

    // ---- q_i = min(1, tau / sigma_i);  U (registers, initial slot layout again) -> LDS over G;  Q = U diag(q) U^H
    const float tv = tau ? tau[t] : prm[t].tauY_rho;
    for (int i = tid; i < NE; i += NT) {
        const float sig = sqrtf(fmaxf(G[i + LD * i].x, 0.f));
        qv[i] = (sig > 0.f) ? fminf(1.f, tv / sig) : 1.f;
    }
    __syncthreads();
    float2 *U = G;                                              // [NE][NE], U[r + NE * col]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int P = 8 * uj + k;
        U[ur + NE * col_top(P, 0)] = top[k];
        U[ur + NE * col_bot(P, 0)] = bot[k];
    }
    __syncthreads();
    if (Uw)                                                     // keep the basis for the next call of the sequence
        for (int e = tid; e < n * n; e += NT) Uw[e] = U[(e % n) + NE * (e / n)];
    if (!Q) return;                                             // (basis only: the block Jacobi of eig_large.hip)
    // Q[i][j] = sum_k q_k U[i][k] conj(U[j][k]): 16 blocks of 32 x 32, one per wave; MFMA fed (A-op = conj(U_j), B-op = q U_i)
    {
        const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
        const int i0 = (wave & 3) * 32, j0 = (wave >> 2) * 32;
        f32x16 re, im;
#pragma unroll
        for (int r = 0; r < 16; ++r) { re[r] = 0.f; im[r] = 0.f; }
#pragma unroll 4
        for (int kp = 0; kp < NE / 2; ++kp) {
            const int k = 2 * kp + lhi;
            const float qk = qv[k];
            float2 av = U[(i0 + l31) + NE * k];
            float2 bv = U[(j0 + l31) + NE * k];
            av.x *= qk; av.y *= qk;
            bv.y = -bv.y;
            re = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.x, re, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f32_32x32x2f32(-bv.y, av.y, re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.y, im, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, av.x, im, 0, 0, 0);
        }
        const int i = i0 + l31;
        float2 *Qt = Q + (size_t)t * n * n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (i < n && j < n) Qt[i + (size_t)n * j] = make_float2(re[r], im[r]);
        }
    }
}

}  // namespace

// SVT projector Q[t] = U diag(min(1, tau_t / sigma_i)) U^H of the n x n Gram sum_s Gpart[t][s], 64 < n <= 128.
// Q: nullptr = basis only (Uwarm required).  Uwarm: nullptr, or batch * n*n float2 that receives the eigenvector basis; warm != 0: Uwarm holds the basis of the
// previous call AND the caller has already transformed the Gram to that basis (G <- Uw^H G Uw).
int launch_eig128(jstsp_ctx *ctx, int n, int batch, const float2 *Gpart, long long sGt, int nsplit, long long sGs,
                  const TrialParams *prm, const float *tau, float2 *Q, float2 *Uwarm, int warm, int max_sweeps, float stop_level)
{
    JSTSP_REQUIRE(n > 64 && n <= 128, JSTSP_E_UNSUPPORTED, "launch_eig128: n = %d outside (64, 128]", n);
    JSTSP_REQUIRE(Q || Uwarm, JSTSP_E_ARG, "launch_eig128: neither a projector nor a basis requested");
    const size_t sh = (size_t)NE * LD * sizeof(float2) + (size_t)(4 * H + 24 + NE) * sizeof(float);
    JSTSP_HIP(hipFuncSetAttribute((const void *)jacobi128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    const float tol = 1e-4f;
    const float pre_tol = stop_level;       // level of the matrix as it stands below which no further sweep is run (0: off)
    const int maxsw = max_sweeps < 1 ? 1 : (max_sweeps > 16 ? 16 : max_sweeps);
    hipLaunchKernelGGL(jacobi128_kernel, dim3(batch), dim3(NT), sh, ctx->stream, n, Gpart, sGt, nsplit, sGs, prm, tau, Q,
                       tol, maxsw, (int *)nullptr, Uwarm, warm, pre_tol);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
