// Batched complex-fp32 GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel serves every dense contraction of the solver path — the dictionary
// correlation A^H K B^H (proposed_algorithm.m:47 `K2'*k`, OMP.m:17 `A'*r`), the synthesis
// A S B (:38,:58), the Gram applies (A^H A) V (B B^H) (:47-48 `R*v`) and the SVT's Gram /
// re-projection (svt.m:5-10) — through generic element strides, so no operand is ever
// transposed or conjugated in HBM.
//
// Tiling (wave64): a 256-thread workgroup owns a 64 x BN complex output tile; its 4 waves
// form a 2 x 2 grid, each wave a 32 x (BN/2) tile = BN/64 MFMA blocks of 32 x 32, with
// separate re/im accumulators (4 real MFMAs per complex block per k-pair).  Operand panels
// are staged global -> registers -> LDS, k-major with the non-contracted index contiguous
// (row pitch padded by one element so the transposing store of a k-contiguous source is
// bank-conflict free); the next panel's global loads are issued before the MFMAs of the
// current one and written to the other LDS buffer afterwards (one barrier per k-step).
// The MFMA is fed (A-op = b-panel, B-op = a-panel) so that the 32 lanes of a half-wave
// hold 32 consecutive rows i of one output column j: stores are 256-B contiguous in the
// column-major output.
//
// Workgroup -> tile mapping is XCD-aware: block b runs on XCD b % 8 (observed dispatch
// order), so problem t = 8*q + (b % 8) keeps all tiles of one Monte-Carlo trial — which
// share the a-panel (K or A*S) — on one XCD's L2.
#include "common.h"
#include <cstdlib>

namespace jstsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64;
constexpr int BK = 16;
#ifndef JSTSP_M64_STEPS
#define JSTSP_M64_STEPS 2
#endif
constexpr int M64_STEPS = JSTSP_M64_STEPS;   // fp32 chain length = 16 * M64_STEPS terms between fp64 flushes

// TAG only gives the hot call sites their own kernel symbol (identical code), so that
// rocprofv3's per-kernel statistics and the roofline in bench.py refer to one shape each.
//
// M64 (long contractions): the fp32 MFMA accumulates as one k-ordered fmaf chain, whose
// rounding error grows ~ eps*k.  Over the 4096-term correlation that noise, re-injected
// every ADMM iteration through res = K2'k - R v (a difference of two nearly equal terms),
// is what limits NMSE parity with the float64 reference.  With M64 the fp32 accumulators
// are flushed into fp64 master accumulators every 32 k (two panels) and restarted, so the
// chain length is 32 and the long sum is carried in double — on the VALU, next to the MFMAs.
//
// M3: three real MFMAs per complex block instead of four (Gauss / "3M"):
//   P1 = sum br*ar, P2 = sum bi*ai, P3 = sum (br+bi)(ar+ai);  re = P1 - P2, im = P3 - P1 - P2.
// The fp32 matrix pipe on gfx950 is issue-throttled by data-dependent power (profiles/
// r01_gemm_ladder_ubench.txt), so time follows the MFMA count: -25 %.  The operand sums are one
// VALU add per fragment; the imaginary part carries a ~2-3x larger rounding error (difference of
// larger sums), which is why M3 is combined with the fp64 master accumulators on the long chain.
//
// EPI: the element-wise ADMM updates that consume a GEMM result are applied to the accumulator
// tile before it leaves the registers (one pass over the N x M state instead of a GEMM store
// plus a separate streaming kernel): EPI_UPDATE_C after the synthesis, EPI_UPDATE_X after the
// SVT re-projection.
template <int BN, int TAG, bool M64, bool M3, int EPI>
__global__ __launch_bounds__(256) void cgemm_kernel(GemmDesc d, int tiles_m, int tiles_n)
{
    constexpr int LDA = BM + 1;
    constexpr int LDB = BN + 1;
    constexpr int NB = BN / 64;            // 32-wide MFMA blocks per wave along j
    constexpr int NLA = BM * BK / 256;     // a-panel elements per thread per k-step
    constexpr int NLB = BN * BK / 256;
    __shared__ float2 smem[2 * BK * (LDA + LDB)];
    float2 *sA = smem;                     // [2][BK*LDA]
    float2 *sB = smem + 2 * BK * LDA;      // [2][BK*LDB]

    // ---- decode (problem, tile, split) with the XCD-aware order -------------------------
    const int tiles = tiles_m * tiles_n * d.splitk;
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int slot = b >> 3;
    const int t = (slot / tiles) * 8 + xcd;
    if (t >= d.batch) return;
    int rem = slot % tiles;
    const int split = rem % d.splitk; rem /= d.splitk;
    const int tm = rem % tiles_m;
    const int tn = rem / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    if (d.herm_upper && n0 + BN <= m0) return;          // entirely below the diagonal: mirrored by the caller

    const int kchunk = ((d.k + d.splitk - 1) / d.splitk + BK - 1) / BK * BK;
    const int kbeg = split * kchunk;
    const int kend = min(d.k, kbeg + kchunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;

    const float2 *Ap = d.A + (long long)t * d.sAt;
    const float2 *Bp = d.B + (long long)t * d.sBt;
    const bool a_icont = (d.sAi == 1);     // a-panel source contiguous along i (else along k)
    const bool b_jcont = (d.sBj == 1);
    const float sgnA = d.conjA ? -1.f : 1.f;
    const float sgnB = d.conjB ? -1.f : 1.f;

    // Per-thread panel coordinates are fixed for the whole k loop: element pointers advance by
    // one panel per step (no 64-bit index arithmetic inside the loop), rows/columns outside the
    // matrix are redirected to element 0 and zeroed by a select, so every load is unconditional
    // (a predicated load makes hipcc branch around it and serialises the panel fetch).
    float2 ra[NLA], rb[NLB];
    float2 rb2[(EPI == EPI_UPDATE_X) ? NLB : 1];        // second b source (EPI_UPDATE_X: Z = X - V1/rho formed here)
    const bool two_b = (EPI == EPI_UPDATE_X) && d.B2 != nullptr;
    const long long boff2 = two_b ? (d.B2 - d.B) : 0;   // same strides: element e of B2 sits boff2 elements after B's
    const float bir = two_b ? d.prm[t].irho : 0.f, birl = two_b ? d.prm[t].irho_lo : 0.f;     // (1/rho as two floats: common.h)
    const float2 *pa[NLA], *pb[NLB];
    int la[NLA], lb[NLB], ka[NLA], kb[NLB];
    bool va[NLA], vb[NLB];
#pragma unroll
    for (int p = 0; p < NLA; ++p) {
        const int e = tid + 256 * p;
        const int i = a_icont ? (e % BM) : (e / BK);
        ka[p] = a_icont ? (e / BM) : (e % BK);
        const int gi = m0 + i;
        va[p] = gi < d.m;
        pa[p] = Ap + (long long)(va[p] ? gi : 0) * d.sAi + (long long)(kbeg + ka[p]) * d.sAk;
        la[p] = ka[p] * LDA + i;
    }
#pragma unroll
    for (int p = 0; p < NLB; ++p) {
        const int e = tid + 256 * p;
        const int j = b_jcont ? (e % BN) : (e / BK);
        kb[p] = b_jcont ? (e / BN) : (e % BK);
        const int gj = n0 + j;
        vb[p] = gj < d.n;
        pb[p] = Bp + (long long)(kbeg + kb[p]) * d.sBk + (long long)(vb[p] ? gj : 0) * d.sBj;
        lb[p] = kb[p] * LDB + j;
    }
    const long long stepA = (long long)BK * d.sAk, stepB = (long long)BK * d.sBk;

    // gload only ISSUES the loads (raw values stay in registers across the MFMA loop); the
    // zero-select / conjugation happen in sstore, after the MFMAs, so that hipcc places the
    // vmcnt wait there and the HBM latency is covered by the matrix pipe.
    bool tail = false;
    int tail_k0 = 0;
    auto gload = [&](int k0) {
        tail = (k0 + BK > kend);
        tail_k0 = k0;
        if (!tail) {                    // full panel (wave-uniform branch)
#pragma unroll
            for (int p = 0; p < NLA; ++p) { ra[p] = *pa[p]; pa[p] += stepA; }
#pragma unroll
            for (int p = 0; p < NLB; ++p) {
                rb[p] = *pb[p];
                if constexpr (EPI == EPI_UPDATE_X) { if (two_b) rb2[p] = pb[p][boff2]; }
                pb[p] += stepB;
            }
        } else {                        // k tail: out-of-range elements read element 0 (zeroed in sstore)
#pragma unroll
            for (int p = 0; p < NLA; ++p) ra[p] = *((va[p] && (k0 + ka[p] < kend)) ? pa[p] : Ap);
#pragma unroll
            for (int p = 0; p < NLB; ++p) {
                const float2 *q = (vb[p] && (k0 + kb[p] < kend)) ? pb[p] : Bp;
                rb[p] = *q;
                if constexpr (EPI == EPI_UPDATE_X) { if (two_b) rb2[p] = q[boff2]; }
            }
        }
    };
    auto sstore = [&](int buf) {
        float2 *a = sA + buf * BK * LDA;
        float2 *bb = sB + buf * BK * LDB;
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            float2 v = ra[p];
            const bool ok = va[p] && (!tail || (tail_k0 + ka[p] < kend));
            if (!ok) v = make_float2(0.f, 0.f);
            v.y *= sgnA;
            a[la[p]] = v;
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            float2 v = rb[p];
            if constexpr (EPI == EPI_UPDATE_X) { if (two_b) { v.x = fmaf(-bir, rb2[p].x, v.x) - birl * rb2[p].x; v.y = fmaf(-bir, rb2[p].y, v.y) - birl * rb2[p].y; } }
            const bool ok = vb[p] && (!tail || (tail_k0 + kb[p] < kend));
            if (!ok) v = make_float2(0.f, 0.f);
            v.y *= sgnB;
            bb[lb[p]] = v;
        }
    };

    constexpr int NACC = M3 ? 3 : 2;       // M3: P1, P2, P3;  else: re, im
    f32x16 acc[NB][NACC];
    double mst[M64 ? NB : 1][2][16];       // fp64 masters always hold (re, im): P1..P3 are combined at flush time
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[nb][c][r] = 0.f;
            if (M64) { mst[nb][0][r] = 0.0; mst[nb][1][r] = 0.0; }
        }

    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    if (nk > 0) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool refresh = (kt + 1 < nk);
        if (refresh) gload(kbeg + (kt + 1) * BK);
        const float2 *a = sA + buf * BK * LDA + wi * 32 + l31;
        const float2 *bb = sB + buf * BK * LDB + wj * (BN / 2) + l31;
        // Fragments of k-pair kp+1 are fetched from LDS before the MFMAs of k-pair kp are issued
        // (explicit register double buffer + scheduling groups), so a wave's MFMA stream does not
        // stall on LDS latency: on a SIMD the oldest wave owns the matrix pipe, and whatever it
        // stalls on is idle pipe time.
        float2 av = a[lhi * LDA], bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = bb[lhi * LDB + nb * 32];
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float2 an = av, bn[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bn[nb] = bv[nb];
            if (kp + 1 < BK / 2) {
                const int kr = 2 * (kp + 1) + lhi;
                an = a[kr * LDA];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bn[nb] = bb[kr * LDB + nb * 32];
            }
            // (b_re + i b_im)(a_re + i a_im): MFMA A-op = b (rows j), B-op = a (cols i)
            if (M3) {
                const float asum = av.x + av.y;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float bsum = bv[nb].x + bv[nb].y;
                    acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[nb].x, av.x, acc[nb][0], 0, 0, 0);
                    acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[nb].y, av.y, acc[nb][1], 0, 0, 0);
                    acc[nb][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(bsum, asum, acc[nb][2], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[nb].x, av.x, acc[nb][0], 0, 0, 0);
                    acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[nb].x, av.y, acc[nb][1], 0, 0, 0);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(-bv[nb].y, av.y, acc[nb][0], 0, 0, 0);
                    acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[nb].y, av.x, acc[nb][1], 0, 0, 0);
                }
            }
            if (kp + 1 < BK / 2) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1 + NB, 0);   // DS reads of the next k-pair first
                __builtin_amdgcn_sched_group_barrier(0x008, (M3 ? 3 : 4) * NB, 0);   // then this k-pair's MFMAs
            }
            av = an;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = bn[nb];
        }
        if (M64 && ((kt % M64_STEPS) == M64_STEPS - 1 || kt + 1 == nk)) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (M3) {       // exact in fp64: re = P1 - P2, im = P3 - P1 - P2
                        const double p1 = (double)acc[nb][0][r], p2 = (double)acc[nb][1][r];
                        mst[nb][0][r] += p1 - p2;
                        mst[nb][1][r] += (double)acc[nb][NACC - 1][r] - p1 - p2;
                    } else {
                        mst[nb][0][r] += (double)acc[nb][0][r];
                        mst[nb][1][r] += (double)acc[nb][1][r];
                    }
#pragma unroll
                    for (int c = 0; c < NACC; ++c) acc[nb][c][r] = 0.f;
                }
        }
        if (refresh) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C = alpha*acc + beta*D ------------------------------------------------
    float2 *Cp = d.C + (long long)t * d.sCt + (long long)split * d.sCsplit;
    const float2 *Dp = (d.D && d.splitk == 1) ? d.D + (long long)t * d.sDt : nullptr;
    const int gi = m0 + wi * 32 + l31;
    float tmax = 0.f;           // max(|re|, |im|) of what this thread produces (d.amax_out)
    float xmax = 0.f, v1max = 0.f, zmax = 0.f;
    // EPI_UPDATE_X with 16-byte accesses: accumulator registers r, r+1 are two adjacent columns of one row; lanes 2q and
    // 2q+1 (adjacent rows) swap one of them (DPP quad_perm [1,0,3,2]), after which the even lane owns rows (i, i+1) of
    // column j_r and the odd lane rows (i-1, i) of column j_r+1 — five float4 loads + one float2 (invD) and four float4
    // stores per register pair instead of twice as many 8-byte ones (the epilogue is memory-instruction-issue bound).
    bool vec4 = false;
    if constexpr (EPI == EPI_UPDATE_X && !M3 && !M64) {
        vec4 = ((d.m & 1) == 0) && ((d.ldc & 1) == 0) && ((d.sCt & 1) == 0) && !Dp &&
               ((((uintptr_t)d.C | (uintptr_t)d.e_rw0 | (uintptr_t)d.e_w1 | (uintptr_t)d.e_w2 | (uintptr_t)d.e_w3 |
                  (uintptr_t)d.e_r0 | (uintptr_t)d.e_r2 | (uintptr_t)d.e_r3) & 15) == 0) && (((uintptr_t)d.e_f0 & 7) == 0);
        if (vec4) {
            auto swap1 = [](float x) {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
            };
            const bool odd = lane & 1;
            const int gi2 = gi & ~1;
            const TrialParams prm = d.prm[t];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    const int r0 = 2 * rp, r1 = r0 + 1;
                    const float2 o0 = make_float2(d.alpha * acc[nb][0][r0], d.alpha * acc[nb][1][r0]);
                    const float2 o1 = make_float2(d.alpha * acc[nb][0][r1], d.alpha * acc[nb][1][r1]);
                    const float2 snd = odd ? o0 : o1;
                    const float2 rcv = make_float2(swap1(snd.x), swap1(snd.y));
                    const float4 y = odd ? make_float4(rcv.x, rcv.y, o1.x, o1.y) : make_float4(o0.x, o0.y, rcv.x, rcv.y);
                    const int gj = n0 + wj * (BN / 2) + nb * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * lhi + (odd ? 1 : 0);
                    if (gi2 >= d.m || gj >= d.n) continue;
                    if (d.epi_store_c == 2) {       // Y only (the fused pass applies the updates): nothing else is touched
                        *reinterpret_cast<float4 *>(Cp + gi2 + (long long)gj * d.ldc) = y;
                        continue;
                    }
                    const long long ix = (long long)t * d.sCt + gi2 + (long long)gj * d.ldc;
                    float4 v1 = *reinterpret_cast<const float4 *>(d.e_rw0 + ix);
                    const float4 v2 = *reinterpret_cast<const float4 *>(d.e_r0 + ix);
                    const float4 xs = *reinterpret_cast<const float4 *>(d.e_r2 + ix);
                    const float4 sy = *reinterpret_cast<const float4 *>(d.e_r3 + ix);
                    const float2 id = *reinterpret_cast<const float2 *>(d.e_f0 + ix);
                    const float4 x = make_float4(admm_x(prm, v1.x, y.x, sy.x, v2.x, xs.x, id.x), admm_x(prm, v1.y, y.y, sy.y, v2.y, xs.y, id.x),
                                                 admm_x(prm, v1.z, y.z, sy.z, v2.z, xs.z, id.y), admm_x(prm, v1.w, y.w, sy.w, v2.w, xs.w, id.y));
                    *reinterpret_cast<float4 *>(d.e_w1 + ix) = x;
                    const float4 kk = make_float4(admm_k(prm, x.x, v2.x), admm_k(prm, x.y, v2.y), admm_k(prm, x.z, v2.z), admm_k(prm, x.w, v2.w));
                    *reinterpret_cast<float4 *>(d.e_w2 + ix) = kk;
                    tmax = fmaxf(fmaxf(tmax, fmaxf(fabsf(kk.x), fabsf(kk.y))), fmaxf(fabsf(kk.z), fabsf(kk.w)));
                    v1 = make_float4(admm_v1(prm, v1.x, y.x, x.x), admm_v1(prm, v1.y, y.y, x.y), admm_v1(prm, v1.z, y.z, x.z),
                                     admm_v1(prm, v1.w, y.w, x.w));
                    *reinterpret_cast<float4 *>(d.e_rw0 + ix) = v1;
                    const float4 zn = make_float4(admm_z(prm, x.x, v1.x), admm_z(prm, x.y, v1.y), admm_z(prm, x.z, v1.z), admm_z(prm, x.w, v1.w));
                    if (d.e_w3) *reinterpret_cast<float4 *>(d.e_w3 + ix) = zn;
                    xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
                    v1max = fmaxf(fmaxf(v1max, fmaxf(fabsf(v1.x), fabsf(v1.y))), fmaxf(fabsf(v1.z), fabsf(v1.w)));
                    zmax = fmaxf(fmaxf(zmax, fmaxf(fabsf(zn.x), fabsf(zn.y))), fmaxf(fabsf(zn.z), fabsf(zn.w)));
                    if (d.epi_store_c) *reinterpret_cast<float4 *>(Cp + gi2 + (long long)gj * d.ldc) = y;
                }
            }
        }
    }
    if (!vec4 && gi < d.m) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gj = n0 + wj * (BN / 2) + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (gj < d.n) {
                    float vr, vi;
                    if (M3) {
                        if (M64) {
                            vr = (float)mst[nb][0][r];
                            vi = (float)mst[nb][1][r];
                        } else {
                            vr = acc[nb][0][r] - acc[nb][1][r];
                            vi = acc[nb][2][r] - acc[nb][0][r] - acc[nb][1][r];
                        }
                    } else {
                        vr = M64 ? (float)mst[nb][0][r] : acc[nb][0][r];
                        vi = M64 ? (float)mst[nb][1][r] : acc[nb][1][r];
                    }
                    float2 o = make_float2(d.alpha * vr, d.alpha * vi);
                    if (Dp) {
                        const float2 dv = Dp[gi + (long long)gj * d.ldd];
                        o.x += d.beta * dv.x;
                        o.y += d.beta * dv.y;
                        if (d.D_lo) {       // (after the leading part: o - D is where the cancellation happens)
                            const float2 dl = d.D_lo[(long long)t * d.sDt + gi + (long long)gj * d.ldd];
                            o.x += d.beta * dl.x;
                            o.y += d.beta * dl.y;
                        }
                    }
                    const long long ix = (long long)t * d.sCt + gi + (long long)gj * d.ldc;
                    if (M64 && EPI == EPI_NONE && d.C_lo)       // what the fp32 result leaves of the float64 sum
                        d.C_lo[(long long)t * d.sCt + gi + (long long)gj * d.ldc] =
                            make_float2((float)(mst[nb][0][r] - (double)vr), (float)(mst[nb][1][r] - (double)vi));
                    if (EPI == EPI_UPDATE_C) {
                        // o = Xs;  V2 <- (1 - cc)(V2 - rho (X - Xs))   (= the reference's :61 + :65, C == -V2)
                        const TrialParams prm = d.prm[t];
                        const float2 x = d.e_r0[ix];
                        float2 v2 = d.e_rw0[ix];
                        v2.x = admm_v2(prm, v2.x, x.x, o.x);
                        v2.y = admm_v2(prm, v2.y, x.y, o.y);
                        d.e_rw0[ix] = v2;
                        Cp[gi + (long long)gj * d.ldc] = o;
                    } else if (EPI == EPI_UPDATE_X) {
                        // o = Y
                        if (d.epi_store_c == 2) { Cp[gi + (long long)gj * d.ldc] = o; continue; }
                        const TrialParams prm = d.prm[t];
                        float2 v1 = d.e_rw0[ix];
                        const float2 v2 = d.e_r0[ix], xs = d.e_r2[ix], sy = d.e_r3[ix];
                        const float id = d.e_f0[ix];
                        const float2 x = make_float2(admm_x(prm, v1.x, o.x, sy.x, v2.x, xs.x, id), admm_x(prm, v1.y, o.y, sy.y, v2.y, xs.y, id));
                        d.e_w1[ix] = x;
                        const float2 kk = make_float2(admm_k(prm, x.x, v2.x), admm_k(prm, x.y, v2.y));
                        d.e_w2[ix] = kk;
                        tmax = fmaxf(tmax, fmaxf(fabsf(kk.x), fabsf(kk.y)));
                        v1 = make_float2(admm_v1(prm, v1.x, o.x, x.x), admm_v1(prm, v1.y, o.y, x.y));
                        d.e_rw0[ix] = v1;
                        const float2 zn = make_float2(admm_z(prm, x.x, v1.x), admm_z(prm, x.y, v1.y));
                        if (d.e_w3) d.e_w3[ix] = zn;
                        xmax = fmaxf(xmax, fmaxf(fabsf(x.x), fabsf(x.y)));
                        v1max = fmaxf(v1max, fmaxf(fabsf(v1.x), fabsf(v1.y)));
                        zmax = fmaxf(zmax, fmaxf(fabsf(zn.x), fabsf(zn.y)));
                        if (d.epi_store_c) Cp[gi + (long long)gj * d.ldc] = o;
                    } else if (EPI == EPI_SADMM) {
                        if (d.sa_mode == 1) {
                            const float den = sadmm_den(d.sa_lr[gi], d.sa_lt[gj], d.sa_rho);
                            Cp[gi + (long long)gj * d.ldc] = make_float2(o.x / den, o.y / den);
                        } else {
                            const float rho = d.sa_rho, ir = 1.f / rho;
                            float2 z = d.e_rw0[ix];
                            const float2 sv = d.e_r0[ix], a = d.e_r2[ix];
                            z = make_float2(sadmm_dual1(z.x, o.x, sv.x, rho), sadmm_dual1(z.y, o.y, sv.y, rho));
                            d.e_rw0[ix] = z;
                            const float2 sn = make_float2(sadmm_soft1(o.x, z.x, ir, d.sa_thr), sadmm_soft1(o.y, z.y, ir, d.sa_thr));
                            d.e_w1[ix] = sn;
                            d.e_w2[ix] = make_float2(sadmm_rhs1(z.x, sn.x, a.x, rho), sadmm_rhs1(z.y, sn.y, a.y, rho));
                        }
                    } else {
                        Cp[gi + (long long)gj * d.ldc] = o;
                        tmax = fmaxf(tmax, fmaxf(fabsf(o.x), fabsf(o.y)));
                    }
                }
            }
        }
    }
    // operand maximum for the split-f16 consumer (hgemm.hip): K after EPI_UPDATE_X, the product otherwise
    if (EPI != EPI_UPDATE_C && d.amax_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
        if (lane == 0) atomicMax(&d.amax_out[t], __float_as_uint(tmax));
    }
    if (EPI == EPI_UPDATE_X && d.amax_x) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            xmax = fmaxf(xmax, __shfl_xor(xmax, o));
            v1max = fmaxf(v1max, __shfl_xor(v1max, o));
            zmax = fmaxf(zmax, __shfl_xor(zmax, o));
        }
        if (lane == 0) {
            atomicMax(&d.amax_x[t], __float_as_uint(xmax));
            atomicMax(&d.amax_v1[t], __float_as_uint(v1max));
            atomicMax(&d.amax_z[t], __float_as_uint(zmax));
        }
    }
}

template <int TAG>
static void launch_tagged(jstsp_ctx *ctx, const GemmDesc &d, int variant, bool m3, long long grid, int tiles_m,
                          int tiles_n)
{
    const dim3 g((unsigned)grid), b(256);
    if (variant == 2) {
        if (m3) hipLaunchKernelGGL((cgemm_kernel<64, TAG, true, true, EPI_NONE>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
        else hipLaunchKernelGGL((cgemm_kernel<64, TAG, true, false, EPI_NONE>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
    } else if (variant == 1) {
        if (m3) hipLaunchKernelGGL((cgemm_kernel<128, TAG, false, true, EPI_NONE>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
        else hipLaunchKernelGGL((cgemm_kernel<128, TAG, false, false, EPI_NONE>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
    } else {
        if (m3) hipLaunchKernelGGL((cgemm_kernel<64, TAG, false, true, EPI_NONE>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
        else hipLaunchKernelGGL((cgemm_kernel<64, TAG, false, false, EPI_NONE>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
    }
}

// Fused-epilogue launches: 128- or 64-wide fp32 tile (never split-K, never M64).
template <int TAG, int EPI>
static void launch_epi(jstsp_ctx *ctx, const GemmDesc &d, bool wide, bool m3, long long grid, int tiles_m, int tiles_n)
{
    const dim3 g((unsigned)grid), b(256);
    if (wide) {
        if (m3) hipLaunchKernelGGL((cgemm_kernel<128, TAG, false, true, EPI>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
        else hipLaunchKernelGGL((cgemm_kernel<128, TAG, false, false, EPI>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
    } else {
        hipLaunchKernelGGL((cgemm_kernel<64, TAG, false, false, EPI>), g, b, 0, ctx->stream, d, tiles_m, tiles_n);
    }
}

static const char *const kTagNames[] = {"misc", "correlate", "synthesize", "gram"};

int launch_cgemm(jstsp_ctx *ctx, const GemmDesc &d, int tag)
{
    if (d.m <= 0 || d.n <= 0 || d.batch <= 0) return 0;
    const int tiles_m = (d.m + BM - 1) / BM;
    // variant 2: 64-wide tile with fp64 master accumulators for long contractions.  Measured at
    // BASELINE configs[1]: only the 4096-term correlation needs it for |dNMSE| <= 1e-6 (1.2e-7 with
    // the threshold at 2048, 1.7e-7 at 256, 4e-6 without); 512/1024-term chains stay on the faster
    // 128-wide fp32 kernel.
    const int long_k = 2048;
    const int kper = (d.k + d.splitk - 1) / d.splitk;
    // Grams only feed the SVT projector / spectral norms (not the gradient): plain fp32 chains are enough
    // Short contractions (the k = 64 products of the gradient step: A^H Tc, G_A Res, A S) take the 64-wide tile whatever n
    // is: 109 instead of 179 registers per lane and 33 instead of 50 KiB of LDS, i.e. 4 instead of 2 workgroups per CU alone and
    // 2 instead of 1 beside a resident eigen-decomposition / Gram workgroup - these launches are all prologue and epilogue,
    // what hides their latency is the number of workgroups in flight (round 3: 340 -> ? us beside the side chains).
    const int bn64_maxk = 64;
    const bool small_k = tag == GEMM_MISC && d.epi == EPI_NONE && kper <= bn64_maxk;
    const int variant = ((kper >= long_k && tag != GEMM_GRAM && d.epi == EPI_NONE) || (d.force_m64 && d.epi == EPI_NONE && d.splitk == 1))
                            ? 2 : ((d.n > 64 && !small_k) ? 1 : 0);
    JSTSP_REQUIRE(!d.C_lo || (variant == 2 && d.alpha == 1.f && !d.D), JSTSP_E_ARG, "cgemm: C_lo needs the fp64-master variant, alpha = 1, no D");
    const int bn = variant == 1 ? 128 : 64;
    // 3M where it pays and was validated: the dominant contractions, the Grams, and other products of 256 terms or more
    static const int m3_mink = [] { const char *e = xp_getenv("JSTSP_M3_MINK"); return e ? atoi(e) : 256; }();
    const bool m3 = tag == GEMM_CORRELATE || tag == GEMM_SYNTH || tag == GEMM_GRAM || (tag == GEMM_MISC && kper >= m3_mink);
    const int tiles_n = (d.n + bn - 1) / bn;
    const long long groups = (d.batch + 7) / 8;
    const long long grid = groups * 8 * tiles_m * tiles_n * d.splitk;
    JSTSP_REQUIRE(grid < (1ll << 31), JSTSP_E_UNSUPPORTED, "cgemm grid too large");
    const char *prof_name = (tag != GEMM_MISC) ? kTagNames[tag] : nullptr;
    if (prof_name) prof_begin(ctx, prof_name);
    if (d.epi == EPI_UPDATE_C) launch_epi<GEMM_SYNTH, EPI_UPDATE_C>(ctx, d, variant == 1, m3, grid, tiles_m, tiles_n);
    else if (d.epi == EPI_UPDATE_X) launch_epi<GEMM_MISC, EPI_UPDATE_X>(ctx, d, variant == 1, false, grid, tiles_m, tiles_n);
    else if (d.epi == EPI_SADMM) launch_epi<GEMM_MISC, EPI_SADMM>(ctx, d, variant == 1, m3, grid, tiles_m, tiles_n);
    else
    switch (tag) {
    case GEMM_CORRELATE: launch_tagged<GEMM_CORRELATE>(ctx, d, variant, m3, grid, tiles_m, tiles_n); break;
    case GEMM_SYNTH: launch_tagged<GEMM_SYNTH>(ctx, d, variant, m3, grid, tiles_m, tiles_n); break;
    case GEMM_GRAM: launch_tagged<GEMM_GRAM>(ctx, d, variant, m3, grid, tiles_m, tiles_n); break;
    default: launch_tagged<GEMM_MISC>(ctx, d, variant, m3, grid, tiles_m, tiles_n); break;
    }
    if (prof_name) prof_end(ctx, prof_name);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
