// hsmall.hip - the 64-term products of the gradient step (proposed_algorithm.m:47-48) on the f16 matrix pipe, ONE kernel for
// the head of the critical chain between two passes:
//     Tc  = sum of the pass's partial sums of K B^H (+ the leading columns of a block-Toeplitz dictionary)
//     Res = A^H Tc - R v                                      `K2'*k - R*v`          (:47)
//     P1  = G_A Res                                           first factor of `R*res` (:48)
// and, for the iterations that recompute R v from v, P1 = (G_A,hi + G_A,lo) V.
//
// Why (round 4, tools/divergence_trace.py + tools/precision_study.py): the iterate v has directions that the gradient step
// does not damp, so an error made in Res - which lives in v-space - is never forgotten; it adds up as a random walk over the
// Imax iterations (relative error of S after k iterations: 1.9e-7 sqrt(k), whatever the big contractions run on).  Of that
// per-iteration error the fp32-MFMA product A^H Tc was the largest part: v_mfma_f32_32x32x2_f32 rounds its accumulator 32
// times over a 64-term sum (1.2e-7 relative).  Here the sum runs on the f16 pipe (v_mfma_f32_32x32x16_f16: exact f16 x f16
// products, one accumulator rounding per 16 terms) with
//   * both operands split THREE ways, x s = h + l + ll: 33 bits, i.e. the fp32 value exactly.  (The first version used the
//     two-way split of hgemm.hip: 22 bits of A - a CONSTANT operand, so a constant perturbation of K2' against the exact
//     G_A, G_B of `R*v` - made the result worse than the fp32 chain, rms |dNMSE| 3.7e-7 against 1.7e-7: quantising an
//     operator is a bias, not noise);
//   * the leading stream h h in its own accumulator per 32 terms (two roundings each), the five lower-order streams
//     (h l, l h, h ll, l l, ll h: 2^-11 and below) together in another, the three summed in float64: about three roundings
//     at full scale instead of 32.
// It is also less matrix-pipe time than the fp32 form (24 f16 MFMAs per 32 x 32 x 16 block against 64 + of 4x the issue
// time), and three launches of the chain become one.
//
// Shapes: N = Gr = 64 (every shape the fused pass takes), G2 a multiple of 64.  One 256-thread workgroup per (trial, 64
// columns of G2): both operands of a product are split on the fly into MFMA fragment order in LDS (the layout of hgemm.hip:
// [row half][k-step][plane][lane][8 halves], here six planes re_h re_l re_ll im_h im_l im_ll and one 32-term stage at a time),
// scales are exact powers of two from block maxima.
#include "solver_common.h"

namespace jstsp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __host__ inline int scale_exp_s(float amax)
{
    const uint32_t bits = __builtin_bit_cast(uint32_t, amax);
    const int be = (int)((bits >> 23) & 0xff);
    if (be == 0 || be == 255) return 0;
    return 13 - (be - 127);
}
__device__ __forceinline__ half8 as_h8(uint4 u) { return *reinterpret_cast<half8 *>(&u); }
__device__ __forceinline__ half8 neg_h8(uint4 u)
{
    u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
    return *reinterpret_cast<half8 *>(&u);
}

// One operand of a 64 x 64 x 64 product held by the workgroup as 16 values per thread: row = tid & 63, the 8 consecutive
// k = 32 s + 8 (tid >> 6) + v of stage s = 0, 1.
struct Rows { float2 v[2][8]; };

__device__ __forceinline__ float rows_max(const Rows &r)
{
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int v = 0; v < 8; ++v) m = fmaxf(m, fmaxf(fabsf(r.v[s][v].x), fabsf(r.v[s][v].y)));
    return m;
}

// rows of a column-major matrix whose column `row` holds the k index contiguously: element (row, k) = P[k + 64 row]
__device__ __forceinline__ void load_rows(Rows &r, const float2 *P, int tid)
{
    const float4 *p = reinterpret_cast<const float4 *>(P + 64 * (tid & 63) + 8 * (tid >> 6));
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = p[16 * s + q];
            r.v[s][2 * q] = make_float2(x.x, x.y);
            r.v[s][2 * q + 1] = make_float2(x.z, x.w);
        }
}

// fragments of stage st (32 terms) into buf (24 x 64 uint4: [row half 2][k-step 2][plane 6][lane]); the imaginary parts
// negated if cj (the operand is the conjugate)
__device__ __forceinline__ void split3(float x, _Float16 &h, _Float16 &l, _Float16 &ll)
{
    h = (_Float16)x;
    const float r1 = x - (float)h;          // exact
    l = (_Float16)r1;
    ll = (_Float16)(r1 - (float)l);         // exact difference, then the last 11 bits
}
__device__ __forceinline__ void store_frags(const Rows &r, int st, uint4 *buf, int tid, float s, bool cj)
{
    const int row = tid & 63, akg = tid >> 6;
    const int slot = ((((row >> 5) * 2 + (akg >> 1)) * 6) * 64) + (akg & 1) * 32 + (row & 31);
    const float si = cj ? -s : s;
    half8 rh, rl, rll, ih, il, ill;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        _Float16 h, l, ll;
        split3(r.v[st][v].x * s, h, l, ll); rh[v] = h; rl[v] = l; rll[v] = ll;
        split3(r.v[st][v].y * si, h, l, ll); ih[v] = h; il[v] = l; ill[v] = ll;
    }
    uint4 *q = buf + slot;
    q[0] = *reinterpret_cast<uint4 *>(&rh);
    q[64] = *reinterpret_cast<uint4 *>(&rl);
    q[128] = *reinterpret_cast<uint4 *>(&rll);
    q[192] = *reinterpret_cast<uint4 *>(&ih);
    q[256] = *reinterpret_cast<uint4 *>(&il);
    q[320] = *reinterpret_cast<uint4 *>(&ill);
}

// one 32-term stage:  hh(lane, r) += sum_k xh(g, k) ch(a, k) (the leading stream), lo += the five lower-order streams;
// a = 32 wi + (lane & 31) [fragments bufC],  g = 32 wj + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) [fragments bufX]; complex:
// re = xr cr - xi ci,  im = xr ci + xi cr
#define JSTSP_CMAC(ACCR, ACCI, XR, NXI, XI, CR, CI)                                      \
    ACCR = __builtin_amdgcn_mfma_f32_32x32x16_f16(XR, CR, ACCR, 0, 0, 0);                \
    ACCI = __builtin_amdgcn_mfma_f32_32x32x16_f16(XR, CI, ACCI, 0, 0, 0);                \
    ACCR = __builtin_amdgcn_mfma_f32_32x32x16_f16(NXI, CI, ACCR, 0, 0, 0);               \
    ACCI = __builtin_amdgcn_mfma_f32_32x32x16_f16(XI, CR, ACCI, 0, 0, 0)
__device__ __forceinline__ void product(const uint4 *bufX, const uint4 *bufC, int wi, int wj, int lane, f32x16 &hr, f32x16 &hi,
                                        f32x16 &lr, f32x16 &li)
{
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint4 *fc = bufC + ((wi * 2 + ks) * 6) * 64 + lane;
        const uint4 *fx = bufX + ((wj * 2 + ks) * 6) * 64 + lane;
        const half8 cr_h = as_h8(fc[0]), cr_l = as_h8(fc[64]), cr_m = as_h8(fc[128]);
        const half8 ci_h = as_h8(fc[192]), ci_l = as_h8(fc[256]), ci_m = as_h8(fc[320]);
        const uint4 uxi_h = fx[192], uxi_l = fx[256], uxi_m = fx[320];
        const half8 xr_h = as_h8(fx[0]), xr_l = as_h8(fx[64]), xr_m = as_h8(fx[128]);
        const half8 xi_h = as_h8(uxi_h), xi_l = as_h8(uxi_l), xi_m = as_h8(uxi_m);
        const half8 nxi_h = neg_h8(uxi_h), nxi_l = neg_h8(uxi_l), nxi_m = neg_h8(uxi_m);
        JSTSP_CMAC(lr, li, xr_m, nxi_m, xi_m, cr_h, ci_h);      // ll h
        JSTSP_CMAC(lr, li, xr_h, nxi_h, xi_h, cr_m, ci_m);      // h ll
        JSTSP_CMAC(lr, li, xr_l, nxi_l, xi_l, cr_l, ci_l);      // l l
        JSTSP_CMAC(lr, li, xr_l, nxi_l, xi_l, cr_h, ci_h);      // l h
        JSTSP_CMAC(lr, li, xr_h, nxi_h, xi_h, cr_l, ci_l);      // h l
        JSTSP_CMAC(hr, hi, xr_h, nxi_h, xi_h, cr_h, ci_h);      // h h
    }
}
#undef JSTSP_CMAC

// the whole 64-term product of two operands held as Rows: both stages through the two stage buffers
struct Acc { f32x16 h0r, h0i, h1r, h1i, lr, li; };
__device__ __forceinline__ void product64(const Rows &rx, float sx, const Rows &rc, float sc, uint4 *bufX, uint4 *bufC, int tid,
                                          int wi, int wj, Acc &a)
{
    const int lane = tid & 63;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a.h0r[r] = 0.f; a.h0i[r] = 0.f; a.h1r[r] = 0.f; a.h1i[r] = 0.f; a.lr[r] = 0.f; a.li[r] = 0.f; }
    store_frags(rx, 0, bufX, tid, sx, false);
    store_frags(rc, 0, bufC, tid, sc, true);
    __syncthreads();
    product(bufX, bufC, wi, wj, lane, a.h0r, a.h0i, a.lr, a.li);
    __syncthreads();
    store_frags(rx, 1, bufX, tid, sx, false);
    store_frags(rc, 1, bufC, tid, sc, true);
    __syncthreads();
    product(bufX, bufC, wi, wj, lane, a.h1r, a.h1i, a.lr, a.li);
}
__device__ __forceinline__ float2 acc_value(const Acc &a, int r, float alpha)
{
    const double vr = (double)a.h0r[r] + (double)a.h1r[r] + (double)a.lr[r], vi = (double)a.h0i[r] + (double)a.h1i[r] + (double)a.li[r];
    return make_float2((float)vr * alpha, (float)vi * alpha);
}

// maxima of two per-thread values over the workgroup (one barrier pair)
__device__ __forceinline__ void block_max2(float &a, float &b, float *sh, int tid)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
    __syncthreads();
    if ((tid & 63) == 0) { sh[tid >> 6] = a; sh[4 + (tid >> 6)] = b; }
    __syncthreads();
    a = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    b = fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]));
}

struct HeadDesc {
    const float2 *P; long long sPt, sPp; int parts;       // Tc[t] = sum_p P[t sPt + p sPp + n + 64 g]
    const float2 *Kf, *Bdl; long long sBdl;               // leading columns of a block-Toeplitz dictionary (fused_pass64), or NULL
    const float2 *A; long long sA;                        // N x Gr = 64 x 64 (sA = 0: shared)
    const float2 *GA; long long sGA;                      // Gr x Gr Hermitian
    const float2 *RV;                                     // Gr x G2 per trial, or NULL (R v = 0)
    const float2 *RVlo;                                   // its low-order part (R v carried as two floats), or NULL
    float2 *Tc;                                           // optional output: the summed Tc (N x G2), NULL = not stored
    float2 *Res, *P1;                                     // Gr x G2 per trial
    uint32_t *pmax;                                       // [batch] atomicMax of max(|re|, |im|) of P1 (float bits)
    int G2;
};

__global__ __launch_bounds__(256, 2) void grad_head_kernel(HeadDesc d)
{
    __shared__ uint4 lds[2 * 24 * 64];      // one 32-term stage of the right (Tc, then Res) and of the left (conj(A) columns, then
    __shared__ float shm[8];                //  G_A) operand's fragments, 24 KiB each; in between: Res as float2 [g][a] (32 KiB)
    uint4 *bufX = lds, *bufC = lds + 24 * 64;
    const int nb = d.G2 >> 6;
    const int t = blockIdx.x / nb, g0 = (blockIdx.x % nb) << 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave & 1, wj = wave >> 1;
    const long long sg = 64ll * d.G2;
    // ---- operands of the first product: rows a of A^H = conj of column a of A;  rows g of Tc^T = column g of Tc
    Rows ra, rt;
    load_rows(ra, d.A + (long long)t * d.sA, tid);
    {
        const float2 *p = d.P + (long long)t * d.sPt + 64ll * g0;
        load_rows(rt, p, tid);
        for (int s = 1; s < d.parts; ++s) {
            Rows rp;
            load_rows(rp, p + (long long)s * d.sPp, tid);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int v = 0; v < 8; ++v) { rt.v[st][v].x += rp.v[st][v].x; rt.v[st][v].y += rp.v[st][v].y; }
        }
        if (d.Kf) {         // Tc[n, g] += sum over m < ld(g) of k[n, m] conj(B[g, m])   (fused.hip: reduce_parts_delta_kernel)
            const int g = g0 + (tid & 63), ld = g >> 6;
            const float2 *kf = d.Kf + (long long)t * 512, *b = d.Bdl + (long long)t * d.sBdl + g * 8;
            for (int m = 0; m < ld; ++m) {
                const float2 c = b[m];
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const float2 k0 = kf[32 * st + 8 * (tid >> 6) + v + 64 * m];
                        rt.v[st][v].x = fmaf(k0.x, c.x, fmaf(k0.y, c.y, rt.v[st][v].x));
                        rt.v[st][v].y = fmaf(k0.y, c.x, fmaf(-k0.x, c.y, rt.v[st][v].y));
                    }
            }
        }
        if (d.Tc) {
            float4 *o = reinterpret_cast<float4 *>(d.Tc + (long long)t * sg + 64ll * g0 + 64 * (tid & 63) + 8 * (tid >> 6));
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[16 * st + q] = make_float4(rt.v[st][2 * q].x, rt.v[st][2 * q].y, rt.v[st][2 * q + 1].x, rt.v[st][2 * q + 1].y);
        }
    }
    float ma = rows_max(ra), mt = rows_max(rt);
    block_max2(ma, mt, shm, tid);
    const int ea = scale_exp_s(ma), et = scale_exp_s(mt);
    Acc acc;
    product64(rt, ldexpf(1.f, et), ra, ldexpf(1.f, ea), bufX, bufC, tid, wi, wj, acc);
    // ---- Res = A^H Tc - R v
    const float al1 = ldexpf(1.f, -(ea + et));
    const int a = wi * 32 + (lane & 31);
    const long long obase = (long long)t * sg + a + 64ll * g0;
    float2 res[16];
    float mr = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float2 v = acc_value(acc, r, al1);
        if (d.RV) { const float2 rv = d.RV[obase + 64ll * g]; v.x -= rv.x; v.y -= rv.y; }
        if (d.RVlo) { const float2 rl = d.RVlo[obase + 64ll * g]; v.x -= rl.x; v.y -= rl.y; }
        res[r] = v;
        d.Res[obase + 64ll * g] = v;
        mr = fmaxf(mr, fmaxf(fabsf(v.x), fabsf(v.y)));
    }
    // ---- second product: rows a' of G_A (= conj of column a': Hermitian), rows g of Res^T through LDS
    Rows rg;
    load_rows(rg, d.GA + (long long)t * d.sGA, tid);
    float mg = rows_max(rg);
    block_max2(mr, mg, shm, tid);                          // (its barriers also end every wave's reads of the stage buffers)
    const int er = scale_exp_s(mr), eg = scale_exp_s(mg);
    float2 *scr = reinterpret_cast<float2 *>(lds);         // Res as [g][a], 64 x 64 float2 = 32 of the 48 KiB
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scr[64 * g + a] = res[r];
    }
    __syncthreads();
    Rows rr;
    load_rows(rr, scr, tid);
    __syncthreads();
    product64(rr, ldexpf(1.f, er), rg, ldexpf(1.f, eg), bufX, bufC, tid, wi, wj, acc);
    const float al2 = ldexpf(1.f, -(er + eg));
    float mp = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float2 v = acc_value(acc, r, al2);
        d.P1[obase + 64ll * g] = v;
        mp = fmaxf(mp, fmaxf(fabsf(v.x), fabsf(v.y)));
    }
    if (d.pmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mp = fmaxf(mp, __shfl_xor(mp, o));
        if (lane == 0) atomicMax(&d.pmax[t], __float_as_uint(mp));
    }
}

// P1 = (G_hi + G_lo) X   (G Hermitian 64 x 64 in two floats; X 64 x G2): the first factor of R v when it is recomputed from v
__global__ __launch_bounds__(256, 2) void left2_kernel(const float2 *Ghi, const float2 *Glo, long long sG, const float2 *X, float2 *P1,
                                                       uint32_t *pmax, int G2)
{
    __shared__ uint4 lds[2 * 24 * 64];
    __shared__ float shm[8];
    uint4 *bufX = lds, *bufC = lds + 24 * 64;
    const int nb = G2 >> 6;
    const int t = blockIdx.x / nb, g0 = (blockIdx.x % nb) << 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave & 1, wj = wave >> 1;
    const long long sg = 64ll * G2;
    Rows rx, rh, rl;
    load_rows(rx, X + (long long)t * sg + 64ll * g0, tid);
    load_rows(rh, Ghi + (long long)t * sG, tid);
    load_rows(rl, Glo + (long long)t * sG, tid);
    float mx = rows_max(rx), mh = rows_max(rh), ml = rows_max(rl), dummy = 0.f;
    block_max2(mx, mh, shm, tid);
    block_max2(ml, dummy, shm, tid);
    const int ex = scale_exp_s(mx), eh = scale_exp_s(mh), el = scale_exp_s(ml);
    Acc acc;
    product64(rx, ldexpf(1.f, ex), rh, ldexpf(1.f, eh), bufX, bufC, tid, wi, wj, acc);
    const float a1 = ldexpf(1.f, -(ex + eh)), a2 = ldexpf(1.f, -(ex + el));
    float2 out[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = acc_value(acc, r, a1);
    if (ml > 0.f) {                                        // + G_lo X (uniform over the workgroup)
        __syncthreads();
        product64(rx, ldexpf(1.f, ex), rl, ldexpf(1.f, el), bufX, bufC, tid, wi, wj, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float2 v = acc_value(acc, r, a2); out[r].x += v.x; out[r].y += v.y; }
    }
    const int a = wi * 32 + (lane & 31);
    const long long obase = (long long)t * sg + a + 64ll * g0;
    float mp = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        P1[obase + 64ll * g] = out[r];
        mp = fmaxf(mp, fmaxf(fabsf(out[r].x), fabsf(out[r].y)));
    }
    if (pmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mp = fmaxf(mp, __shfl_xor(mp, o));
        if (lane == 0) atomicMax(&pmax[t], __float_as_uint(mp));
    }
}

}  // namespace

bool grad_head_shape_ok(int N, int Gr, int G2) { return N == 64 && Gr == 64 && G2 >= 64 && (G2 & 63) == 0; }

int launch_grad_head(jstsp_ctx *ctx, int G2, int batch, const float2 *P, long long sPt, long long sPp, int parts, const float2 *Kf,
                     const float2 *Bdl, long long sBdl, const float2 *A, long long sA, const float2 *GA, long long sGA,
                     const float2 *RV, float2 *Tc, float2 *Res, float2 *P1, uint32_t *pmax, const float2 *RVlo)
{
    JSTSP_REQUIRE((G2 & 63) == 0 && parts >= 1, JSTSP_E_SHAPE, "grad_head: G2 = %d, parts = %d", G2, parts);
    HeadDesc d{P, sPt, sPp, parts, Kf, Bdl, sBdl, A, sA, GA, sGA, RV, RVlo, Tc, Res, P1, pmax, G2};
    hipLaunchKernelGGL(grad_head_kernel, dim3((unsigned)(batch * (G2 >> 6))), dim3(256), 0, ctx->stream, d);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

int launch_left2(jstsp_ctx *ctx, int G2, int batch, const float2 *Ghi, const float2 *Glo, long long sG, const float2 *X, float2 *P1,
                 uint32_t *pmax)
{
    JSTSP_REQUIRE((G2 & 63) == 0, JSTSP_E_SHAPE, "left2: G2 = %d", G2);
    hipLaunchKernelGGL(left2_kernel, dim3((unsigned)(batch * (G2 >> 6))), dim3(256), 0, ctx->stream, Ghi, Glo, sG, X, P1, pmax, G2);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
