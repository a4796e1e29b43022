// One pass over the dictionary per ADMM iteration (proposed_algorithm.m:38-65, 'approximate').
//
// An iteration of the reference touches B twice: Xs = A S B (:58) and, one iteration later, K2'*k = A^H K B^H (:47),
// with the element-wise updates of X, V1, V2, C and k (:38-43,:61-65) in between.  Both products stream the whole
// dictionary from HBM (the dominant traffic of the solver).  Here one kernel walks over B ONCE per iteration: for a
// tile of 32 columns m of B (all G2 rows, 128 KiB in split-f16 form) resident in LDS it computes
//   phase A   Xs(:, tile)  = (A S) B(:, tile)                            contraction over g   (:58)
//   update    V2, X, V1, k of the NEXT iteration on that tile            element-wise        (:61-65, :38-43)
//   phase B   P += k(:, tile) B(:, tile)^H                               contraction over m   (:47, first factor)
// and keeps the N x G2 sums P of its column range in registers; the per-range partials are summed afterwards.
//
// The same B tile serves as an MFMA operand with g as contraction index (phase A) and with m as contraction index
// (phase B).  Its LDS image is [plane][m/4][g/8][4 m][8 g] halves: a row of a micro-block (16 bytes: 8 consecutive g of one m)
// is a phase-A fragment (ds_read_b128); a column of four halves (4 consecutive m of one g), delivered by the transposing LDS
// read ds_read_b64_tr_b16, is half a phase-B fragment.  Everything is computed transposed (Xs^T = B^T (A S)^T,
// P^T = conj(B) K^T) so that the accumulator layout of phase A (lane = n, registers = 4 consecutive m) IS the B-operand
// layout of phase B's 16x16x32 MFMA: k goes from the element-wise update to the second product through a 24-KiB LDS
// exchange (6 planes: k_re, k_im split in two halves each, and -k_re) and never touches HBM.  Y = (I - Q) Z of the next
// iteration is formed in the kernel as well (YIN): Z of the next tile is requested during phase B and multiplied by the
// fragments of I - Q after it.
//
// The kernel lives at the register limit (256 VGPRs with the 64 x 512 complex sums of a column range in 128 of them):
// see DESIGN.md section 5 for what made it fit without spills, and for the one ordering rule its use under three
// streams needs (section 5, 'Reproducibility').
//
// Shapes: N = 64, G2 = 128, 256, 384 or 512 (GB = G2 / 128 blocks of 16 rows g per wave), M a multiple of 32 * parts.
// Everything else keeps the three-kernel path.
#include "solver_common.h"
#include <cstdlib>

namespace jstsp {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FPAD = 128;         // bytes of padding per LDS row: row stride = 2 (mod 4) 64-byte slots, for which the four
                                  // 16-lane groups of the phase-A ds_read_b128 each cover all 64 banks once
// (FusedDesc::kback, 4 by default: the k scale is taken 2^4 below the one its previous maximum would give, see fused_pass_kernel)

// e such that amax * 2^e lies in [2^13, 2^14)   (as hgemm.hip)
__device__ __host__ inline int fscale_exp(uint32_t amax_bits)
{
    const int be = (int)((amax_bits >> 23) & 0xff);
    if (be == 0 || be == 255) return 0;
    return 13 - (be - 127);
}
__device__ __forceinline__ void fsplit(float x, _Float16 &h, _Float16 &l)
{
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // native vectors: arrays of them stay in registers
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));      // (arrays of HIP's uint4 struct were kept in scratch)
__device__ __forceinline__ u32x4 negu(u32x4 u) { return u ^ 0x80008000u; }
// uniform base pointer + 32-bit BYTE offset per lane: the global_load / store with an SGPR base and one VGPR of offset
// (an element index makes hipcc build, hoist and spill a 64-bit address per access)
template <class T> __device__ __forceinline__ T ldg(const void *base, uint32_t boff)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + boff);
}
// the same, non-temporal: lines that one CU reads once (dictionary tiles, state arrays) must not push the (A S) fragments,
// which every tile re-reads, out of L2
template <class T> __device__ __forceinline__ T ldg_nt(const void *base, uint32_t boff)
{
    return __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + boff));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ldg_nt2(const void *base, uint32_t boff)      // (the builtin wants a native vector type)
{
    const f32x2 v = ldg_nt<f32x2>(base, boff);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void stg_nt2(void *base, uint32_t boff, float2 v)
{
    __builtin_nontemporal_store(f32x2{v.x, v.y}, reinterpret_cast<f32x2 *>(reinterpret_cast<char *>(base) + boff));
}
template <class T> __device__ __forceinline__ void stg(void *base, uint32_t boff, T v)
{
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + boff) = v;
}
__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// ---- B -> tile images.  A tile is 32 columns m of B, all G2 rows, as halves in micro-blocks [4 m][8 g] (64 bytes): one
//      row (16 bytes: 8 consecutive g of one m) is a phase-A fragment, a column of four halves a phase-B fragment piece.
//      16-byte chunk (p, mq, g8, r) = row r of micro-block (m quad mq, g octet g8) of plane p.  In HBM the chunks are ordered
//      [wave w][block gb][p][mq][g8 & 1][r] with g8 = 2 (GB w + gb) + (g8 & 1), GB = G2 / 128: the 4 KiB a wave fetches per
//      refill step (its own 16 rows g of phase B) are contiguous.
__global__ __launch_bounds__(256) void pack_bf_kernel(const float2 *B, long long sBt, int G2, int M, const uint32_t *bmax,
                                                      int sbmax, uint4 *out, long long sOut)
{
    const int t = blockIdx.y;
    const int G8 = G2 >> 3;
    const long long per_tile = 32ll * G8;                   // 8 mq * 4 r * G8
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)(M / 32) * per_tile) return;
    const int tile = (int)(idx / per_tile);
    int rem = (int)(idx % per_tile);
    const int g8 = rem % G8; rem /= G8;                     // g fastest: coalesced reads of B
    const int r = rem & 3, mq = rem >> 2;
    const int m = tile * 32 + 4 * mq + r;
    const float s = ldexpf(1.f, fscale_exp(bmax[(long long)t * sbmax]));
    half8 pl[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float2 x = B[(long long)t * sBt + (8 * g8 + c) + (long long)G2 * m];
        _Float16 h, l;
        fsplit(x.x * s, h, l); pl[0][c] = h; pl[1][c] = l;
        fsplit(x.y * s, h, l); pl[2][c] = h; pl[3][c] = l;
    }
    uint4 *o = out + (long long)t * sOut + (long long)tile * (16ll * G2);
    const int blk = g8 >> 1, g8i = g8 & 1;                  // blk = GB w + gb
#pragma unroll
    for (int p = 0; p < 4; ++p) o[(((long long)(blk * 4 + p) * 8 + mq) * 2 + g8i) * 4 + r] = *reinterpret_cast<uint4 *>(&pl[p]);
}

// ---- W = A S (N x G2, column-major) -> B-operand fragments of (A S)^T: out[t][ks G2/32][nb 4][plane 4][lane 64],
//      lane l: n = 16 nb + (l & 15), g = 32 ks + 8 (l >> 4) + 0..7
__global__ __launch_bounds__(256) void pack_as_kernel(const float2 *W, long long sWt, int G2, const uint32_t *wmax, uint4 *out,
                                                      long long sOut)
{
    const int t = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (G2 / 32) * 256) return;
    const int lane = idx & 63, nb = (idx >> 6) & 3, ks = idx >> 8;
    const int n = 16 * nb + (lane & 15), g0 = 32 * ks + 8 * (lane >> 4);
    const float s = ldexpf(1.f, fscale_exp(wmax[t]));
    half8 pl[4];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const float2 x = W[(long long)t * sWt + n + 64ll * (g0 + v)];
        _Float16 h, l;
        fsplit(x.x * s, h, l); pl[0][v] = h; pl[1][v] = l;
        fsplit(x.y * s, h, l); pl[2][v] = h; pl[3][v] = l;
    }
    uint4 *o = out + (long long)t * sOut + (long long)((ks * 4 + nb) * 4) * 64 + lane;
#pragma unroll
    for (int p = 0; p < 4; ++p) o[p * 64] = *reinterpret_cast<uint4 *>(&pl[p]);
}

// ---- Wq = I - Q (N x N, the SVT re-projection: Y = Wq Z) -> B-operand fragments of Wq^T: out[t][ks 2][nb 4][plane 4][lane 64],
//      lane l: n = 16 nb + (l & 15), n' = 32 ks + 8 (l >> 4) + 0..7, value Wq[n, n'] scaled by 2^13 (|entries| <= 1)
__global__ __launch_bounds__(256) void pack_wq_kernel(const float2 *Q, uint4 *out)
{
    const int t = blockIdx.x;
    const float2 *q = Q + (long long)t * 4096;
    for (int idx = threadIdx.x; idx < 512; idx += 256) {
        const int lane = idx & 63, nb = (idx >> 6) & 3, ks = idx >> 8;
        const int n = 16 * nb + (lane & 15), c0 = 32 * ks + 8 * (lane >> 4);
        half8 pl[4];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float2 x = q[n + 64 * (c0 + v)];
            _Float16 h, l;
            fsplit(((n == c0 + v) ? 1.f : 0.f) * 8192.f - x.x * 8192.f, h, l); pl[0][v] = h; pl[1][v] = l;
            fsplit(-x.y * 8192.f, h, l); pl[2][v] = h; pl[3][v] = l;
        }
        uint4 *o = out + (long long)t * 2048 + (long long)((ks * 4 + nb) * 4) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p * 64] = *reinterpret_cast<uint4 *>(&pl[p]);
    }
}

// ---- Tc[t] = sum over the column ranges of the partial sums
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float4 *P, int parts, long long n4, float4 *out)
{
    const int t = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 *p = P + (long long)t * parts * n4 + i;
    float4 a = p[0];
    for (int s = 1; s < parts; ++s) {
        const float4 b = p[(long long)s * n4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    out[(long long)t * n4 + i] = a;
}

// DBG != 0 (timing experiments only, results are wrong; not instantiated by default): 1 skips the phase-A products,
// 2 the element-wise loads / stores, 4 the phase-B products, 8 the tile refill
// YIN: Y = (I - Q) Z of the next iteration is formed here (Z from d.Zin, fragments of I - Q from d.Wqp) instead of read
template <int GB, int DBG, bool YIN>
__global__ __launch_bounds__(512, 1) void fused_pass_kernel(FusedDesc d)
{
    constexpr int G2 = 128 * GB;
    constexpr int ROWB = G2 * 8 + FPAD;        // bytes of one LDS row: (G2/8) micro-blocks of 64 B
    constexpr int TILEB = 32 * ROWB;           // 4 planes x 8 m-quads
    constexpr int KSH = G2 / 64;               // 32-wide k-steps per g-half
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char *tile = lds;
    unsigned char *xch = lds + TILEB;          // 24 KiB: phase-A partial sums (16), then the k fragments (6 planes)

    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int t = (slot / d.parts) * 8 + xcd;  // the column ranges of one problem run on ONE XCD: (A S) stays in its L2
    if (t >= d.batch) return;
    const int part = slot % d.parts;
    const int tpw = (d.M / 32) / d.parts;
    const int tile0 = part * tpw;

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, q = l >> 4, c16 = l & 15;
    const int nb = w & 3, kh = w >> 2;

    const TrialParams prm = d.prm[t];
    const int eb = fscale_exp(d.bmax[(long long)t * d.sbmax]), ew = fscale_exp(d.wmax[t]);
    // k of this pass is formed and consumed here, its maximum is only known afterwards: the scale comes from the
    // previous iteration's max|k|, backed off by 2^kback (room for a 2^(2+kback)-fold growth before f16 overflows; entries
    // keep 22 bits down to 2^-13 of the maximum and 2^-25 absolute of the scaled range below - far under the fp32 noise
    // of the sums).  An overflow raises d.ovf.
    const int ek = fscale_exp(d.kmax_prev[t]) - d.kback;
    const float sxs = ldexpf(1.f, -(eb + ew)), sk = ldexpf(1.f, ek), sp = ldexpf(1.f, -(eb + ek));
    const float rho = prm.rho, ir = prm.irho, omc = 1.f - prm.c_coef, omr = 1.f - rho, omir = 1.f - ir;

    // refill of the tile: wave w owns the rows g of its phase-B range, block gb of them = 256 chunks, 4 per lane (plane p = c)
    const u32x4 *const bt = reinterpret_cast<const u32x4 *>(d.Bf) + (long long)t * d.sBf + (long long)tile0 * (16ll * G2);
    const uint32_t boff = 16u * ((uint32_t)(w * GB) * 256u + l);          // bytes
    constexpr uint32_t tile_b = 256u * G2;                                // bytes per tile
    unsigned char *rdst = tile + ((l >> 3) & 7) * ROWB + (2 * GB * w + ((l >> 2) & 1)) * 64 + (l & 3) * 16;   // + p 8 ROWB + gb 128

    f32x4 pr[GB][4], pi[GB][4];
#pragma unroll
    for (int gb = 0; gb < GB; ++gb)
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) { pr[gb][n2] = f32x4{0.f, 0.f, 0.f, 0.f}; pi[gb][n2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float kmx = 0.f, xmx = 0.f, v1mx = 0.f, zmx = 0.f, v2mx = 0.f;

    // uniform per-problem pointers (SGPR pairs) + 32-bit lane offsets: 64-bit lane addresses would not fit next to P
    float2 *const Xt = d.X + (long long)t * d.snm, *const V1t = d.V1 + (long long)t * d.snm, *const V2t = d.V2 + (long long)t * d.snm;
    const float2 *const sYt = d.subY + (long long)t * d.snm, *const Yt = d.Y + (long long)t * d.snm;
    const float *const iDt = d.invD + (long long)t * d.snm;
    const float4 *const Zit4 = YIN ? reinterpret_cast<const float4 *>(d.Zin + (long long)t * d.snm) : nullptr;
    float2 *const Zot = YIN ? d.Zout + (long long)t * d.snm : nullptr;
    float2 *const Yot = d.Yout ? d.Yout + (long long)t * d.snm : nullptr;
    const u32x4 *const wqt = YIN ? reinterpret_cast<const u32x4 *>(d.Wqp) + (long long)t * 2048 : nullptr;
    const uint32_t wqoff = 16u * (nb * 256 + l), zoff = 16u * (32u * (uint32_t)(16 * kh + c16) + 4 * q);      // bytes
    const float sy = YIN ? ldexpf(1.f, -(fscale_exp(d.zmax_in[t]) + 13)) : 0.f, sz = YIN ? ldexpf(1.f, fscale_exp(d.zmax_in[t])) : 0.f;
    const uint32_t ebase = 8u * (16 * nb + c16 + 64 * (16 * kh + 4 * q));     // bytes; + 512 (m0 + s)
    const u32x4 *const ast = reinterpret_cast<const u32x4 *>(d.ASp) + (long long)t * d.sAS;
    const uint32_t aoff = 16u * ((uint32_t)(kh * KSH) * 1024u + nb * 256 + l);      // bytes

    {
#pragma unroll
        for (int gb = 0; gb < GB; ++gb)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<u32x4 *>(rdst + c * 8 * ROWB + gb * 128) = ldg_nt<u32x4>(bt, boff + gb * 4096 + c * 1024);
    }
    __syncthreads();

    // Y^T(block) = Z^T Wq^T for this wave's element-wise block of a tile: A operand = Z^T (lane = column m of Z, 8 consecutive
    // rows n' per k-step: 64 contiguous bytes), B operand = the fragments of Wq^T; two k-steps of 32 rows.  The Z loads of the
    // NEXT tile are requested in the middle of phase B and multiplied after it, so that Y costs no exposed latency.
    float4 zz[2][4];
    float2 ey[4];
#define FUSED_ZLOAD(m0_, ks_)                                                                                                \
    {                                                                                                                        \
        const uint32_t zo_ = zoff + 512u * (uint32_t)(m0_) + 256 * (ks_);                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) zz[ks_][j] = ldg<float4>(Zit4, zo_ + 16 * j);                          \
    }
#define FUSED_YCOMP()                                                                                                        \
    {                                                                                                                        \
        f32x4 yr = f32x4{0.f, 0.f, 0.f, 0.f}, yi = yr;                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
        {                                                                                                                    \
            u32x4 wq[4];                                                                                                     \
            uint32_t wo_ = wqoff + ks * 16384;          /* (opaque: else a 64-bit address per plane is built and spilled) */   \
            asm volatile("" : "+v"(wo_));                                                                                    \
            _Pragma("unroll") for (int p = 0; p < 4; ++p) wq[p] = ldg<u32x4>(wqt, wo_ + p * 1024);                                   \
            half8 zp[4];                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
            {                                                                                                                \
                _Float16 h, lo;                                                                                              \
                fsplit(zz[ks][j].x * sz, h, lo); zp[0][2 * j] = h; zp[1][2 * j] = lo;                                        \
                fsplit(zz[ks][j].y * sz, h, lo); zp[2][2 * j] = h; zp[3][2 * j] = lo;                                        \
                fsplit(zz[ks][j].z * sz, h, lo); zp[0][2 * j + 1] = h; zp[1][2 * j + 1] = lo;                                \
                fsplit(zz[ks][j].w * sz, h, lo); zp[2][2 * j + 1] = h; zp[3][2 * j + 1] = lo;                                \
            }                                                                                                                \
            u32x4 zf[4];                                                                                                     \
            _Pragma("unroll") for (int p = 0; p < 4; ++p) zf[p] = __builtin_bit_cast(u32x4, zp[p]);                          \
            const u32x4 nwi_h = negu(wq[2]), nwi_l = negu(wq[3]);                                                            \
            yr = mma(zf[0], wq[0], yr); yi = mma(zf[0], wq[2], yi);                                                          \
            yr = mma(zf[0], wq[1], yr); yi = mma(zf[0], wq[3], yi);                                                          \
            yr = mma(zf[1], wq[0], yr); yi = mma(zf[1], wq[2], yi);                                                          \
            yr = mma(zf[2], nwi_h, yr); yi = mma(zf[2], wq[0], yi);                                                          \
            yr = mma(zf[2], nwi_l, yr); yi = mma(zf[2], wq[1], yi);                                                          \
            yr = mma(zf[3], nwi_h, yr); yi = mma(zf[3], wq[0], yi);                                                          \
        }                                                                                                                    \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) ey[s] = make_float2(yr[s] * sy, yi[s] * sy);                           \
    }
    if (YIN) {
        FUSED_ZLOAD(tile0 * 32, 0)
        FUSED_ZLOAD(tile0 * 32, 1)
        FUSED_YCOMP()
    }

    for (int i = 0; i < tpw; ++i) {
        const int m0 = (tile0 + i) * 32;
        // ================= phase A: Xs^T(tile) = B^T (A S)^T, this wave: n-block nb, g-half kh, both m-blocks
        // (A S) fragments: requested one k-step (24 products) ahead; B^T fragments of the next
        // (k-step, m-block) are read from LDS before the products of the current one are issued
        f32x4 ar[2], ai[2];
        ar[0] = ar[1] = ai[0] = ai[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(DBG & 1)) {
            u32x4 wr[2][4], bfb[2][4];
            // ONE running offset, advanced opaquely per k-step: with constant offsets the compiler materialises (and spills) an
            // address per (k-step, plane) outside the tile loop
            uint32_t ao = aoff;
            asm volatile("" : "+v"(ao));
#pragma unroll
            for (int p = 0; p < 4; ++p) wr[0][p] = ldg<u32x4>(ast, ao + p * 1024);
            const int goff0 = (kh * (G2 / 2) + 8 * q) * 8 + (c16 & 3) * 16;      // micro-block (m quad, g octet), row m & 3
            const unsigned char *arow = tile + (c16 >> 2) * ROWB + goff0;         // + p 8 ROWB + mb 4 ROWB + ks 256
#pragma unroll
            for (int p = 0; p < 4; ++p) bfb[0][p] = *reinterpret_cast<const u32x4 *>(arow + p * 8 * ROWB);
#pragma unroll
            for (int st = 0; st < 2 * KSH; ++st) {
                const int ks = st >> 1, mb = st & 1;
                if (mb == 0 && ks + 1 < KSH) {
                    ao += 16384;
                    asm volatile("" : "+v"(ao));
#pragma unroll
                    for (int p = 0; p < 4; ++p) wr[(ks + 1) & 1][p] = ldg<u32x4>(ast, ao + p * 1024);
                }
                if (!YIN && st + 1 < 2 * KSH) {
                    const int ks1 = (st + 1) >> 1, mb1 = (st + 1) & 1;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        bfb[(st + 1) & 1][p] = *reinterpret_cast<const u32x4 *>(arow + (p * 8 + 4 * mb1) * ROWB + ks1 * 256);
                }
                if (YIN && st > 0) {        // (no register room for the second fragment set next to the carried Y)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        bfb[0][p] = *reinterpret_cast<const u32x4 *>(arow + (p * 8 + 4 * mb) * ROWB + ks * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 *wf = wr[ks & 1], *bf = bfb[YIN ? 0 : (st & 1)];
                const u32x4 nwi_h = negu(wf[2]), nwi_l = negu(wf[3]);
                // re += Br Wr - Bi Wi ; im += Br Wi + Bi Wr   (h h + h l + l h each)
                ar[mb] = mma(bf[0], wf[0], ar[mb]); ai[mb] = mma(bf[0], wf[2], ai[mb]);
                ar[mb] = mma(bf[0], wf[1], ar[mb]); ai[mb] = mma(bf[0], wf[3], ai[mb]);
                ar[mb] = mma(bf[1], wf[0], ar[mb]); ai[mb] = mma(bf[1], wf[2], ai[mb]);
                ar[mb] = mma(bf[2], nwi_h, ar[mb]); ai[mb] = mma(bf[2], wf[0], ai[mb]);
                ar[mb] = mma(bf[2], nwi_l, ar[mb]); ai[mb] = mma(bf[2], wf[1], ai[mb]);
                ar[mb] = mma(bf[3], nwi_h, ar[mb]); ai[mb] = mma(bf[3], wf[0], ai[mb]);
            }
        }
        // the two g-halves meet: wave (nb, kh) keeps m-block kh and hands m-block 1 - kh to wave (nb, 1 - kh)
        {
            const f32x4 sr = kh ? ar[0] : ar[1], si = kh ? ai[0] : ai[1];
            f32x4 *x4 = reinterpret_cast<f32x4 *>(xch);
            const int dw = nb + 4 * (1 - kh);
            x4[(dw * 2 + 0) * 64 + l] = sr;
            x4[(dw * 2 + 1) * 64 + l] = si;
        }
        // the element-wise operands of this wave's block: n = 16 nb + c16, m = m0 + 16 kh + 4 q + s
        float2 ex[4], ev1[4], ev2[4], esy[4];
        float eid[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t ix = ebase + 512u * (uint32_t)(m0 + s);
            if (DBG & 2) { ex[s] = ev1[s] = ev2[s] = esy[s] = ey[s] = make_float2(1.f, 1.f); eid[s] = 1.f; continue; }
            ex[s] = ldg_nt2(Xt, ix); ev1[s] = ldg_nt2(V1t, ix); ev2[s] = ldg_nt2(V2t, ix);
            esy[s] = ldg_nt2(sYt, ix); eid[s] = ldg_nt<float>(iDt, ix >> 1);
            if (!YIN) ey[s] = ldg_nt2(Yt, ix);
        }
        __syncthreads();
        f32x4 xr = kh ? ar[1] : ar[0], xi = kh ? ai[1] : ai[0];
        {
            const f32x4 *x4 = reinterpret_cast<const f32x4 *>(xch);
            const f32x4 orr = x4[(w * 2 + 0) * 64 + l], oi = x4[(w * 2 + 1) * 64 + l];
            xr += orr; xi += oi;
        }
        half4 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t ix = ebase + 512u * (uint32_t)(m0 + s);
            const float2 xs = make_float2(xr[s] * sxs, xi[s] * sxs);
            // V2 <- (1 - cc)(V2 - rho (X - Xs))                       (:61 + :65, C == -V2)
            const float2 v2 = make_float2(omc * (ev2[s].x - rho * (ex[s].x - xs.x)), omc * (ev2[s].y - rho * (ex[s].y - xs.y)));
            // X <- iK1 (V1 + rho Y + subY + V2 + rho C + rho Xs)      (:38-40)
            const float2 x = make_float2((ev1[s].x + rho * ey[s].x + esy[s].x + omr * v2.x + rho * xs.x) * eid[s],
                                         (ev1[s].y + rho * ey[s].y + esy[s].y + omr * v2.y + rho * xs.y) * eid[s]);
            const float2 kk = make_float2(x.x + omir * v2.x, x.y + omir * v2.y);                  // (:43)
            const float2 v1 = make_float2(ev1[s].x + rho * (ey[s].x - x.x), ev1[s].y + rho * (ey[s].y - x.y));   // (:64)
            if (!(DBG & 2)) { stg_nt2(V2t, ix, v2); stg_nt2(Xt, ix, x); stg_nt2(V1t, ix, v1); }
            const float2 zn = make_float2(x.x - ir * v1.x, x.y - ir * v1.y);
            if (YIN) stg_nt2(Zot, ix, zn);
            if (Yot) stg(Yot, ix, ey[s]);
            v2mx = fmaxf(v2mx, fmaxf(fabsf(v2.x), fabsf(v2.y)));
            xmx = fmaxf(xmx, fmaxf(fabsf(x.x), fabsf(x.y)));
            v1mx = fmaxf(v1mx, fmaxf(fabsf(v1.x), fabsf(v1.y)));
            zmx = fmaxf(zmx, fmaxf(fabsf(zn.x), fabsf(zn.y)));
            kmx = fmaxf(kmx, fmaxf(fabsf(kk.x), fabsf(kk.y)));
            _Float16 h, lo;
            fsplit(kk.x * sk, h, lo); kf[0][s] = h; kf[1][s] = lo;
            fsplit(kk.y * sk, h, lo); kf[2][s] = h; kf[3][s] = lo;
        }
        __syncthreads();                        // every wave has read its partial sums: the exchange area is free
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<half4 *>(xch + ((nb * 6 + p) * 64 + l) * 16 + kh * 8) = kf[p];
        // planes 4, 5: -k_re (the imaginary part of conj(B) k needs it; negating per product costs registers)
        *reinterpret_cast<half4 *>(xch + ((nb * 6 + 4) * 64 + l) * 16 + kh * 8) = -kf[0];
        *reinterpret_cast<half4 *>(xch + ((nb * 6 + 5) * 64 + l) * 16 + kh * 8) = -kf[1];
        __syncthreads();
        // the next tile (the last one is fetched again: unconditional loads) replaces this one block by block: only this
        // wave reads its rows g in phase B, so block gb is overwritten as soon as its products are issued
        const uint32_t noff = boff + (uint32_t)min(i + 1, tpw - 1) * tile_b;         // (< 4 GiB of tiles per problem and range)
        u32x4 rf[2][4];
        if (!(DBG & 8)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { rf[0][c] = ldg_nt<u32x4>(bt, noff + c * 1024); if (GB > 1) rf[1][c] = ldg_nt<u32x4>(bt, noff + 4096 + c * 1024); }
        }
        // ================= phase B: P^T += conj(B)(g, tile) k^T(tile, :), this wave: g in [16 GB w, 16 GB (w + 1))
        // A operand: lane = g, registers = 8 of the 32 columns m - two transposing reads (ds_read_b64_tr_b16) of the
        // micro-block image: lane i' of a 16-lane group points at the four halves g = g0 + 4 (i' & 3) .. + 3 of row i' >> 2 and
        // lane i receives column g0 + i of the four rows.  
        if (!(DBG & 4)) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const uint32_t tr0 = (uint32_t)(uintptr_t)(tile - lds) + q * ROWB + (2 * GB * w + ((c16 >> 1) & 1)) * 64 + (c16 >> 2) * 16 +
                                 (c16 & 1) * 8;
            auto *lbase = (__attribute__((address_space(3))) unsigned char *)lds;
#define FUSED_BFRAG(dst, gb_)                                                                                              \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                                          \
    {                                                                                                                      \
        const uint32_t o_ = tr0 + p * 8 * ROWB + (gb_) * 128;                                                             \
        const u32x2 lo_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lbase + o_)));   \
        const u32x2 hi_ = __builtin_bit_cast(                                                                              \
            u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lbase + o_ + 4 * ROWB)));                        \
        dst[p] = u32x4{lo_.x, lo_.y, hi_.x, hi_.y};                                                                       \
    }
#pragma unroll
            for (int gb = 0; gb < GB; ++gb) {
                // (before the products of the last block: by then one refill register set is free)
                if (YIN && gb == GB - 1) {
                    FUSED_ZLOAD((tile0 + min(i + 1, tpw - 1)) * 32, 0)
                    FUSED_ZLOAD((tile0 + min(i + 1, tpw - 1)) * 32, 1)
                    __builtin_amdgcn_sched_barrier(0);
                }
                u32x4 bf[4];
                FUSED_BFRAG(bf, gb)
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) {
                    const unsigned char *kp = xch + (n2 * 6 * 64 + l) * 16;
                    u32x4 k0 = *reinterpret_cast<const u32x4 *>(kp), k1 = *reinterpret_cast<const u32x4 *>(kp + 1024);
                    const u32x4 k2 = *reinterpret_cast<const u32x4 *>(kp + 2048), k3 = *reinterpret_cast<const u32x4 *>(kp + 3072);
                    // re += Br kr + Bi ki ; im += Br ki - Bi kr
                    pr[gb][n2] = mma(bf[0], k0, pr[gb][n2]); pi[gb][n2] = mma(bf[0], k2, pi[gb][n2]);
                    pr[gb][n2] = mma(bf[0], k1, pr[gb][n2]); pi[gb][n2] = mma(bf[0], k3, pi[gb][n2]);
                    pr[gb][n2] = mma(bf[1], k0, pr[gb][n2]); pi[gb][n2] = mma(bf[1], k2, pi[gb][n2]);
                    k0 = *reinterpret_cast<const u32x4 *>(kp + 4096); k1 = *reinterpret_cast<const u32x4 *>(kp + 5120);   // -kr
                    pr[gb][n2] = mma(bf[2], k2, pr[gb][n2]); pi[gb][n2] = mma(bf[2], k0, pi[gb][n2]);
                    pr[gb][n2] = mma(bf[2], k3, pr[gb][n2]); pi[gb][n2] = mma(bf[2], k1, pi[gb][n2]);
                    pr[gb][n2] = mma(bf[3], k2, pr[gb][n2]); pi[gb][n2] = mma(bf[3], k0, pi[gb][n2]);
                    __builtin_amdgcn_sched_barrier(0);      // (else the fragment reads of all four n-blocks are hoisted: spills)
                }
                if (!(DBG & 8)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        *reinterpret_cast<u32x4 *>(rdst + c * 8 * ROWB + gb * 128) = rf[gb & 1][c];
                        if (gb + 2 < GB) rf[gb & 1][c] = ldg_nt<u32x4>(bt, noff + (gb + 2) * 4096 + c * 1024);
                    }
                }
            }
#undef FUSED_BFRAG
        }
        if (YIN) FUSED_YCOMP()
        __syncthreads();                        // next tile in place, k fragments dead
    }

#undef FUSED_ZLOAD
#undef FUSED_YCOMP
    // ---- partial sums of this column range: Ppart[t][part][n + 64 g]
    float2 *po = d.Ppart + ((long long)t * d.parts + part) * (64ll * G2);
#pragma unroll
    for (int gb = 0; gb < GB; ++gb)
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int g = 16 * GB * w + 16 * gb + 4 * q + s;
                po[64ll * g + 16 * n2 + c16] = make_float2(pr[gb][n2][s] * sp, pi[gb][n2][s] * sp);
            }
    // ---- operand maxima of the next consumers, overflow flag of the k scale
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmx = fmaxf(kmx, __shfl_xor(kmx, o)); xmx = fmaxf(xmx, __shfl_xor(xmx, o)); v1mx = fmaxf(v1mx, __shfl_xor(v1mx, o));
        zmx = fmaxf(zmx, __shfl_xor(zmx, o)); v2mx = fmaxf(v2mx, __shfl_xor(v2mx, o));
    }
    if (l == 0) {
        if (d.kmax_out) atomicMax(&d.kmax_out[t], __float_as_uint(kmx));
        if (d.xmax) atomicMax(&d.xmax[t], __float_as_uint(xmx));
        if (d.v1max) atomicMax(&d.v1max[t], __float_as_uint(v1mx));
        if (d.zmax) atomicMax(&d.zmax[t], __float_as_uint(zmx));
        if (d.v2max) atomicMax(&d.v2max[t], __float_as_uint(v2mx));
        if (d.ovf && !(kmx * sk < 60000.f)) atomicOr(&d.ovf[t], 1u);      // per trial: the caller re-solves exactly those
    }
}

}  // namespace

bool fused_shape_ok(int N, int M, int G2, int parts)
{
    return N == 64 && G2 >= 128 && G2 <= 512 && G2 % 128 == 0 && parts > 0 && M % (32 * parts) == 0;
}

size_t fused_bytes(int M, int G2, int nB, int batch, int parts)
{
    return rnd256((size_t)nB * (M / 32) * 16 * G2 * sizeof(uint4)) + rnd256((size_t)batch * (G2 / 32) * 1024 * sizeof(uint4)) +
           rnd256((size_t)batch * parts * 64 * G2 * sizeof(float2)) + rnd256((size_t)batch * sizeof(uint32_t)) + rnd256((size_t)batch * 2048 * sizeof(uint4));
}

int fused_alloc(Arena &ar, FusedWS &f, int M, int G2, int nB, int batch, int parts)
{
    f.parts = parts;
    f.sBf = (long long)(M / 32) * 16 * G2;
    f.sAS = (long long)(G2 / 32) * 1024;
    f.Bf = ar.get<uint4>((size_t)nB * f.sBf);
    f.ASp = ar.get<uint4>((size_t)batch * f.sAS);
    f.Ppart = ar.get<float2>((size_t)batch * parts * 64 * G2);
    f.ovf = ar.get<uint32_t>(batch);
    f.Wqp = ar.get<uint4>((size_t)batch * 2048);
    JSTSP_REQUIRE(f.Bf && f.ASp && f.Ppart && f.ovf && f.Wqp, JSTSP_E_NOMEM, "fused pass: workspace exhausted");
    return 0;
}

int fused_pack_b(jstsp_ctx *ctx, const FusedWS &f, const float2 *B, long long sBt, int G2, int M, int nB, const uint32_t *bmax)
{
    const long long n = (long long)(M / 32) * 16 * (G2 / 4);
    hipLaunchKernelGGL(pack_bf_kernel, dim3((unsigned)((n + 255) / 256), nB), dim3(256), 0, ctx->stream, B, sBt, G2, M, bmax, 1,
                       f.Bf, f.sBf);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

int fused_pack_as(jstsp_ctx *ctx, const FusedWS &f, const float2 *W, long long sWt, int G2, int batch, const uint32_t *wmax)
{
    hipLaunchKernelGGL(pack_as_kernel, dim3((G2 / 32), batch), dim3(256), 0, ctx->stream, W, sWt, G2, wmax, f.ASp, f.sAS);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

template <int GB, bool YIN> static int launch_fused_gb(jstsp_ctx *ctx, const FusedDesc &d)
{
    const size_t sh = (size_t)32 * (128 * GB * 8 + FPAD) + 24576;
    const int grid = ((d.batch + 7) / 8) * 8 * d.parts;
    JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass_kernel<GB, 0, YIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL((fused_pass_kernel<GB, 0, YIN>), dim3(grid), dim3(512), sh, ctx->stream, d);
    return 0;
}

int launch_fused_pass(jstsp_ctx *ctx, const FusedDesc &d)
{
    JSTSP_REQUIRE(fused_shape_ok(64, d.M, d.G2, d.parts), JSTSP_E_UNSUPPORTED, "fused pass: shape");
    prof_begin(ctx, "fused_pass");
    int rc = 0;
    if (d.Wqp) {
        switch (d.G2 / 128) {
        case 1: rc = launch_fused_gb<1, true>(ctx, d); break;
        case 2: rc = launch_fused_gb<2, true>(ctx, d); break;
        case 3: rc = launch_fused_gb<3, true>(ctx, d); break;
        default: rc = launch_fused_gb<4, true>(ctx, d); break;
        }
    } else {        // Y read from memory (JSTSP_FUSED_Y=0)
        switch (d.G2 / 128) {
        case 1: rc = launch_fused_gb<1, false>(ctx, d); break;
        case 2: rc = launch_fused_gb<2, false>(ctx, d); break;
        case 3: rc = launch_fused_gb<3, false>(ctx, d); break;
        default: rc = launch_fused_gb<4, false>(ctx, d); break;
        }
    }
    prof_end(ctx, "fused_pass");
    JSTSP_TRY(rc);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

int fused_pack_wq(jstsp_ctx *ctx, const FusedWS &f, const float2 *Q, int batch)
{
    hipLaunchKernelGGL(pack_wq_kernel, dim3(batch), dim3(256), 0, ctx->stream, Q, f.Wqp);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

int fused_reduce(jstsp_ctx *ctx, const FusedWS &f, int G2, int batch, float2 *Tc)
{
    const long long n4 = 64ll * G2 / 2;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((n4 + 255) / 256), batch), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float4 *>(f.Ppart), f.parts, n4, reinterpret_cast<float4 *>(Tc));
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
