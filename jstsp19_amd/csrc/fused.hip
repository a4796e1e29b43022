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
//
// Round 3: when the dictionary is block-Toeplitz in the delay index - B(ld Gt + g, m) == B(g, m - ld), what the reference's
// drivers build; probed exactly, never assumed - the same pass runs on much less data: fused_pass_kernel<.., TOEP> refills
// its tile from a compact image of the first block, and for block height 64 fused_pass64_kernel keeps only the 64 x 39
// window of that block in LDS and uses the room for a prefetch of the next tile's element-wise operands (second half of
// this file).
#include "solver_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

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

// ---- Block-Toeplitz dictionaries.  The dictionaries the reference's drivers build (B = the pilot frame delayed by ld samples
//      under every transmit steering vector: system_model.m / errorVSsnr.m:36-47) satisfy, bit for bit,
//          B(ld Gt + g, m) == B(g, m - ld)     for m >= ld,   ld = 0 .. L - 1,  G2 = L Gt:
//      block ld is block 0 shifted right by ld columns; only the ld leading columns of block ld differ.  The property is
//      a fact about the DATA, so it is probed (exact comparison of every entry, all candidate block heights in one pass)
//      and never assumed.  When it holds, the tile image shrinks L-fold: one 16-byte chunk (8 rows g of one column m, one
//      plane) of block ld is the chunk of block 0 at column m - ld, so the pass's refill reads the COMPACT image
//          E[p 4][g8' Gt/8][halo | columns 0 .. M-1]
//      at shifted columns; the L re-reads of a chunk by the L waves of a workgroup hit in L2 / L1 instead of HBM, and the
//      LDS tile - hence every product and every result bit - is the same as with the full image.  The columns m < ld of
//      block ld live in the halo: chunk (ld, c = m - ld < 0) at index c - ld (ld - 1) / 2 relative to column 0.
__global__ __launch_bounds__(256) void toeplitz_probe_kernel(const float2 *B, long long sBt, int G2, int M, uint32_t cand, uint32_t *mism)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;   // (G2 M < 2^31)
    uint32_t bad = 0;
    if (idx < (uint32_t)G2 * (uint32_t)M) {
        const int m = (int)(idx / (uint32_t)G2), g = (int)(idx - (uint32_t)m * (uint32_t)G2);
        const uint2 *b = reinterpret_cast<const uint2 *>(B) + (long long)blockIdx.y * sBt;
        const uint2 x = b[idx];
#pragma unroll
        for (int c = 0; c < 5; ++c) {                       // candidate block heights 16 .. 256
            const int gt = 16 << c, ld = g >> (4 + c);
            if (!((cand >> c) & 1u) || ld == 0 || m < ld) continue;
            const uint2 y = b[idx - (uint32_t)gt - (uint32_t)G2];
            if (x.x != y.x || x.y != y.y) bad |= 1u << c;
        }
    }
    // (the wrong candidates fail almost everywhere: one atomic per wave, and only for bits that are not raised yet)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && (bad & ~__atomic_load_n(mism, __ATOMIC_RELAXED))) atomicOr(mism, bad);
}

// the same for ONE block height over all dictionaries (the confirming pass): four entries per thread, 16-byte loads
__global__ __launch_bounds__(256) void toeplitz_verify_kernel(const float2 *B, long long sBt, int G2, int M, int c, uint32_t *mism)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;     // quad of rows g = 4 (q mod G2/4) .. + 3 of column q / (G2/4)
    const uint32_t qpc = (uint32_t)G2 >> 2;
    uint32_t bad = 0;
    if (q < qpc * (uint32_t)M) {
        const int m = (int)(q / qpc), g = 4 * (int)(q - (uint32_t)m * qpc);
        const int gt = 16 << c, ld = g >> (4 + c);
        if (ld >= 1 && m >= ld) {
            const uint4 *b = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint2 *>(B) + (long long)blockIdx.y * sBt);
            const uint32_t e = (uint32_t)g + (uint32_t)G2 * (uint32_t)m;     // (even: 16-byte aligned pairs)
            const uint4 x0 = b[e >> 1], x1 = b[(e >> 1) + 1];
            const uint32_t f = e - (uint32_t)gt - (uint32_t)G2;
            const uint4 y0 = b[f >> 1], y1 = b[(f >> 1) + 1];
            if (x0.x != y0.x || x0.y != y0.y || x0.z != y0.z || x0.w != y0.w || x1.x != y1.x || x1.y != y1.y || x1.z != y1.z ||
                x1.w != y1.w)
                bad = 1u << c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && (bad & ~__atomic_load_n(mism, __ATOMIC_RELAXED))) atomicOr(mism, bad);
}

__global__ __launch_bounds__(256) void pack_e_kernel(const float2 *B, long long sBt, int G2, int M, int gt, int ecols, int ehalo,
                                                     const uint32_t *bmax, int sbmax, uint4 *out, long long sOut)
{
    const int t = blockIdx.y;
    const int G8t = gt >> 3, L = G2 / gt;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)G8t * ecols) return;
    const int g8 = (int)(idx % G8t), j = (int)(idx / G8t);  // g fastest: coalesced reads of B
    int row0 = 8 * g8, m = j - ehalo;
    bool live = true;
    if (m < 0) {                                            // halo: chunk (ld, c) at -(ld (ld - 1) / 2) + c, c in [-ld, -1]
        const int jh = -m;
        int ld = 1;
        while (ld * (ld + 1) / 2 < jh) ++ld;
        live = ld < L;
        const int c = -(jh - ld * (ld - 1) / 2);
        m = c + ld; row0 += ld * gt;
    }
    const float s = ldexpf(1.f, fscale_exp(bmax[(long long)t * sbmax]));
    half8 pl[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float2 x = live ? B[(long long)t * sBt + (row0 + c) + (long long)G2 * m] : make_float2(0.f, 0.f);
        _Float16 h, l;
        fsplit(x.x * s, h, l); pl[0][c] = h; pl[1][c] = l;
        fsplit(x.y * s, h, l); pl[2][c] = h; pl[3][c] = l;
    }
    uint4 *o = out + (long long)t * sOut;
#pragma unroll
    for (int p = 0; p < 4; ++p) o[(long long)(p * G8t + g8) * ecols + j] = *reinterpret_cast<uint4 *>(&pl[p]);
}

// ---- v2 (block height 64, fused_pass64_kernel): the image IS the LDS layout of the window.  Plane p is an array of columns
//      cc = m + 7 (7 zero columns in front, one behind: the window of tile T is columns 32 T .. 32 T + 39, one contiguous
//      5-KiB piece per plane), a column is 8 octets of 16 bytes (8 rows g each), octet o stored at o ^ swz(cc):
//      swz(x) = 2 ((x >> 1) & 3) | ((x >> 3) & 1) makes both the phase-A reads (16 consecutive columns, one octet) and the
//      phase-B transposing reads (8 consecutive columns, an octet pair) conflict-free at ANY column offset - the shift by
//      the delay ld moves the window of each wave.  The columns m < ld of block ld are NOT in this image (zeros instead): they
//      are applied in fp32 outside the products (xs_delta below, reduce_parts_delta_kernel).
__host__ __device__ inline int eswz(int x) { return (((x >> 1) & 3) << 1) | ((x >> 3) & 1); }

__global__ __launch_bounds__(256) void pack_e2_kernel(const float2 *B, long long sBt, int G2, int M, const uint32_t *bmax, int sbmax,
                                                      uint4 *out, long long sOut, float2 *Bdl)
{
    const int t = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < G2 * 8)                                       // the leading columns, row-major: Bdl[t][g][m < 8]
        Bdl[(long long)t * G2 * 8 + idx] = B[(long long)t * sBt + (idx >> 3) + (long long)G2 * (idx & 7)];
    if (idx >= (M + 8) * 8) return;
    const int o = idx & 7, cc = idx >> 3, m = cc - 7;
    const bool live = m >= 0 && m < M;
    const float s = ldexpf(1.f, fscale_exp(bmax[(long long)t * sbmax]));
    half8 pl[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float2 x = live ? B[(long long)t * sBt + (8 * o + c) + (long long)G2 * m] : make_float2(0.f, 0.f);
        _Float16 h, l;
        fsplit(x.x * s, h, l); pl[0][c] = h; pl[1][c] = l;
        fsplit(x.y * s, h, l); pl[2][c] = h; pl[3][c] = l;
    }
    uint4 *dst = out + (long long)t * sOut;
#pragma unroll
    for (int p = 0; p < 4; ++p) dst[((long long)p * (M + 8) + cc) * 8 + (o ^ eswz(cc))] = *reinterpret_cast<uint4 *>(&pl[p]);
}

// XsD[t][r][n + 64 m], summed over r = 0..3: sum over ld > m, g < 64 of W[n, 64 ld + g] B[64 ld + g, m]   (m < L - 1: what the
// leading columns of the delayed blocks add to Xs = (A S) B; fp32).  Four blocks of 256 threads per (problem, column): lane = n,
// the sixteen waves split the rows g (the pass adds the four partial sums); Bdl = the leading columns row-major.
__device__ __forceinline__ void xs_delta(const float2 *W, const float2 *Bdl, int G2, int m, int r, float2 *out)
{
    __shared__ float2 part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g0 = 64 * (m + 1), cnt = G2 - g0;             // rows of the blocks ld > m; this block: quarter r of them
    float2 a = make_float2(0.f, 0.f), a2 = a;
    const int per = (cnt + 15) / 16, lo = min(g0 + (4 * r + wv) * per, G2), hi = min(lo + per, G2);
    // (eight rows per trip: the loads of a trip are independent and in flight together - with two, the loop is a chain of
    //  memory latencies, 75 us for 0.2 MFLOP per problem on the critical path of the iteration)
    int g = lo;
    for (; g + 8 <= hi; g += 8) {
        float2 w[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { w[u] = W[lane + 64ll * (g + u)]; b[u] = Bdl[(g + u) * 8 + m]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            a.x = fmaf(w[u].x, b[u].x, fmaf(-w[u].y, b[u].y, a.x)); a.y = fmaf(w[u].x, b[u].y, fmaf(w[u].y, b[u].x, a.y));
            a2.x = fmaf(w[u + 1].x, b[u + 1].x, fmaf(-w[u + 1].y, b[u + 1].y, a2.x));
            a2.y = fmaf(w[u + 1].x, b[u + 1].y, fmaf(w[u + 1].y, b[u + 1].x, a2.y));
        }
    }
    for (; g < hi; ++g) {
        const float2 w0 = W[lane + 64ll * g], b0 = Bdl[g * 8 + m];
        a.x = fmaf(w0.x, b0.x, fmaf(-w0.y, b0.y, a.x)); a.y = fmaf(w0.x, b0.y, fmaf(w0.y, b0.x, a.y));
    }
    part[wv][lane] = make_float2(a.x + a2.x, a.y + a2.y);
    __syncthreads();
    if (wv == 0) {
        const float2 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
        out[lane + 64 * m] = make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
    }
}

// ---- W = A S (N x G2, column-major) -> B-operand fragments of (A S)^T: out[t][ks G2/32][nb 4][plane 4][lane 64],
//      lane l: n = 16 nb + (l & 15), g = 32 ks + 8 (l >> 4) + 0..7
__global__ __launch_bounds__(256) void pack_as_kernel(const float2 *W, long long sWt, int G2, const uint32_t *wmax, uint4 *out,
                                                      long long sOut, const float2 *Bd, long long sBd, float2 *XsD)
{
    const int t = blockIdx.y;
    if ((int)blockIdx.x >= G2 / 32) {            // (v2 only: four more blocks per leading column of the problem)
        const int e = (int)blockIdx.x - G2 / 32;
        xs_delta(W + (long long)t * sWt, Bd + (long long)t * sBd, G2, e >> 2, e & 3, XsD + (long long)t * 2048 + (e & 3) * 512);
        return;
    }
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
    // (float64: the sum of the column ranges is part of the 4096-term sum K B^H, whose error goes straight into v-space -
    //  hsmall.hip's note; a memory-bound kernel adds in double for free)
    const float4 a0 = p[0];
    double ax = a0.x, ay = a0.y, az = a0.z, aw = a0.w;
    for (int s = 1; s < parts; ++s) {
        const float4 b = p[(long long)s * n4];
        ax += b.x; ay += b.y; az += b.z; aw += b.w;
    }
    out[(long long)t * n4 + i] = make_float4((float)ax, (float)ay, (float)az, (float)aw);
}

// the same with the leading columns of a block-Toeplitz dictionary (v2): Tc[n, g] += sum over m < ld(g) of k[n, m] conj(B[g, m])
__global__ __launch_bounds__(256) void reduce_parts_delta_kernel(const float4 *P, int parts, long long n4, float4 *out, const float2 *Kf,
                                                                 const float2 *B, long long sBt, int G2)
{
    const int t = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 *p = P + (long long)t * parts * n4 + i;
    const float4 a0 = p[0];
    double ax = a0.x, ay = a0.y, az = a0.z, aw = a0.w;      // (float64: see reduce_parts_kernel)
    for (int s = 1; s < parts; ++s) {
        const float4 b = p[(long long)s * n4];
        ax += b.x; ay += b.y; az += b.z; aw += b.w;
    }
    const int n = (int)((2 * i) & 63), g = (int)((2 * i) >> 6), ld = g >> 6;
    const float2 *kf = Kf + (long long)t * 512, *b = B + (long long)t * sBt + g * 8;     // (the leading columns row-major)
    for (int m = 0; m < ld; ++m) {
        const float2 k0 = kf[n + 64 * m], k1 = kf[n + 1 + 64 * m], c = b[m];
        ax += (double)k0.x * c.x + (double)k0.y * c.y; ay += (double)k0.y * c.x - (double)k0.x * c.y;
        az += (double)k1.x * c.x + (double)k1.y * c.y; aw += (double)k1.y * c.x - (double)k1.x * c.y;
    }
    out[(long long)t * n4 + i] = make_float4((float)ax, (float)ay, (float)az, (float)aw);
}

// DBG != 0 (timing experiments only, results are wrong; not instantiated by default): 1 skips the phase-A products,
// 2 the element-wise loads / stores, 4 the phase-B products, 8 the tile refill
// YIN: Y = (I - Q) Z of the next iteration is formed here (Z from d.Zin, fragments of I - Q from d.Wqp) instead of read
// TOEP: the tile comes from the compact image of a block-Toeplitz dictionary (above) instead of the full tile image
template <int GB, int DBG, bool YIN, bool TOEP>
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
    // (the coefficients of the element-wise updates: admm_* of common.h, two floats each)

    // refill of the tile: wave w owns the rows g of its phase-B range, block gb of them = 256 chunks, 4 per lane (plane p = c)
    // compact image: chunk (block gb, plane c) of lane l = E[(c G8t + g8'(gb) + g8i) ecols + ehalo + m - ld(gb)], m = 4 mq + r:
    // a lane part (g8i, mq, r), a part that is uniform in the wave (ug[gb] + c ps) and 512 bytes per tile
    const u32x4 *const bt = TOEP ? reinterpret_cast<const u32x4 *>(d.Ec) + (long long)t * d.sEc + 32ll * tile0
                                 : reinterpret_cast<const u32x4 *>(d.Bf) + (long long)t * d.sBf + (long long)tile0 * (16ll * G2);
    const uint32_t boff = TOEP ? 16u * ((uint32_t)((l >> 2) & 1) * (uint32_t)d.ecols + 4u * ((l >> 3) & 7) + (l & 3))
                               : 16u * ((uint32_t)(w * GB) * 256u + l);   // bytes
    const uint32_t tile_b = TOEP ? 512u : 256u * G2;                      // bytes per tile
    uint32_t ug[GB], eld[GB];
    const uint32_t ps = TOEP ? 16u * ((uint32_t)d.ecols << (d.gsh - 3)) : 1024u;
    {
        const int ws = __builtin_amdgcn_readfirstlane(w);
#pragma unroll
        for (int gb = 0; gb < GB; ++gb) {
            const int blk = GB * ws + gb;
            eld[gb] = TOEP ? (uint32_t)((16 * blk) >> d.gsh) : 0u;
            ug[gb] = TOEP ? 16u * ((uint32_t)((2 * blk) & ((1 << (d.gsh - 3)) - 1)) * (uint32_t)d.ecols + (uint32_t)d.ehalo - eld[gb])
                          : (uint32_t)gb * 4096u;
        }
    }
#define FUSED_BLD(off_, gb_, c_) (TOEP ? ldg<u32x4>(bt, (off_) + ug[gb_] + (c_) * ps) : ldg_nt<u32x4>(bt, (off_) + (gb_) * 4096 + (c_) * 1024))
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
        for (int gb = 0; gb < GB; ++gb) {
            // (the columns m < ld of block ld - first tile of the dictionary only - come from the halo)
            const int cc = 4 * ((l >> 3) & 7) + (l & 3) - (int)eld[gb];
            const uint32_t fo = boff - ((TOEP && tile0 == 0 && cc < 0) ? 8u * eld[gb] * (eld[gb] - 1u) : 0u);
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4 *>(rdst + c * 8 * ROWB + gb * 128) = FUSED_BLD(fo, gb, c);
        }
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
            const float2 v2 = make_float2(admm_v2(prm, ev2[s].x, ex[s].x, xs.x), admm_v2(prm, ev2[s].y, ex[s].y, xs.y));
            // X <- iK1 (V1 + rho Y + subY + V2 + rho C + rho Xs)      (:38-40)
            float rh = eid[s], rl = 0.f;            // 1 / (Omega + 2 rho): read (one float), or formed here from Omega (two)
            if (d.inv_is_omega) admm_invd(prm, eid[s], rh, rl);
            const float2 x = make_float2(admm_x2(prm, ev1[s].x, ey[s].x, esy[s].x, v2.x, xs.x, rh, rl),
                                         admm_x2(prm, ev1[s].y, ey[s].y, esy[s].y, v2.y, xs.y, rh, rl));
            const float2 kk = make_float2(admm_k(prm, x.x, v2.x), admm_k(prm, x.y, v2.y));        // (:43)
            const float2 v1 = make_float2(admm_v1(prm, ev1[s].x, ey[s].x, x.x), admm_v1(prm, ev1[s].y, ey[s].y, x.y));   // (:64)
            if (!(DBG & 2)) { stg_nt2(V2t, ix, v2); stg_nt2(Xt, ix, x); stg_nt2(V1t, ix, v1); }
            const float2 zn = make_float2(admm_z(prm, x.x, v1.x), admm_z(prm, x.y, v1.y));
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
            for (int c = 0; c < 4; ++c) { rf[0][c] = FUSED_BLD(noff, 0, c); if (GB > 1) rf[1][c] = FUSED_BLD(noff, (GB > 1 ? 1 : 0), c); }
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
                    // (round 5: the six products of a tile's block are summed in fresh accumulators and added to the running sums
                    //  ONCE - see fused_pass64_kernel, ACC = 1)
                    {
                    f32x4 tr = mma(bf[0], k0, f32x4{0.f, 0.f, 0.f, 0.f}), ti = mma(bf[0], k2, f32x4{0.f, 0.f, 0.f, 0.f});
                    tr = mma(bf[0], k1, tr); ti = mma(bf[0], k3, ti);
                    tr = mma(bf[1], k0, tr); ti = mma(bf[1], k2, ti);
                    k0 = *reinterpret_cast<const u32x4 *>(kp + 4096); k1 = *reinterpret_cast<const u32x4 *>(kp + 5120);   // -kr
                    tr = mma(bf[2], k2, tr); ti = mma(bf[2], k0, ti);
                    tr = mma(bf[2], k3, tr); ti = mma(bf[2], k1, ti);
                    tr = mma(bf[3], k2, tr); ti = mma(bf[3], k0, ti);
                    pr[gb][n2] += tr; pi[gb][n2] += ti;
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (else the fragment reads of all four n-blocks are hoisted: spills)
                }
                if (!(DBG & 8)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        *reinterpret_cast<u32x4 *>(rdst + c * 8 * ROWB + gb * 128) = rf[gb & 1][c];
                        if (gb + 2 < GB) rf[gb & 1][c] = FUSED_BLD(noff, (gb + 2 < GB ? gb + 2 : 0), c);
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
#undef FUSED_BLD
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


// ================================================================================================================
// v2: the pass for a block-Toeplitz dictionary of block height 64 (G2 = 64 L, L = 2 GB delays).
//
// Block ld of the dictionary is block 0 shifted right by ld columns, so the LDS tile of 32 columns is not G2 x 32 entries
// (128 KiB) but the 64 x 39 window of block 0 they all come from (20 KiB): a fragment of block ld, column m, is read at
// window column m - ld.  Both products are the ones of fused_pass_kernel - same fragments, same order, same bits when the
// leading columns m < ld of the delayed blocks are zero - only the LDS addresses differ.  What the 108 KiB buy:
//   * the element-wise operands of the NEXT tile (X, V1, V2, subY, 1/D: 72 KiB per tile) are fetched into an LDS staging
//     area while phase B runs (through the 32 registers the 128-KiB refill of fused_pass_kernel needed), each wave for its
//     own block; in fused_pass_kernel those loads sit between the two product phases with nothing to hide them (1.30 ms
//     of products + 0.69 ms of exposed element-wise traffic = 1.98 ms per pass, measured by switching parts off:
//     tools/pass_breakdown.py);
//   * the svt argument Z = X - V1 / rho of Y = (I - Q) Z is formed from the staged X and V1 of the four waves of a column
//     half instead of being written by one pass and read by the next (- 1.07 GB per pass with convergence_error; without
//     it the Gram of the stored Z still needs the write);
//   * the window of the next tile arrives the same way into a second buffer.
// (LDS-direct loads - global_load_lds_dwordx4, no registers at all - were tried first and are slower here: every one of
//  the 12 per wave and tile holds its wave for 250-300 cycles at issue, 3000 cycles per tile, wherever in the tile they are
//  issued: tools/probe/lds_dma_probe.hip checks their addressing, profiles/r03_pass64_sections.txt has the timings.)
// The leading columns (m < ld of block ld: 28 of the 4096 x 8 column-blocks) are outside the Toeplitz part.  They enter
// as fp32 corrections: XsD = (A S) Delta is added to Xs in the first tile (xs_delta, formed with the (A S) fragments), the
// first L - 1 columns of k are stored, and the sum of the partial sums adds k(:, m) conj(B(g, m)) (reduce_parts_delta_kernel).
template <int GB, int DBG, int ACC = 0>
__global__ __launch_bounds__(512, 1) void fused_pass64_kernel(FusedDesc d)
{
    constexpr int G2 = 128 * GB;
    constexpr int KSH = G2 / 64;               // 32-wide k-steps per g-half
    constexpr int EPL = 40 * 128;              // bytes of one plane of the window: 40 columns x 8 octets x 16 B
    constexpr int EBUF = 4 * EPL;              // 20 KiB
    // staged element-wise operands of one wave (its 16 rows n x 16 columns m of the tile): X, V1, V2, subY as [column][n],
    // 128 bytes per column, then 1/D with 80 bytes per column
    constexpr int SCOL = 128, SFLD = 16 * SCOL, SINV = 4 * SFLD, ICOL = 80, STW = SINV + 16 * ICOL;
    constexpr int STG0 = 2 * EBUF + 24576;
    // the svt argument Z = X - V1 / rho of the tile as A-operand fragments of Y = (I - Q) Z (16 KiB): [column half kh][k-step ks]
    // [plane][q][column c16] 16-byte chunks (8 consecutive rows n' = 32 ks + 8 q .. of one column, one f16 plane)
    constexpr int ZF0 = STG0 + 8 * STW;
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char *xch = lds + 2 * EBUF;       // 24 KiB: the k fragments (6 planes)

    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int t = (slot / d.parts) * 8 + xcd;  // the column ranges of one problem run on ONE XCD: (A S) stays in its L2
    if (t >= d.batch) return;
    const int part = slot % d.parts;
    const int tpw = (d.M / 32) / d.parts;
    const int tile0 = part * tpw;

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, q = l >> 4, c16 = l & 15;
    const int nb = w & 3, kh = w >> 2;
    const int ws = __builtin_amdgcn_readfirstlane(w);
    unsigned char *stag = lds + STG0 + ws * STW;

    const TrialParams prm = d.prm[t];
    const int eb = fscale_exp(d.bmax[(long long)t * d.sbmax]), ew = fscale_exp(d.wmax[t]);
    const int ek = fscale_exp(d.kmax_prev[t]) - d.kback;       // (see fused_pass_kernel)
    const float sxs = ldexpf(1.f, -(eb + ew)), sk = ldexpf(1.f, ek), sp = ldexpf(1.f, -(eb + ek));
    // (the coefficients of the element-wise updates: admm_* of common.h, two floats each)

    f32x4 pr[GB][4], pi[GB][4];
#pragma unroll
    for (int gb = 0; gb < GB; ++gb)
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) { pr[gb][n2] = f32x4{0.f, 0.f, 0.f, 0.f}; pi[gb][n2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float kmx = 0.f, xmx = 0.f, v1mx = 0.f, zmx = 0.f, v2mx = 0.f;

    float2 *const Xt = d.X + (long long)t * d.snm, *const V1t = d.V1 + (long long)t * d.snm, *const V2t = d.V2 + (long long)t * d.snm;
    const float2 *const sYt = d.subY + (long long)t * d.snm;
    const float *const iDt = d.invD + (long long)t * d.snm;
    float2 *const Zot = d.Zout ? d.Zout + (long long)t * d.snm : nullptr;
    float2 *const Yot = d.Yout ? d.Yout + (long long)t * d.snm : nullptr;
    const u32x4 *const wqt = reinterpret_cast<const u32x4 *>(d.Wqp) + (long long)t * 2048;
    const uint32_t wqoff = 16u * (nb * 256 + l);                                                              // bytes
    const float sy = ldexpf(1.f, -(fscale_exp(d.zmax_in[t]) + 13));
    const uint32_t ebase = 8u * (16 * nb + c16 + 64 * (16 * kh + 4 * q));     // bytes; + 512 (m0 + s)
    const u32x4 *const ast = reinterpret_cast<const u32x4 *>(d.ASp) + (long long)t * d.sAS;
    const uint32_t aoff = 16u * ((uint32_t)(kh * KSH) * 1024u + nb * 256 + l);      // bytes
    const uint4 *const Et = d.Ec + (long long)t * d.sEc;
    const float2 *const XsDt = d.XsD + (long long)t * 2048;
    float2 *const Kft = d.Kf + (long long)t * 512;
    const uint32_t epl = 128u * (uint32_t)d.ecols;                            // bytes per plane of the image

    // ---- the 12 pieces of 1 KiB (16 bytes per lane) a wave moves per tile, through registers:
    //   0..2   window of the tile: per plane the 5 KiB at column 32 T of the image, 20 pieces, wave w takes w, w + 8, w + 16
    //   3..10  X, V1 (3: X j = 0, 4: V1 j = 0, 5: X j = 1, 6: V1 j = 1), V2 (7, 8), subY (9, 10) of this wave's block:
    //          lane -> rows n = 2 (l & 7), + 1 of column (l >> 3) + 8 j
    //   11     1/D: lane -> rows 4 (l & 3) .. + 3 of column l >> 2
    // When the V1 piece of a column half goes to LDS, the X piece of the same half is still in its register slot: the wave
    // forms Z = X - V1 / rho of its block there and writes it as split-f16 fragments - the four waves of a column half
    // read fragments (8 reads per tile and lane) instead of every one of them reading and converting X and V1 (32 reads)
    const uint32_t so = 128u * nb + 16u * (l & 7) + 512u * (16 * kh + (l >> 3));
    // rows n' = 16 nb + 2 (l & 7), + 1: k-step nb >> 1, q = 2 (nb & 1) + ((l & 7) >> 2), halves 2 (l & 3), + 1 of the chunk
    const float sz = ldexpf(1.f, fscale_exp(d.zmax_in[t]));
#define F64_FLD(pc_) ((pc_) < 7 ? ((pc_) - 3) & 1 : 2 + (((pc_) - 7) >> 1))
#define F64_J(pc_) ((pc_) < 7 ? ((pc_) - 3) >> 1 : ((pc_) - 7) & 1)
#define F64_LOAD(pc_, T_, dst_)                                                                                              \
    {                                                                                                                        \
        if ((pc_) < 3) {                                                                                                     \
            const int e_ = ws + 8 * (pc_) - ((pc_) == 2 && ws >= 4 ? 8 : 0);    /* (waves 4-7 have two pieces: the second again) */ \
            if (!(DBG & 8)) {                                                                                                \
                const int p_ = e_ / 5, j_ = e_ - 5 * p_;                                                                     \
                dst_ = ldg_nt<u32x4>(Et, (uint32_t)p_ * epl + 4096u * (uint32_t)(T_) + 1024u * j_ + 16u * l);               \
            }                                                                                                                \
        } else if (!(DBG & 2)) {                                                                                             \
            const uint32_t o_ = so + 16384u * (uint32_t)(T_) + 4096u * F64_J(pc_);                                           \
            if ((pc_) < 11 && F64_FLD(pc_) == 0) dst_ = ldg_nt<u32x4>(Xt, o_);                                               \
            if ((pc_) < 11 && F64_FLD(pc_) == 1) dst_ = ldg_nt<u32x4>(V1t, o_);                                              \
            if ((pc_) < 11 && F64_FLD(pc_) == 2) dst_ = ldg_nt<u32x4>(V2t, o_);                                              \
            if ((pc_) < 11 && F64_FLD(pc_) == 3) dst_ = ldg_nt<u32x4>(sYt, o_);                                              \
            if ((pc_) == 11) {      /* (lane offsets of the one 1/D piece are formed here, not kept in registers across the tile) */ \
                int lo_ = l;                                                                                                 \
                asm volatile("" : "+v"(lo_));                                                                                \
                dst_ = ldg_nt<u32x4>(iDt, 64u * nb + 16u * (lo_ & 3) + 256u * (16 * kh + (lo_ >> 2)) + 8192u * (uint32_t)(T_)); \
            }                                                                                                                \
        }                                                                                                                    \
    }
    // (slot of piece pc: rf[(pc >> 2) & 1][pc & 3]; the X piece of V1 piece pc is pc - 1)
#define F64_STORE(pc_, buf_, src_)                                                                                           \
    {                                                                                                                        \
        if ((pc_) < 3) {                                                                                                     \
            const int e_ = ws + 8 * (pc_) - ((pc_) == 2 && ws >= 4 ? 8 : 0);                                                 \
            if (!(DBG & 8)) {                                                                                                \
                const int p_ = e_ / 5, j_ = e_ - 5 * p_;                                                                     \
                *reinterpret_cast<u32x4 *>(lds + (buf_) * EBUF + p_ * EPL + j_ * 1024 + 16 * l) = src_;                      \
            }                                                                                                                \
        } else if (!(DBG & 2)) {                                                                                             \
            if ((pc_) == 11) {                                                                                               \
                int lo_ = l;                                                                                                 \
                asm volatile("" : "+v"(lo_));                                                                                \
                *reinterpret_cast<u32x4 *>(stag + SINV + (lo_ >> 2) * ICOL + (lo_ & 3) * 16) = src_;                         \
            }                                                                                                                \
            else {      /* (lane offset formed here, not kept in a register across the tile) */                              \
                int ls_ = l;                                                                                                 \
                asm volatile("" : "+v"(ls_));                                                                                \
                *reinterpret_cast<u32x4 *>(stag + (ls_ >> 3) * SCOL + (ls_ & 7) * 16 + F64_FLD(pc_) * SFLD + F64_J(pc_) * 8 * SCOL) = src_; \
            }                                                                                                                \
            if ((pc_) == 4 || (pc_) == 6) {                                                                                  \
                const f32x4 xv_ = __builtin_bit_cast(f32x4, rf[(((pc_) - 1) >> 2) & 1][((pc_) - 1) & 3]);                    \
                const f32x4 vv_ = __builtin_bit_cast(f32x4, src_);                                                           \
                _Float16 h0_, l0_, h1_, l1_;                                                                                 \
                fsplit(admm_z(prm, xv_[0], vv_[0]) * sz, h0_, l0_); fsplit(admm_z(prm, xv_[2], vv_[2]) * sz, h1_, l1_);      \
                typedef _Float16 half2_ __attribute__((ext_vector_type(2)));                                                 \
                int lz_ = l;                                                                                                 \
                asm volatile("" : "+v"(lz_));                                                                                \
                unsigned char *z_ = lds + ZF0 + ((kh * 2 + (nb >> 1)) * 16 + 2 * (nb & 1) + ((lz_ & 7) >> 2)) * 256 +       \
                                    (lz_ >> 3) * 16 + 4 * (lz_ & 3) + F64_J(pc_) * 128;                                      \
                *reinterpret_cast<half2_ *>(z_) = half2_{h0_, h1_}; *reinterpret_cast<half2_ *>(z_ + 1024) = half2_{l0_, l1_}; \
                fsplit(admm_z(prm, xv_[1], vv_[1]) * sz, h0_, l0_); fsplit(admm_z(prm, xv_[3], vv_[3]) * sz, h1_, l1_);      \
                *reinterpret_cast<half2_ *>(z_ + 2048) = half2_{h0_, h1_}; *reinterpret_cast<half2_ *>(z_ + 3072) = half2_{l0_, l1_}; \
            }                                                                                                                \
        }                                                                                                                    \
    }
    u32x4 rf[2][4];
    {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) F64_LOAD(pc, tile0, rf[pc >> 2][pc & 3])
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) F64_STORE(pc, 0, rf[pc >> 2][pc & 3])
#pragma unroll
        for (int pc = 8; pc < 12; ++pc) F64_LOAD(pc, tile0, rf[0][pc & 3])
#pragma unroll
        for (int pc = 8; pc < 12; ++pc) F64_STORE(pc, 0, rf[0][pc & 3])
    }
    __syncthreads();

    // phase-A window column of this lane before the delay: block ld = kh GB + (ks >> 1) sits ld columns to the left
    const int cA = c16 + 7 - kh * GB;
    const int cB0 = 4 * q + (c16 >> 2) + 7;    // phase B: window column of this lane's row 4 q + (c16 >> 2), before the delay
    // Z fragments of Y = (I - Q) Z for this lane: column 16 kh + c16, rows n' = 32 ks + 8 q .. + 7
    const unsigned char *const zsrc = lds + ZF0 + (kh * 2 * 16 + q) * 256 + c16 * 16;      // + ks 4096 + plane 1024
    // the fragments of (I - Q)^T (the same for every tile; no registers to keep them): requested behind the last products of a
    // tile, so that they arrive while the wave waits at the barrier
    u32x4 wq[2][4];
#define F64_WQLOAD()                                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                        \
    {                                                                                                                        \
        uint32_t wo_ = wqoff + ks * 16384;          /* (opaque: else a 64-bit address per plane is built and spilled) */     \
        asm volatile("" : "+v"(wo_));                                                                                        \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) wq[ks][p] = ldg<u32x4>(wqt, wo_ + p * 1024);                           \
    }
    F64_WQLOAD()
    long long t_wait = 0, t_bar = 0, t_ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_start = (DBG & 16) ? (long long)__builtin_readcyclecounter() : 0;
    for (int i = 0; i < tpw; ++i) {
        const int m0 = (tile0 + i) * 32;
        const unsigned char *ebuf = lds + (i & 1) * EBUF;
        long long tp = (DBG & 16) ? (long long)__builtin_readcyclecounter() : 0;
        // ================= Y^T(block) = Z^T Wq^T for this wave's element-wise block: A operand = Z^T (lane = column m, 8
        // consecutive rows n' per k-step), Z = X - V1 / rho formed from the staged operands of the four waves of this
        // column half - the svt argument is neither read from memory nor (with convergence_error) written to it;
        // B operand = the fragments of Wq^T = (I - Q)^T; two k-steps of 32 rows
        float2 ey[4];
        {
            f32x4 yr = f32x4{0.f, 0.f, 0.f, 0.f}, yi = yr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 zf[4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    zf[p] = (DBG & 2) ? u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}
                                      : *reinterpret_cast<const u32x4 *>(zsrc + ks * 4096 + p * 1024);
                const u32x4 nwi_h = negu(wq[ks][2]), nwi_l = negu(wq[ks][3]);
                yr = mma(zf[0], wq[ks][0], yr); yi = mma(zf[0], wq[ks][2], yi);
                yr = mma(zf[0], wq[ks][1], yr); yi = mma(zf[0], wq[ks][3], yi);
                yr = mma(zf[1], wq[ks][0], yr); yi = mma(zf[1], wq[ks][2], yi);
                yr = mma(zf[2], nwi_h, yr); yi = mma(zf[2], wq[ks][0], yi);
                yr = mma(zf[2], nwi_l, yr); yi = mma(zf[2], wq[ks][1], yi);
                yr = mma(zf[3], nwi_h, yr); yi = mma(zf[3], wq[ks][0], yi);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) ey[s] = make_float2(yr[s] * sy, yi[s] * sy);
        }
        __builtin_amdgcn_sched_barrier(0);      // (else the first (A S) fragments of phase A are requested above these products: spills)
        if (DBG & 16) { const long long tq = __builtin_readcyclecounter(); t_ph[3] += tq - tp; tp = tq; }
        // ================= phase A: Xs^T(tile) = B^T (A S)^T, this wave: n-block nb, g-half kh, both m-blocks
        f32x4 ar[2], ai[2];
        ar[0] = ar[1] = ai[0] = ai[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(DBG & 1)) {
            u32x4 wr[2][4], bfb[4];
            uint32_t ao = aoff;
            asm volatile("" : "+v"(ao));
            int cK = cA;
#pragma unroll
            for (int p = 0; p < 4; ++p) wr[0][p] = ldg<u32x4>(ast, ao + p * 1024);
#pragma unroll
            for (int st = 0; st < 2 * KSH; ++st) {
                const int ks = st >> 1, mb = st & 1;
                if (mb == 0 && ks + 1 < KSH) {
                    ao += 16384;
                    asm volatile("" : "+v"(ao));
#pragma unroll
                    for (int p = 0; p < 4; ++p) wr[(ks + 1) & 1][p] = ldg<u32x4>(ast, ao + p * 1024);
                }
                {
                    // (the column is advanced opaquely per delay block: computed up front, the addresses of all (block, octet)
                    //  pairs are loop invariants of the tile loop and the compiler keeps - and spills - them)
                    if (mb == 0 && (ks & 1) == 0) {
                        if (ks > 0) cK -= 1;
                        asm volatile("" : "+v"(cK));
                    }
                    const unsigned char *arow = ebuf + cK * 128 + 16 * ((4 * (ks & 1) + q) ^ eswz(cK)) + mb * 2048;
#pragma unroll
                    for (int p = 0; p < 4; ++p) bfb[p] = *reinterpret_cast<const u32x4 *>(arow + p * EPL);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 *wf = wr[ks & 1];
                const u32x4 nwi_h = negu(wf[2]), nwi_l = negu(wf[3]);
                // re += Br Wr - Bi Wi ; im += Br Wi + Bi Wr   (h h + h l + l h each)
                ar[mb] = mma(bfb[0], wf[0], ar[mb]); ai[mb] = mma(bfb[0], wf[2], ai[mb]);
                ar[mb] = mma(bfb[0], wf[1], ar[mb]); ai[mb] = mma(bfb[0], wf[3], ai[mb]);
                ar[mb] = mma(bfb[1], wf[0], ar[mb]); ai[mb] = mma(bfb[1], wf[2], ai[mb]);
                ar[mb] = mma(bfb[2], nwi_h, ar[mb]); ai[mb] = mma(bfb[2], wf[0], ai[mb]);
                ar[mb] = mma(bfb[2], nwi_l, ar[mb]); ai[mb] = mma(bfb[2], wf[1], ai[mb]);
                ar[mb] = mma(bfb[3], nwi_h, ar[mb]); ai[mb] = mma(bfb[3], wf[0], ai[mb]);
            }
        }
        if (DBG & 16) { const long long tq = __builtin_readcyclecounter(); t_ph[0] += tq - tp; tp = tq; }
        // the two g-halves meet: wave (nb, kh) keeps m-block kh and hands m-block 1 - kh to wave (nb, 1 - kh) - through the
        // window buffer that is not in use (the next window arrives there during phase B), so that the k fragments below
        // need not wait for these reads
        unsigned char *const pex = lds + ((i + 1) & 1) * EBUF;
        {
            const f32x4 sr = kh ? ar[0] : ar[1], si = kh ? ai[0] : ai[1];
            f32x4 *x4 = reinterpret_cast<f32x4 *>(pex);
            const int dw = nb + 4 * (1 - kh);
            x4[(dw * 2 + 0) * 64 + l] = sr;
            x4[(dw * 2 + 1) * 64 + l] = si;
        }
        // the element-wise operands of this wave's block (n = 16 nb + c16, m = m0 + 16 kh + 4 q + s), staged by this wave
        float2 ex[4], ev1[4], ev2[4], esy[4];
        float eid[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (DBG & 2) { ex[s] = ev1[s] = ev2[s] = esy[s] = make_float2(1.f, 1.f); eid[s] = 1.f; continue; }
            const unsigned char *sp_ = stag + (4 * q + s) * SCOL + c16 * 8;
            ex[s] = *reinterpret_cast<const float2 *>(sp_);              ev1[s] = *reinterpret_cast<const float2 *>(sp_ + SFLD);
            ev2[s] = *reinterpret_cast<const float2 *>(sp_ + 2 * SFLD);  esy[s] = *reinterpret_cast<const float2 *>(sp_ + 3 * SFLD);
            eid[s] = *reinterpret_cast<const float *>(stag + SINV + (4 * q + s) * ICOL + c16 * 4);
        }
        __syncthreads();
        if (DBG & 16) { const long long tq = __builtin_readcyclecounter(); t_ph[4] += tq - tp; tp = tq; }
        f32x4 xr = kh ? ar[1] : ar[0], xi = kh ? ai[1] : ai[0];
        {
            const f32x4 *x4 = reinterpret_cast<const f32x4 *>(pex);
            const f32x4 orr = x4[(w * 2 + 0) * 64 + l], oi = x4[(w * 2 + 1) * 64 + l];
            xr += orr; xi += oi;
        }
        half4 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t ix = ebase + 512u * (uint32_t)(m0 + s);
            float2 xs = make_float2(xr[s] * sxs, xi[s] * sxs);
            const bool lead = m0 == 0 && kh == 0 && 4 * q + s < 2 * GB - 1;      // a leading column: + (A S) Delta
            if (lead) {                          // (m0 = 0, kh = 0: ix is 8 (n + 64 m))
                const float2 d0 = ldg<float2>(XsDt, ix), d1 = ldg<float2>(XsDt, ix + 4096), d2 = ldg<float2>(XsDt, ix + 8192),
                             d3 = ldg<float2>(XsDt, ix + 12288);
                xs.x += (d0.x + d1.x) + (d2.x + d3.x); xs.y += (d0.y + d1.y) + (d2.y + d3.y);
            }
            // V2 <- (1 - cc)(V2 - rho (X - Xs))                       (:61 + :65, C == -V2)
            const float2 v2 = make_float2(admm_v2(prm, ev2[s].x, ex[s].x, xs.x), admm_v2(prm, ev2[s].y, ex[s].y, xs.y));
            // X <- iK1 (V1 + rho Y + subY + V2 + rho C + rho Xs)      (:38-40)
            float rh = eid[s], rl = 0.f;            // 1 / (Omega + 2 rho): read (one float), or formed here from Omega (two)
            if (d.inv_is_omega) admm_invd(prm, eid[s], rh, rl);
            const float2 x = make_float2(admm_x2(prm, ev1[s].x, ey[s].x, esy[s].x, v2.x, xs.x, rh, rl),
                                         admm_x2(prm, ev1[s].y, ey[s].y, esy[s].y, v2.y, xs.y, rh, rl));
            const float2 kk = make_float2(admm_k(prm, x.x, v2.x), admm_k(prm, x.y, v2.y));        // (:43)
            const float2 v1 = make_float2(admm_v1(prm, ev1[s].x, ey[s].x, x.x), admm_v1(prm, ev1[s].y, ey[s].y, x.y));   // (:64)
            if (!(DBG & 2)) { stg_nt2(V2t, ix, v2); stg_nt2(Xt, ix, x); stg_nt2(V1t, ix, v1); }
            const float2 zn = make_float2(admm_z(prm, x.x, v1.x), admm_z(prm, x.y, v1.y));
            if (Zot) stg_nt2(Zot, ix, zn);
            if (Yot) stg(Yot, ix, ey[s]);
            if (lead) stg(Kft, ix, kk);
            v2mx = fmaxf(v2mx, fmaxf(fabsf(v2.x), fabsf(v2.y)));
            xmx = fmaxf(xmx, fmaxf(fabsf(x.x), fabsf(x.y)));
            v1mx = fmaxf(v1mx, fmaxf(fabsf(v1.x), fabsf(v1.y)));
            zmx = fmaxf(zmx, fmaxf(fabsf(zn.x), fabsf(zn.y)));
            kmx = fmaxf(kmx, fmaxf(fabsf(kk.x), fabsf(kk.y)));
            _Float16 h, lo;
            fsplit(kk.x * sk, h, lo); kf[0][s] = h; kf[1][s] = lo;
            fsplit(kk.y * sk, h, lo); kf[2][s] = h; kf[3][s] = lo;
        }
        if (DBG & 16) { const long long tq = __builtin_readcyclecounter(); t_ph[5] += tq - tp; tp = tq; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<half4 *>(xch + ((nb * 6 + p) * 64 + l) * 16 + kh * 8) = kf[p];
        // planes 4, 5: -k_re (the imaginary part of conj(B) k needs it; negating per product costs registers)
        *reinterpret_cast<half4 *>(xch + ((nb * 6 + 4) * 64 + l) * 16 + kh * 8) = -kf[0];
        *reinterpret_cast<half4 *>(xch + ((nb * 6 + 5) * 64 + l) * 16 + kh * 8) = -kf[1];
        __syncthreads();
        if (DBG & 16) { const long long tq = __builtin_readcyclecounter(); t_ph[7] += tq - tp; tp = tq; }
        // the next tile (the last one is fetched again: unconditional loads): its window goes into the other buffer, its
        // element-wise operands into this wave's staging area (every wave has read what it needs of it: Y above, the operands
        // before the first barrier).  One piece is requested behind each product group of phase B and written to LDS six
        // groups later (eight register slots): requested together, the 8 KiB per wave of all eight waves queue up at the CU's
        // 64-byte-per-clock load path and every wave waits 1200 cycles before its first product (tools/pass_breakdown.py).
        const int tn = tile0 + min(i + 1, tpw - 1);
        if (DBG & 16) { const long long tq = __builtin_readcyclecounter(); t_ph[1] += tq - tp; tp = tq; }
        // ================= phase B: P^T += conj(B)(g, tile) k^T(tile, :), this wave: g in [16 GB w, 16 GB (w + 1))
        if (!(DBG & 4)) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            auto *lbase = (__attribute__((address_space(3))) unsigned char *)lds;
#pragma unroll
            for (int gb = 0; gb < GB; ++gb) {
                // rows 16 (GB w + gb) ..: delay ld, octet pair jp of block 0; lane: window column of row 4 q + (c16 >> 2)
                const int blk = GB * ws + gb, ld = blk >> 2, jp = 2 * (blk & 3);
                int cB = cB0 - ld;
                asm volatile("" : "+v"(cB));                 // (as in phase A: no address per block kept across the tile loop)
                const uint32_t tr0 = (uint32_t)((i & 1) * EBUF) + cB * 128 + 16 * ((jp + ((c16 >> 1) & 1)) ^ eswz(cB)) + (c16 & 1) * 8;
                u32x4 bf[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t o_ = tr0 + p * EPL;
                    const u32x2 lo_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lbase + o_)));
                    const u32x2 hi_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lbase + o_ + 2048)));
                    bf[p] = u32x4{lo_.x, lo_.y, hi_.x, hi_.y};
                }
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) {
                    const unsigned char *kp = xch + (n2 * 6 * 64 + l) * 16;
                    u32x4 k0 = *reinterpret_cast<const u32x4 *>(kp), k1 = *reinterpret_cast<const u32x4 *>(kp + 1024);
                    const u32x4 k2 = *reinterpret_cast<const u32x4 *>(kp + 2048), k3 = *reinterpret_cast<const u32x4 *>(kp + 3072);
                    // re += Br kr + Bi ki ; im += Br ki - Bi kr
                    // ACC = 0: every product accumulates straight into the running sums (rounds 2-4).  ACC = 1 (round 5, default): the
                    // six products of a tile's block are summed in fresh accumulators and added to the running sums ONCE - the
                    // running sum, 32 tiles long, is rounded once per tile instead of six times: 1.0e-7 of the 1.73e-7 rms dNMSE
                    // of round 4 was this.  +3 % kernel time (the adds wait for the last product of their
                    // chain; measured slower still: the two chains one after the other, 795 vs 798 channel-estimates/s, and the adds
                    // deferred behind the next block's first products, 803 vs 815 with 4 more spilled registers)
                    if constexpr (ACC == 0) {
                        pr[gb][n2] = mma(bf[0], k0, pr[gb][n2]); pi[gb][n2] = mma(bf[0], k2, pi[gb][n2]);
                        pr[gb][n2] = mma(bf[0], k1, pr[gb][n2]); pi[gb][n2] = mma(bf[0], k3, pi[gb][n2]);
                        pr[gb][n2] = mma(bf[1], k0, pr[gb][n2]); pi[gb][n2] = mma(bf[1], k2, pi[gb][n2]);
                        k0 = *reinterpret_cast<const u32x4 *>(kp + 4096); k1 = *reinterpret_cast<const u32x4 *>(kp + 5120);   // -kr
                        pr[gb][n2] = mma(bf[2], k2, pr[gb][n2]); pi[gb][n2] = mma(bf[2], k0, pi[gb][n2]);
                        pr[gb][n2] = mma(bf[2], k3, pr[gb][n2]); pi[gb][n2] = mma(bf[2], k1, pi[gb][n2]);
                        pr[gb][n2] = mma(bf[3], k2, pr[gb][n2]); pi[gb][n2] = mma(bf[3], k0, pi[gb][n2]);
                    } else if constexpr (ACC == 1) {
                        f32x4 tr = mma(bf[0], k0, f32x4{0.f, 0.f, 0.f, 0.f}), ti = mma(bf[0], k2, f32x4{0.f, 0.f, 0.f, 0.f});
                        tr = mma(bf[0], k1, tr); ti = mma(bf[0], k3, ti);
                        tr = mma(bf[1], k0, tr); ti = mma(bf[1], k2, ti);
                        k0 = *reinterpret_cast<const u32x4 *>(kp + 4096); k1 = *reinterpret_cast<const u32x4 *>(kp + 5120);   // -kr
                        tr = mma(bf[2], k2, tr); ti = mma(bf[2], k0, ti);
                        tr = mma(bf[2], k3, tr); ti = mma(bf[2], k1, ti);
                        tr = mma(bf[3], k2, tr); ti = mma(bf[3], k0, ti);
                        {       // (this block's prefetch traffic is issued while the last products drain; then the sums)
                            const int grp = 4 * gb + n2;
                            if (grp >= 6 && grp - 6 < 12) F64_STORE(grp - 6, (i + 1) & 1, rf[((grp - 6) >> 2) & 1][(grp - 6) & 3])
                            if (grp < 12) F64_LOAD(grp, tn, rf[(grp >> 2) & 1][grp & 3])
                        }
                        pr[gb][n2] += tr; pi[gb][n2] += ti;
                    }
                    if constexpr (ACC != 1) {
                        const int grp = 4 * gb + n2;
                        if (grp >= 6 && grp - 6 < 12) F64_STORE(grp - 6, (i + 1) & 1, rf[((grp - 6) >> 2) & 1][(grp - 6) & 3])
                        if (grp < 12) F64_LOAD(grp, tn, rf[(grp >> 2) & 1][grp & 3])
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (else the fragment reads of all four n-blocks are hoisted: spills)
                }
            }
        }
        {
            constexpr int NG = (DBG & 4) ? 0 : 4 * GB;      // product groups that ran; what is left of the 12 pieces:
#pragma unroll
            for (int pc = (NG > 6 ? NG - 6 : 0); pc < 12; ++pc) {
                if (pc >= NG) F64_LOAD(pc, tn, rf[(pc >> 2) & 1][pc & 3])
                F64_STORE(pc, (i + 1) & 1, rf[(pc >> 2) & 1][pc & 3])
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        F64_WQLOAD()
        long long tw0 = 0;
        if (DBG & 16) { tw0 = __builtin_readcyclecounter(); t_ph[2] += tw0 - tp; }
        __syncthreads();                        // window and operands of the next tile in place for every wave; k fragments dead
        if (DBG & 16) t_bar += __builtin_readcyclecounter() - tw0;
    }
    if ((DBG & 16) && l == 0) {                 // (timing experiment: cycles per section, summed over the waves of the trial)
        unsigned long long *cnt = reinterpret_cast<unsigned long long *>(d.Kf + (long long)t * 512 + 448);
        atomicAdd(cnt, (unsigned long long)t_wait); atomicAdd(cnt + 1, (unsigned long long)t_bar);
        atomicAdd(cnt + 2, (unsigned long long)(__builtin_readcyclecounter() - t_start));
        for (int z = 0; z < 9; ++z) atomicAdd(cnt + 3 + z, (unsigned long long)t_ph[z]);
    }
#undef F64_LOAD
#undef F64_STORE
#undef F64_FLD
#undef F64_J
#undef F64_WQLOAD
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

static void compact_dims(int M, int G2, int gt, int *ecols, int *ehalo)
{
    const int L = G2 / gt;
    *ehalo = ((L * (L - 1) / 2 + 3) / 4) * 4;               // column 0 stays 64-byte aligned
    *ecols = *ehalo + M;
}

size_t fused_bytes(int M, int G2, int nB, int batch, int parts)
{
    int ec, eh;
    compact_dims(M, G2, 16, &ec, &eh);                      // (the probe decides between the two images after this is sized)
    return rnd256((size_t)nB * (M / 32) * 16 * G2 * sizeof(uint4)) + rnd256((size_t)batch * (G2 / 32) * 1024 * sizeof(uint4)) +
           rnd256((size_t)batch * parts * 64 * G2 * sizeof(float2)) + rnd256((size_t)batch * sizeof(uint32_t)) + rnd256((size_t)batch * 2048 * sizeof(uint4)) +
           rnd256((size_t)nB * 4 * (G2 / 16) * ec * sizeof(uint4)) + 5 * rnd256((size_t)batch * 512 * sizeof(float2)) +
           rnd256((size_t)nB * G2 * 8 * sizeof(float2)) + 512;
}

// Exact probe of the block-Toeplitz property over all nB dictionaries; *gt = the smallest block height (16 .. 256, a power
// of two, G2 / gt >= 2) for which EVERY entry outside the leading columns repeats, 0 if none.  One pass over B and one
// 4-byte read-back: the caller chooses the image (and the kernel instance) from it, so the stream is synchronised here.
int fused_probe_toeplitz(jstsp_ctx *ctx, Arena &ar, const float2 *B, long long sBt, int G2, int M, int nB, int *gt)
{
    *gt = 0;
    uint32_t *flag = ar.get<uint32_t>(1);
    JSTSP_REQUIRE(flag, JSTSP_E_NOMEM, "fused pass: workspace exhausted");
    JSTSP_REQUIRE((long long)G2 * M < (1ll << 31), JSTSP_E_UNSUPPORTED, "fused pass: dictionary too large");
    uint32_t cand = 0;
    for (int c = 0; c < 5; ++c)
        if (G2 % (16 << c) == 0 && 2 * (16 << c) <= G2) cand |= 1u << c;
    const unsigned nblk = (unsigned)(((long long)G2 * M + 255) / 256);
    // every candidate on the first dictionary, then the smallest surviving one on all of them (a wrong candidate fails on
    // nearly every entry: testing all five on the whole batch is five times the traffic for nothing)
    for (int step = 0; step < (nB > 1 ? 2 : 1) && cand; ++step) {
        JSTSP_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), ctx->stream));
        if (step && sBt % 2 == 0)
            hipLaunchKernelGGL(toeplitz_verify_kernel, dim3((nblk + 3) / 4, nB), dim3(256), 0, ctx->stream, B, sBt, G2, M,
                               31 - __builtin_clz(cand), flag);
        else
            hipLaunchKernelGGL(toeplitz_probe_kernel, dim3(nblk, step ? nB : 1), dim3(256), 0, ctx->stream, B, sBt, G2, M, cand, flag);
        JSTSP_HIP(hipGetLastError());
        uint32_t bad = ~0u;
        JSTSP_HIP(hipMemcpyAsync(&bad, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        JSTSP_HIP(hipStreamSynchronize(ctx->stream));
        cand &= ~bad;
        cand &= cand ? (~cand + 1u) : 0u;                   // the lowest surviving bit
    }
    if (cand) *gt = 16 << (31 - __builtin_clz(cand));
    return 0;
}

// ---- G_B = B B^H of a block-Toeplitz dictionary from its first block row.  With B(ld Gt + g, m) = B0(g, m - ld) for m >= ld,
//      block (ld, ld') of G_B, ld <= ld', d = ld' - ld, is
//        G(0, d)  -  sum_{m < d} B(g, m) conj(B(d Gt + g', m))                 (what block (0, d) owes to its leading columns)
//                 -  sum_{u = M - ld}^{M - 1} B(g, u) conj(B(g', u - d))       (the ld last terms of the shifted sum)
//                 +  sum_{m < ld'} B(ld Gt + g, m) conj(B(ld' Gt + g', m))     (the true leading columns of both blocks)
//      so one Gt x G2 product (1 / L of the G2 x G2 one: 7 -> 1 ms at BASELINE configs[1]) and at most 3 (L - 1) fp32 terms per
//      entry give all of it; the lower block triangle is the conjugate transpose.
__global__ __launch_bounds__(256) void toeplitz_gram_kernel(const float2 *B, long long sBt, int G2, int M, int gt, const float2 *G0,
                                                            const float2 *G0lo, float2 *G, float2 *Glo)
{
    const int t = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= G2 * G2) return;
    const int r = idx % G2, c = idx / G2;
    const int ld = r / gt, g = r - ld * gt, l2 = c / gt, g2 = c - l2 * gt;
    if (ld > l2) return;                                    // written by the mirror entry
    const int d = l2 - ld;
    const float2 *b = B + (long long)t * sBt;
    const long long i0 = (long long)t * gt * G2 + g + (long long)gt * (d * gt + g2);
    // float64 throughout (G0 + G0lo is the block row to 1e-9; the corrections are exact products of fp32 numbers): with fp32
    // sums every block of a block diagonal would carry the SAME rounding error of its source entry, which adds up coherently
    // in G_A V G_B instead of averaging out (round 3: max |dNMSE| 8.1e-7 -> 1.36e-6 with that)
    double vr = (double)G0[i0].x + (G0lo ? (double)G0lo[i0].x : 0.0), vi = (double)G0[i0].y + (G0lo ? (double)G0lo[i0].y : 0.0);
    auto mac = [&](float2 x, float2 y, double sgn) {        // v += sgn x conj(y)
        vr += sgn * ((double)x.x * y.x + (double)x.y * y.y);
        vi += sgn * ((double)x.y * y.x - (double)x.x * y.y);
    };
    for (int m = 0; m < d; ++m) mac(b[g + (long long)G2 * m], b[d * gt + g2 + (long long)G2 * m], -1.0);
    for (int u = M - ld; u < M; ++u) mac(b[g + (long long)G2 * u], b[g2 + (long long)G2 * (u - d)], -1.0);
    for (int m = 0; m < l2; ++m) mac(b[r + (long long)G2 * m], b[c + (long long)G2 * m], 1.0);
    if (r == c) vi = 0.0;
    const float hr = (float)vr, hi = (float)vi;
    float2 *o = G + (long long)t * G2 * G2;
    // (diagonal blocks, ld == l2: both (r, c) and (c, r) have their own thread - one writer per entry)
    o[r + (long long)G2 * c] = make_float2(hr, hi);
    if (ld < l2) o[c + (long long)G2 * r] = make_float2(hr, -hi);
    if (Glo) {
        float2 *ol = Glo + (long long)t * G2 * G2;
        const float lr = (float)(vr - (double)hr), li = (float)(vi - (double)hi);
        ol[r + (long long)G2 * c] = make_float2(lr, li);
        if (ld < l2) ol[c + (long long)G2 * r] = make_float2(lr, -li);
    }
}

int toeplitz_gram_assemble(jstsp_ctx *ctx, const float2 *B, long long sBt, int G2, int M, int gt, int nB, const float2 *G0,
                           const float2 *G0lo, float2 *G, float2 *Glo)
{
    hipLaunchKernelGGL(toeplitz_gram_kernel, dim3((unsigned)((G2 * G2 + 255) / 256), nB), dim3(256), 0, ctx->stream, B, sBt, G2, M, gt,
                       G0, G0lo, G, Glo);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

int fused_alloc(Arena &ar, FusedWS &f, int M, int G2, int nB, int batch, int parts, int gt, int v2)
{
    f.parts = parts;
    f.sBf = (long long)(M / 32) * 16 * G2;
    f.sAS = (long long)(G2 / 32) * 1024;
    f.gt = gt;
    f.v2 = (gt == 64 && v2) ? 1 : 0;
    if (f.v2) {                                             // the window image: 4 planes x (M + 8) columns x 8 octets
        f.ecols = M + 8; f.ehalo = 7;
        f.sEc = 4ll * f.ecols * 8;
        f.Ec = ar.get<uint4>((size_t)nB * f.sEc);
        f.XsD = ar.get<float2>((size_t)batch * 2048);
        f.Kf = ar.get<float2>((size_t)batch * 512);
        f.Bdl = ar.get<float2>((size_t)nB * G2 * 8);
        JSTSP_REQUIRE(f.XsD && f.Kf && f.Bdl, JSTSP_E_NOMEM, "fused pass: workspace exhausted");
    } else if (gt) {
        compact_dims(M, G2, gt, &f.ecols, &f.ehalo);
        f.sEc = 4ll * (gt / 8) * f.ecols;
        f.Ec = ar.get<uint4>((size_t)nB * f.sEc);
    } else
        f.Bf = ar.get<uint4>((size_t)nB * f.sBf);
    f.ASp = ar.get<uint4>((size_t)batch * f.sAS);
    f.Ppart = ar.get<float2>((size_t)batch * parts * 64 * G2);
    f.ovf = ar.get<uint32_t>(batch);
    f.Wqp = ar.get<uint4>((size_t)batch * 2048);
    JSTSP_REQUIRE((f.Bf || f.Ec) && f.ASp && f.Ppart && f.ovf && f.Wqp, JSTSP_E_NOMEM, "fused pass: workspace exhausted");
    return 0;
}

int fused_pack_b(jstsp_ctx *ctx, FusedWS &f, const float2 *B, long long sBt, int G2, int M, int nB, const uint32_t *bmax)
{
    f.Bsrc = B; f.sBsrc = sBt;
    if (f.v2) {
        hipLaunchKernelGGL(pack_e2_kernel, dim3((unsigned)((std::max(M + 8, G2) * 8 + 255) / 256), nB), dim3(256), 0, ctx->stream, B, sBt, G2, M,
                           bmax, 1, f.Ec, f.sEc, f.Bdl);
        JSTSP_HIP(hipGetLastError());
        f.sBdl = sBt ? (long long)G2 * 8 : 0;
#ifdef JSTSP_FUSED_DBG_BUILD
        JSTSP_HIP(hipMemsetAsync(f.Kf, 0, 4096, ctx->stream));     // (cycle counters of the timing experiment, trial 0)
#endif
        return 0;
    }
    if (f.gt) {
        const long long n = (long long)(f.gt / 8) * f.ecols;
        hipLaunchKernelGGL(pack_e_kernel, dim3((unsigned)((n + 255) / 256), nB), dim3(256), 0, ctx->stream, B, sBt, G2, M, f.gt,
                           f.ecols, f.ehalo, bmax, 1, f.Ec, f.sEc);
        JSTSP_HIP(hipGetLastError());
        return 0;
    }
    const long long n = (long long)(M / 32) * 16 * (G2 / 4);
    hipLaunchKernelGGL(pack_bf_kernel, dim3((unsigned)((n + 255) / 256), nB), dim3(256), 0, ctx->stream, B, sBt, G2, M, bmax, 1,
                       f.Bf, f.sBf);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

int fused_pack_as(jstsp_ctx *ctx, const FusedWS &f, const float2 *W, long long sWt, int G2, int M, int batch, const uint32_t *wmax)
{
    // (v2: two more blocks per problem form (A S) Delta, the leading columns' share of Xs)
    hipLaunchKernelGGL(pack_as_kernel, dim3((G2 / 32) + (f.v2 ? 4 * (G2 / 64 - 1) : 0), batch), dim3(256), 0, ctx->stream, W, sWt, G2, wmax,
                       f.ASp, f.sAS, f.Bdl, f.sBdl, f.XsD);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

template <int GB> static int launch_fused64(jstsp_ctx *ctx, const FusedDesc &d)
{
    const size_t sh = 2 * 20480 + 24576 + 8 * (4 * 16 * 128 + 16 * 80) + 16384;
    const int grid = ((d.batch + 7) / 8) * 8 * d.parts;
#ifdef JSTSP_FUSED_DBG_BUILD
    if (GB == 4 && getenv("JSTSP_FUSED_DBG") && atoi(getenv("JSTSP_FUSED_DBG"))) {
#define DBG_CASE(k_)                                                                                                          \
    case k_:                                                                                                                  \
        JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass64_kernel<4, k_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        hipLaunchKernelGGL((fused_pass64_kernel<4, k_>), dim3(grid), dim3(512), sh, ctx->stream, d);                          \
        return 0;
        const int dbg = atoi(getenv("JSTSP_FUSED_DBG"));
        if (dbg & 16) {
            static int calls = 0;
            if (dbg == 16) {
                JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass64_kernel<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                hipLaunchKernelGGL((fused_pass64_kernel<4, 16>), dim3(grid), dim3(512), sh, ctx->stream, d);
            } else {
                JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass64_kernel<4, 26>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                hipLaunchKernelGGL((fused_pass64_kernel<4, 26>), dim3(grid), dim3(512), sh, ctx->stream, d);
            }
            if (++calls == 7) {
                unsigned long long c[12];
                JSTSP_HIP(hipStreamSynchronize(ctx->stream));
                JSTSP_HIP(hipMemcpy(c, d.Kf + 448, sizeof(c), hipMemcpyDeviceToHost));
                const double n = 8.0 * d.parts * calls * ((d.M / 32) / d.parts);     // (waves x tiles) of trial 0 so far
                fprintf(stderr, "pass64 timing dbg=%d (trial 0, cycles per wave and tile): phase A %.0f, [exchange+stage reads+barrier %.0f, update+stores %.0f, barrier %.0f, k fragments+barrier %.0f, "
                        "issue %.0f], phase B %.0f, Y %.0f, load wait %.0f, barrier %.0f; loop %.0f\n", dbg, c[3] / n, c[7] / n, c[8] / n, c[9] / n,
                        c[10] / n, c[4] / n, c[5] / n, c[6] / n, c[0] / n, c[1] / n, c[2] / n);
                calls = 0;
            }
            return 0;
        }
        switch (atoi(getenv("JSTSP_FUSED_DBG"))) {
            DBG_CASE(1) DBG_CASE(2) DBG_CASE(4) DBG_CASE(8) DBG_CASE(5) DBG_CASE(7) DBG_CASE(13) DBG_CASE(15) DBG_CASE(10)
        default: break;
        }
#undef DBG_CASE
    }
#endif
    // JSTSP_PASS_ACC: how the products of K B^H enter their running sums (see phase B of the kernel)
#ifdef JSTSP_EXPERIMENTS
    if (tune().pass_acc != 1) {
        JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass64_kernel<GB, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL((fused_pass64_kernel<GB, 0, 0>), dim3(grid), dim3(512), sh, ctx->stream, d);
        return 0;
    }
#endif
    JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass64_kernel<GB, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL((fused_pass64_kernel<GB, 0, 1>), dim3(grid), dim3(512), sh, ctx->stream, d);
    return 0;
}

template <int GB, bool YIN, bool TOEP> static int launch_fused_gb(jstsp_ctx *ctx, const FusedDesc &d)
{
    const size_t sh = (size_t)32 * (128 * GB * 8 + FPAD) + 24576;
    const int grid = ((d.batch + 7) / 8) * 8 * d.parts;
#ifdef JSTSP_FUSED_DBG_BUILD        // timing experiments (tools/pass_breakdown.py): parts of the kernel switched off, results wrong
    if (GB == 4 && YIN && getenv("JSTSP_FUSED_DBG") && atoi(getenv("JSTSP_FUSED_DBG"))) {
#define DBG_CASE(k_)                                                                                                          \
    case k_:                                                                                                                  \
        JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass_kernel<4, k_, true, TOEP>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));                                  \
        hipLaunchKernelGGL((fused_pass_kernel<4, k_, true, TOEP>), dim3(grid), dim3(512), sh, ctx->stream, d);                \
        return 0;
        switch (atoi(getenv("JSTSP_FUSED_DBG"))) {
            DBG_CASE(1) DBG_CASE(2) DBG_CASE(4) DBG_CASE(8) DBG_CASE(5) DBG_CASE(7) DBG_CASE(13) DBG_CASE(15) DBG_CASE(10)
        default: break;
        }
#undef DBG_CASE
    }
#endif
    JSTSP_HIP(hipFuncSetAttribute((const void *)fused_pass_kernel<GB, 0, YIN, TOEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL((fused_pass_kernel<GB, 0, YIN, TOEP>), dim3(grid), dim3(512), sh, ctx->stream, d);
    return 0;
}

template <bool YIN, bool TOEP> static int launch_fused_y(jstsp_ctx *ctx, const FusedDesc &d)
{
    switch (d.G2 / 128) {
    case 1: return launch_fused_gb<1, YIN, TOEP>(ctx, d);
    case 2: return launch_fused_gb<2, YIN, TOEP>(ctx, d);
    case 3: return launch_fused_gb<3, YIN, TOEP>(ctx, d);
    default: return launch_fused_gb<4, YIN, TOEP>(ctx, d);
    }
}

int launch_fused_pass(jstsp_ctx *ctx, const FusedDesc &d)
{
    JSTSP_REQUIRE(fused_shape_ok(64, d.M, d.G2, d.parts), JSTSP_E_UNSUPPORTED, "fused pass: shape");
    JSTSP_REQUIRE(d.Ec ? (d.gsh >= 4 && d.gsh <= 8) : d.Bf != nullptr, JSTSP_E_UNSUPPORTED, "fused pass: dictionary image");
    prof_begin(ctx, "fused_pass");
    int rc = 0;
    if (d.v2) {
        JSTSP_REQUIRE(d.Wqp && d.Ec && d.XsD && d.Kf && d.gsh == 6, JSTSP_E_UNSUPPORTED, "fused pass: window image");
        switch (d.G2 / 128) {
        case 1: rc = launch_fused64<1>(ctx, d); break;
        case 2: rc = launch_fused64<2>(ctx, d); break;
        case 3: rc = launch_fused64<3>(ctx, d); break;
        default: rc = launch_fused64<4>(ctx, d); break;
        }
    } else
    if (d.Wqp) rc = d.Ec ? launch_fused_y<true, true>(ctx, d) : launch_fused_y<true, false>(ctx, d);
    else       rc = d.Ec ? launch_fused_y<false, true>(ctx, d) : launch_fused_y<false, false>(ctx, d);   // Y read from memory (JSTSP_FUSED_Y=0)
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

int fused_reduce(jstsp_ctx *ctx, const FusedWS &f, int G2, int M, int batch, float2 *Tc)
{
    const long long n4 = 64ll * G2 / 2;
    if (f.v2)
        hipLaunchKernelGGL(reduce_parts_delta_kernel, dim3((unsigned)((n4 + 255) / 256), batch), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const float4 *>(f.Ppart), f.parts, n4, reinterpret_cast<float4 *>(Tc), f.Kf, f.Bdl,
                           f.sBdl, G2);
    else
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((n4 + 255) / 256), batch), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const float4 *>(f.Ppart), f.parts, n4, reinterpret_cast<float4 *>(Tc));
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
