// Round 6, stage 0 of the pass rewrite the round-5 review asked for: what do the TWO PRODUCT PHASES of the one-pass iteration sustain
// in a 4-wave / 512-register workgroup (one wave per SIMD, v_mfma_f32_32x32x16_f16, P = the 64 x 512 complex partial sums of K B^H
// in 256 registers per lane) when their operands really come from where the pass gets them - the 64 x 39 window of the
// block-Toeplitz dictionary in LDS (ds_read_b128 for (A S) B, ds_read_b64_tr_b16 for K B^H), the (A S) fragments from global
// memory / L2, the k fragments from LDS - and nothing else runs (no element-wise section, no Y, no prefetch of state)?
// Compare with `tools/pass_breakdown.py` dbg = 10 ("products only") of the shipped 8-wave fused_pass64_kernel: 1.125 ms per
// launch at BASELINE configs[1] (256 trials x 4 column ranges x 32 tiles).  The numbers decide whether the rewrite can pay:
// the 4-wave design only wins if its product phases are at least as fast as the 8-wave kernel's, because what it can hide is
// the 0.5 ms the 8-wave kernel spends outside them.
//
// Shapes as BASELINE configs[1]: N = 64, G2 = 512 (8 delay blocks of 64 rows), 32 columns per tile, 32 tiles per workgroup,
// 1024 workgroups.  Arithmetic is the pass's (split-f16: h h + h l + l h per real product, 12 MFMAs per complex block-step)
// on random operands; results are written out (so nothing is optimised away) but mean nothing.
//   hipcc --offload-arch=gfx950 -O3 -o pass4_products pass4_products.cpp && ./pass4_products
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int EPL = 40 * 128;          // bytes of one plane of the window: 40 columns x 8 octets x 16 B
constexpr int EBUF = 4 * EPL;          // 20 KiB: re_h, re_l, im_h, im_l
// octet swizzle of the 4-wave image: bit 2 = (cc >> 1) & 1 makes the transposing reads of a 32-lane half (4 columns x 4 octets)
// conflict-free, the bijection in (cc >> 1) & 7 the plain reads of 16 consecutive columns
__host__ __device__ inline int sw4(int cc) { return (((cc >> 1) & 1) << 2) | (((cc >> 2) & 1) << 1) | ((cc >> 3) & 1); }

__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 negu(u32x4 u) { return u ^ 0x80008000u; }

// MODE 0: both phases; 1: phase A only; 2: phase B only; 3: phase A without its global loads ((A S) fragments loaded once);
// 4: phase A without its LDS reads (window fragments read once); 5: both phases, (A S) fragments loaded once
template <int MODE>
__global__ __launch_bounds__(256) void pass4_products(const u32x4 *__restrict__ E, const u32x4 *__restrict__ AS, float *__restrict__ out, int tiles)
{
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char *xch = lds + 2 * EBUF;               // k fragments: [n-half 2][k-step 2][plane 6][lane 64] 16 B = 24 KiB
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int nb2 = w & 1, h = w >> 1;
    const int m32 = l & 31, kg = l >> 5;
    const int wg = blockIdx.x;

    f32x16 pr[8], pi[8];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { pr[b][r] = 0.f; pi[b][r] = 0.f; }

    // k fragments: written once (random-ish halves), read every tile
    for (int i = tid; i < 24576 / 16; i += 256) {
        const unsigned v = 0x2c003400u + 0x00010001u * (unsigned)((i * 37 + wg) & 0x3ff);
        reinterpret_cast<u32x4 *>(xch)[i] = u32x4{v, v ^ 0x8000u, v + 0x11u, v ^ 0x80000000u};
    }
    // window of tile 0
    const u32x4 *Et = E + (size_t)(wg & 63) * (4 * 4104 * 8);      // 64 distinct images of 4 planes x 4104 columns x 8 octets
    auto load_window = [&](int T, u32x4 *regs) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int piece = tid + 256 * j;                          // 1280 chunks of 16 B: plane = piece / 320
            const int p = piece / 320, c = piece - 320 * p;
            regs[j] = __builtin_nontemporal_load(Et + (size_t)p * (4104 * 8) + (size_t)T * 256 + c);
        }
    };
    auto store_window = [&](int buf, const u32x4 *regs) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int piece = tid + 256 * j;
            *reinterpret_cast<u32x4 *>(lds + buf * EBUF + piece * 16) = regs[j];
        }
    };
    u32x4 wreg[5];
    load_window(0, wreg);
    store_window(0, wreg);
    __syncthreads();

    // the pass's workgroup -> (trial, column range) map: the four column ranges of a trial run on ONE XCD, so that its (A S) fragments
    // (256 KiB) stay in that XCD's L2 - 8 trials per XCD at a time
    const int trial = ((wg >> 3) / 4) * 8 + (wg & 7);
    const u32x4 *ast = AS + (size_t)(trial & 63) * (32 * 2 * 4 * 64);      // [ks 32][nb2 2][plane 4][lane 64]
    auto *lbase = (__attribute__((address_space(3))) unsigned char *)lds;

    for (int i = 0; i < tiles; ++i) {
        const unsigned char *ebuf = lds + (i & 1) * EBUF;
        if (i + 1 < tiles) load_window(i + 1, wreg);
        // ================= phase A: Xs^T (32 m x 32 n of n-half nb2) over the g-half h: 16 k-steps of 16 rows g
        f32x16 ar, ai;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ar[r] = 0.f; ai[r] = 0.f; }
        if (MODE != 2) {
            constexpr bool NOG = MODE == 3 || MODE == 5, NOL = MODE == 4;
            constexpr int WD = 3;               // (A S) fragments requested WD - 1 k-steps ahead: one wave per SIMD has nobody to hide an L2 round trip
            u32x4 wf[WD][4];
            // ONE running byte offset, advanced opaquely per k-step (constant offsets make hipcc materialise - and spill - an address
            // per (k-step, plane) outside the tile loop: fused.hip)
            uint32_t ao = 16u * (uint32_t)(((16 * h) * 2 + nb2) * 256 + l);
            asm volatile("" : "+v"(ao));
#pragma unroll
            for (int q = 0; q < WD - 1; ++q) {
#pragma unroll
                for (int p = 0; p < 4; ++p) wf[q][p] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(ast) + ao + p * 1024);
                ao += 8192;
                asm volatile("" : "+v"(ao));
            }
            int cK = m32 + 7 - 4 * h;
            u32x4 bfb[2][4];
            auto rdA = [&](int ks, u32x4 *dst) {
                // rows g = 256 h + 16 ks + 8 kg ..: delay block ld = 4 h + (ks >> 2), octet 2 (ks & 3) + kg of block 0, column m - ld
                if ((ks & 3) == 0) { if (ks > 0) cK -= 1; asm volatile("" : "+v"(cK)); }
                const int oc = (2 * (ks & 3) + kg) ^ sw4(cK);
                const unsigned char *arow = ebuf + cK * 128 + 16 * oc;
#pragma unroll
                for (int p = 0; p < 4; ++p) dst[p] = *reinterpret_cast<const u32x4 *>(arow + p * EPL);
            };
            rdA(0, bfb[0]);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks + WD - 1 < 16 && !(NOG && i > 0)) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) wf[(ks + WD - 1) % WD][p] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(ast) + ao + p * 1024);
                    ao += 8192;
                    asm volatile("" : "+v"(ao));
                }
                if (ks + 1 < 16 && !(NOL && i > 0)) rdA(ks + 1, bfb[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 *wk = wf[ks % WD], *bf = bfb[ks & 1];
                const u32x4 nwi_h = negu(wk[2]), nwi_l = negu(wk[3]);
                ar = mma(bf[0], wk[0], ar); ai = mma(bf[0], wk[2], ai);
                ar = mma(bf[0], wk[1], ar); ai = mma(bf[0], wk[3], ai);
                ar = mma(bf[1], wk[0], ar); ai = mma(bf[1], wk[2], ai);
                ar = mma(bf[2], nwi_h, ar); ai = mma(bf[2], wk[0], ai);
                ar = mma(bf[2], nwi_l, ar); ai = mma(bf[2], wk[1], ai);
                ar = mma(bf[3], nwi_h, ar); ai = mma(bf[3], wk[0], ai);
            }
        }
        // (stand-in for the exchange and the element-wise section: the sums feed the P accumulators so that they stay live)
        pr[0] += ar; pi[0] += ai;
        __syncthreads();
        // ================= phase B: P^T (32 g x 32 n blocks: this wave n-half nb2, blocks 8 h .. 8 h + 7) += conj(B)(g, tile) k^T
        if (MODE != 1 && MODE != 3 && MODE != 4) {
            u32x4 kf[2][6];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned char *kp = xch + (((nb2 * 2 + s) * 6) * 64 + l) * 16;
#pragma unroll
                for (int p = 0; p < 6; ++p) kf[s][p] = *reinterpret_cast<const u32x4 *>(kp + p * 1024);
            }
            const int G = l >> 4, i16 = l & 15;
            u32x4 bfb[2][4];
            auto rdB = [&](int st, u32x4 *dst) {
                const int gb = st >> 1, s = st & 1;
                const int blk = 8 * h + gb, ld = blk >> 1;                 // 32 rows g: delay ld, rows 32 (blk & 1) .. of block 0
                // k-index 8 kg + j of k-step s <-> column m = 16 s + 4 kg + (j & 3) + 8 (j >> 2): two transposing reads; lane l, 16-lane
                // group G: rows g = 32 (blk & 1) + 16 (G & 1) + (l & 15); supplies the address of the four halves g = .. + 4 (l & 3) of
                // row m = 16 s + 4 kg + ((l & 15) >> 2)
                const int gl = 32 * (blk & 1) + 16 * (G & 1) + 4 * (i16 & 3);
                int cB = 4 * kg + (i16 >> 2) + 7 + 16 * s - ld;           // (opaque: no address per (block, step) kept across the tile loop)
                asm volatile("" : "+v"(cB));
                unsigned o2[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int cc = cB + 8 * u;
                    o2[u] = (unsigned)((i & 1) * EBUF + cc * 128 + 16 * ((gl >> 3) ^ sw4(cc)) + 2 * (gl & 7));
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lbase + o2[0] + p * EPL)));
                    const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lbase + o2[1] + p * EPL)));
                    dst[p] = u32x4{p0.x, p0.y, p1.x, p1.y};
                }
            };
            rdB(0, bfb[0]);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int gb = st >> 1, s = st & 1;
                if (st + 1 < 16) rdB(st + 1, bfb[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 *bf = bfb[st & 1], *k = kf[s];
                // re += Br kr + Bi ki ; im += Br ki - Bi kr
                pr[gb] = mma(bf[0], k[0], pr[gb]); pi[gb] = mma(bf[0], k[2], pi[gb]);
                pr[gb] = mma(bf[0], k[1], pr[gb]); pi[gb] = mma(bf[0], k[3], pi[gb]);
                pr[gb] = mma(bf[1], k[0], pr[gb]); pi[gb] = mma(bf[1], k[2], pi[gb]);
                pr[gb] = mma(bf[2], k[2], pr[gb]); pi[gb] = mma(bf[2], k[4], pi[gb]);
                pr[gb] = mma(bf[2], k[3], pr[gb]); pi[gb] = mma(bf[2], k[5], pi[gb]);
                pr[gb] = mma(bf[3], k[2], pr[gb]); pi[gb] = mma(bf[3], k[4], pi[gb]);
            }
        }
        if (i + 1 < tiles) store_window((i + 1) & 1, wreg);
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += pr[b][r] + pi[b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

int main()
{
    const int wgs = 1024, tiles = 32;
    const size_t eN = (size_t)64 * 4 * 4104 * 8, aN = (size_t)64 * 32 * 2 * 4 * 64;
    std::vector<uint32_t> he(eN * 4), ha(aN * 4);
    srand(1);
    auto rh = []() { // a random half in [-2, 2): sign, exponent 8..15, random mantissa
        const unsigned s = rand() & 1, e = 8 + (rand() & 7), m = rand() & 0x3ff;
        return (s << 15) | (e << 10) | m;
    };
    for (auto &x : he) x = rh() | (rh() << 16);
    for (auto &x : ha) x = rh() | (rh() << 16);
    u32x4 *dE, *dA; float *dO;
    CHECK(hipMalloc(&dE, eN * 16)); CHECK(hipMalloc(&dA, aN * 16)); CHECK(hipMalloc(&dO, (size_t)wgs * 256 * 4));
    CHECK(hipMemcpy(dE, he.data(), eN * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dA, ha.data(), aN * 16, hipMemcpyHostToDevice));
    const size_t sh = 2 * EBUF + 24576;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char *name, double mfma_per_wave_tile) {
        CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void *)kern));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), sh, 0, dE, dA, dO, tiles);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), sh, 0, dE, dA, dO, tiles);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        const double flop = mfma_per_wave_tile * 4 * wgs * tiles * 32768.0;
        printf("%-34s %.3f ms per launch  %7.1f TFLOP/s executed  (registers %d, spilled %d bytes, LDS %zu)\n", name, ms, flop / ms / 1e9,
               fa.numRegs, (int)fa.localSizeBytes, sh);
    };
    run(pass4_products<0>, "4-wave products, both phases", 384);
    run(pass4_products<1>, "4-wave products, (A S) B only", 192);
    run(pass4_products<2>, "4-wave products, K B^H only", 192);
    run(pass4_products<3>, "(A S) B only, no global loads", 192);
    run(pass4_products<4>, "(A S) B only, no LDS reads", 192);
    run(pass4_products<5>, "both phases, no global loads", 384);
    return 0;
}
