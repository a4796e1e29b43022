// Ladder micro-benchmark: start from a pure MFMA stream shaped like the complex GEMM's wave tile
// (32 x 64 complex, 8 MFMAs per k-pair) and add, one at a time, LDS fragment reads, the per-step
// barrier, and the global->register->LDS panel pipeline.  Each rung is its own kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int BK = 16, LDA = 65, LDB = 129;

template <int FEAT>
__global__ __launch_bounds__(256) void k(const float2 *A, const float2 *B, float2 *out, int nk)
{
    __shared__ float2 sA[2][BK * LDA];
    __shared__ float2 sB[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave & 1, wj = wave >> 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int e = tid; e < BK * LDA; e += 256) { sA[0][e] = make_float2(1.f + e, 2.f); sA[1][e] = make_float2(2.f, 1.f + e); }
    for (int e = tid; e < BK * LDB; e += 256) { sB[0][e] = make_float2(1.f, 3.f + e); sB[1][e] = make_float2(3.f + e, 1.f); }
    __syncthreads();
    f32x16 r0 = {0}, i0 = {0}, r1 = {0}, i1 = {0};
    // panel pointers (i-contiguous A: 64 x 16 per step; j-contiguous B: 128 x 16 per step)
    const float2 *pa = A + (size_t)blockIdx.x * 64 * BK * nk + tid;     // 4 elements per thread, stride 256
    const float2 *pb = B + (size_t)blockIdx.x * 128 * BK * nk + tid;    // 8 elements per thread
    float2 ra[4], rb[8];
    if (FEAT & 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p) ra[p] = pa[256 * p];
#pragma unroll
        for (int p = 0; p < 8; ++p) rb[p] = pb[256 * p];
    }
    float2 av = make_float2(1.f + lane, 2.f), b0 = make_float2(3.f, 4.f + lane), b1 = make_float2(5.f, 6.f);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (FEAT & 4) {
            pa += 64 * BK; pb += 128 * BK;
            if (kt + 1 < nk) {
#pragma unroll
                for (int p = 0; p < 4; ++p) ra[p] = pa[256 * p];
#pragma unroll
                for (int p = 0; p < 8; ++p) rb[p] = pb[256 * p];
            }
        }
        const float2 *a = &sA[buf][wi * 32 + l31];
        const float2 *bb = &sB[buf][wj * 64 + l31];
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            if (FEAT & 1) {
                const int kr = 2 * kp + lhi;
                av = a[kr * LDA]; b0 = bb[kr * LDB]; b1 = bb[kr * LDB + 32];
            }
            r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.x, av.x, r0, 0, 0, 0);
            i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.x, av.y, i0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.x, av.x, r1, 0, 0, 0);
            i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.x, av.y, i1, 0, 0, 0);
            r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(-b0.y, av.y, r0, 0, 0, 0);
            i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.y, av.x, i0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_32x32x2f32(-b1.y, av.y, r1, 0, 0, 0);
            i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.y, av.x, i1, 0, 0, 0);
        }
        if (FEAT & 4) {
            if (kt + 1 < nk) {
#pragma unroll
                for (int p = 0; p < 4; ++p) { const int e = tid + 256 * p; sA[buf ^ 1][(e / 64) * LDA + e % 64] = ra[p]; }
#pragma unroll
                for (int p = 0; p < 8; ++p) { const int e = tid + 256 * p; sB[buf ^ 1][(e / 128) * LDB + e % 128] = rb[p]; }
            }
        }
        if (FEAT & 2) __syncthreads();
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += r0[r] + i0[r] + r1[r] + i1[r];
    out[blockIdx.x * 256 + tid] = make_float2(s, 0.f);
}

template <int FEAT> int run(const char *name, int blocks, int nk, const float2 *A, const float2 *B, float2 *out)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<FEAT><<<blocks, 256>>>(A, B, out, nk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<FEAT><<<blocks, 256>>>(A, B, out, nk);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double fl = (double)blocks * 4 * nk * 64 * 4096.0;
    printf("%-44s blocks %5d nk %4d  %.3f ms  %.1f TFLOP/s\n", name, blocks, nk, ms, fl / ms / 1e9);
    return 0;
}

__global__ void fill(float *p, size_t n, int zero)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        unsigned x = (unsigned)(i * 2654435761u) ^ 0x9e3779b9u;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = zero ? 0.f : ((float)(x & 0xffffff) / 8388608.f - 1.f);
    }
}

int main(int argc, char **argv)
{
    const int zero = argc > 1 ? atoi(argv[1]) : 0;
    const int nk = 256;          // k = 4096
    const int maxb = 2048;
    float2 *A, *B, *out;
    CHECK(hipMalloc(&A, (size_t)maxb * 64 * BK * nk * sizeof(float2)));
    CHECK(hipMalloc(&B, (size_t)maxb * 128 * BK * nk * sizeof(float2)));
    CHECK(hipMalloc(&out, (size_t)maxb * 256 * sizeof(float2)));
    fill<<<4096, 256>>>((float *)A, (size_t)maxb * 64 * BK * nk * 2, zero);
    fill<<<4096, 256>>>((float *)B, (size_t)maxb * 128 * BK * nk * 2, zero);
    CHECK(hipDeviceSynchronize());
    printf("operands: %s\n", zero ? "zeros" : "uniform random [-1,1)");
    for (int blocks : {1024, 2048}) {
        run<0>("MFMA only", blocks, nk, A, B, out);
        run<1>("+ LDS fragment reads", blocks, nk, A, B, out);
        run<3>("+ LDS reads + barrier/step", blocks, nk, A, B, out);
        run<7>("+ LDS reads + barrier + global->LDS panels", blocks, nk, A, B, out);
        run<6>("barrier + panels, no LDS fragment reads", blocks, nk, A, B, out);
    }
    return 0;
}
