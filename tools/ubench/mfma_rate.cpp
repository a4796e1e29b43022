// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 streams shaped like the complex GEMM's.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
    f32x16 r0 = {0}, i0 = {0}, r1 = {0}, i1 = {0};
    float ax = a0 + threadIdx.x, ay = a0 * 2 + threadIdx.x, b0x = b0, b0y = b0 + 1, b1x = b0 + 2, b1y = b0 + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (VAR == 0) {          // 4 accumulators, plain
                r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0x, ax, r0, 0, 0, 0);
                i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0x, ay, i0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1x, ax, r1, 0, 0, 0);
                i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1x, ay, i1, 0, 0, 0);
                r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0y, ay, r0, 0, 0, 0);
                i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0y, ax, i0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1y, ay, r1, 0, 0, 0);
                i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1y, ax, i1, 0, 0, 0);
            } else if (VAR == 1) {   // with the negations as VALU between
                r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0x, ax, r0, 0, 0, 0);
                i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0x, ay, i0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1x, ax, r1, 0, 0, 0);
                i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1x, ay, i1, 0, 0, 0);
                r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(-b0y, ay, r0, 0, 0, 0);
                i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0y, ax, i0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_32x32x2f32(-b1y, ay, r1, 0, 0, 0);
                i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1y, ax, i1, 0, 0, 0);
                b0y += 1.f; b1y += 1.f;
            } else if (VAR == 2) {   // 16x16x4, 8 accumulators of 4 regs
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 *p = reinterpret_cast<f32x4 *>(&r0);
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0x, ax, p[q], 0, 0, 0);
                f32x4 *p2 = reinterpret_cast<f32x4 *>(&i0);
#pragma unroll
                for (int q = 0; q < 4; ++q) p2[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0y, ay, p2[q], 0, 0, 0);
                f32x4 *p3 = reinterpret_cast<f32x4 *>(&r1);
#pragma unroll
                for (int q = 0; q < 4; ++q) p3[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1x, ax, p3[q], 0, 0, 0);
                f32x4 *p4 = reinterpret_cast<f32x4 *>(&i1);
#pragma unroll
                for (int q = 0; q < 4; ++q) p4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1y, ay, p4[q], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += r0[r] + i0[r] + r1[r] + i1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR> void run(const char *name, int blocks, float flop_per_mfma, int mfma_per_iter)
{
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<VAR><<<blocks, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<VAR><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double fl = (double)blocks * 4 * iters * mfma_per_iter * flop_per_mfma;
    printf("%-28s blocks %4d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
    hipFree(out);
}

int main()
{
    for (int blocks : {256, 512, 1024}) {
        run<0>("32x32x2 4acc plain", blocks, 4096.f, 64);
        run<1>("32x32x2 4acc + VALU neg", blocks, 4096.f, 64);
        run<2>("16x16x4 16acc", blocks, 2048.f, 128);
    }
    return 0;
}
