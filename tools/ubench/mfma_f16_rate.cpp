// Micro-benchmark: what the f16 matrix pipe sustains on MI355X with nothing but MFMAs in flight - v_mfma_f32_32x32x16_f16 and
// v_mfma_f32_16x16x32_f16 streams over 4 / 8 independent accumulators, operands in registers, 1 / 2 waves per SIMD on every
// CU - with ZERO operands and with RANDOM f16 operands (the dense peak of 2.5 PFLOP/s is an issue-rate figure; what the
// pipe sustains depends on the data, as found for the fp32 pipe in round 1: gemm_ladder.cpp).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f16_rate mfma_f16_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// VAR 0: 32x32x16, 4 accumulators (the wave tile of hgemm2_kernel); VAR 1: 16x16x32, 8 accumulators (the pass)
template <int VAR>
__global__ __launch_bounds__(256) void k(const half8 *ops, float *out, int iters, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    // shader-clock cycles (s_memtime) against the constant 100-MHz counter (s_memrealtime): the clock the chip HOLDS under this load
    const unsigned long long c0_ = clock64(), w0_ = wall_clock64();
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 256 + tid) % 2048]; b[i] = ops[((i + 4) * 256 + tid) % 2048]; }
    float s = 0;
    if (VAR == 0) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[u], a[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[u], a[1], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[u], a[2], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[u], a[3], c3, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    } else {
        f32x4 c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u], a[q & 3], c[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
    }
    out[blockIdx.x * 256 + tid] = s;
    if (clk && blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0_; clk[1] = wall_clock64() - w0_; }
}

// 16x16x32 with 16 accumulators (is the one-wave-per-SIMD rate of VAR 1 a dependency limit?)
__global__ __launch_bounds__(256) void k16(const half8 *ops, float *out, int iters)
{
    const int tid = threadIdx.x;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 256 + tid) % 2048]; b[i] = ops[((i + 4) * 256 + tid) % 2048]; }
    f32x4 c[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) c[q] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[(u + q) & 3], a[q & 3], c[q], 0, 0, 0);
    }
    float s = 0;
    for (int q = 0; q < 16; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int VAR> void run(const char *name, const half8 *ops, int blocks, double flop_per_iter)
{
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long *clk, hclk[2];
    hipMalloc(&clk, 16);
    k<VAR><<<blocks, 256>>>(ops, out, 100, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<VAR><<<blocks, 256>>>(ops, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)hclk[0] / ((double)hclk[1] / 100e6) / 1e9;           // (s_memrealtime: 100 MHz)
    const double tf = (double)blocks * 4 * iters * flop_per_iter / ms / 1e9;
    // at the held clock the issue-rate peak is 2500 x clock / 2.4 TFLOP/s: how busy the pipe is at THAT clock
    printf("  %-34s %d waves/SIMD  %8.3f ms  %7.1f TFLOP/s  clock %.2f GHz  (%.0f %% of the issue rate at that clock)\n", name, blocks / 256, ms, tf, ghz,
           100.0 * tf / (2500.0 * ghz / 2.4));
    hipFree(out); hipFree(clk);
}

int main()
{
    std::vector<_Float16> h(2048 * 8);
    half8 *ops;
    hipMalloc(&ops, h.size() * sizeof(_Float16));
    for (int mode = 0; mode < 3; ++mode) {
        srand(7);
        for (auto &x : h) {
            const float u = (float)rand() / RAND_MAX - 0.5f;
            x = (_Float16)(mode == 0 ? 0.f : (mode == 1 ? u : (u > 0 ? 1.f : -1.f) * 0.25f));
        }
        hipMemcpy(ops, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        printf("%s operands\n", mode == 0 ? "zero" : (mode == 1 ? "uniform random (-0.5, 0.5)" : "random sign, one magnitude"));
        for (int blocks : {256, 512}) {
            run<0>("v_mfma_f32_32x32x16_f16, 4 acc", ops, blocks, 16.0 * 2 * 32 * 32 * 16);
            run<1>("v_mfma_f32_16x16x32_f16, 8 acc", ops, blocks, 32.0 * 2 * 16 * 16 * 32);
            {
                float *out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                k16<<<blocks, 256>>>(ops, out, 100); hipDeviceSynchronize();
                hipEventRecord(e0); k16<<<blocks, 256>>>(ops, out, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("  %-34s %d waves/SIMD  %8.3f ms  %7.1f TFLOP/s\n", "v_mfma_f32_16x16x32_f16, 16 acc", blocks / 256, ms,
                       (double)blocks * 4 * 20000 * 32.0 * 2 * 16 * 16 * 32 / ms / 1e9);
                hipFree(out);
            }
        }
    }
    return 0;
}
