// Minimal form of what tools/probe/lanczos_race.cpp found: does a chain of DEPENDENT packed-fp32 operations (v_pk_fma_f32 /
// v_pk_add_f32, as hipcc emits them for complex arithmetic on gfx950) give the same bits when waves of an MFMA kernel share
// the SIMD?  Victim: every thread iterates z <- z * c + d (complex, fp32) `steps` times on fixed data; aggressor on a second
// stream: a register-only MFMA loop with a barrier + LDS round trip every 16 products.  The victim is built twice in this file:
// with packed-fp32 ops (vector types, the default code generation) and without (scalar arithmetic behind an opaque asm fence).
//   hipcc --offload-arch=gfx950 -O3 -o pk_fp32_probe.bin pk_fp32_probe.hip && ./pk_fp32_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void aggressor(float *out, int iters)
{
    __shared__ float sm[256];
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f16v acc0 = {0}, acc1 = {0};
    for (int k = 0; k < iters; ++k) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        if ((k & 15) == 0) { sm[threadIdx.x] = acc0[0]; __syncthreads(); acc1[1] += sm[(threadIdx.x + 64) & 255]; __syncthreads(); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[3];
}

// z <- z * c + d with packed operations: (zr, zi) * cr + (-zi, zr) * ci + d
template <int VAR>      // 0: scalar arithmetic; 1: the full mix; 2..6: one packed instruction form at a time
__global__ __launch_bounds__(256) void victim(const f2 *cin, const f2 *din, f2 *out, int steps)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const f2 c = cin[i], d = din[i];
    f2 z = {0.25f, -0.5f};
    if (VAR > 0) {
        // the instruction forms of the Lanczos kernel's inner product: op_sel, s_nop 0 between dependent packed ops (what
        // hipcc emits), a v_pk_mov_b32 swap of the halves
        for (int k = 0; k < steps; ++k) {
            f2 acc = {0.f, 0.f}, w = d;                     // every step starts from z only: all values stay O(1)
            if (VAR == 1)
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\t"
                             "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
                             "s_nop 0\n\t"
                             "v_pk_add_f32 %1, %1, %0\n\t"
                             "s_nop 0\n\t"
                             "v_pk_mov_b32 %1, %1, %1 op_sel:[1,0]\n\t"
                             "s_nop 0\n\t"
                             "v_pk_mul_f32 %0, %1, %4\n\t"
                             "s_nop 0\n\t"
                             "v_pk_fma_f32 %1, %0, %3, %1 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                             : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d));
            if (VAR == 2)       // plain v_pk_fma_f32 chain
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\ts_nop 0\n\tv_pk_fma_f32 %1, %0, %3, %1\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %0, %1, %4, %0\n\t" : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d));
            if (VAR == 3)       // op_sel forms
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %1, %0, %3, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %0, %1, %4, %0 op_sel_hi:[1,0,1]\n\t" : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d));
            if (VAR == 4)       // v_pk_add_f32 / v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %2, %3\n\ts_nop 0\n\tv_pk_add_f32 %1, %1, %0\n\ts_nop 0\n\t"
                             "v_pk_mul_f32 %0, %1, %4\n\ts_nop 0\n\tv_pk_add_f32 %1, %1, %0\n\t"
                             : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d));
            if (VAR == 5)       // v_pk_mov_b32 swap between packed operations
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\ts_nop 0\n\tv_pk_mov_b32 %0, %0, %0 op_sel:[1,0]\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %1, %0, %4, %1\n\ts_nop 0\n\tv_pk_mov_b32 %1, %1, %1 op_sel:[1,0]\n\t"
                             : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d));
#define OPSEL_CHAIN(NOP)                                                                                               \
    asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t" NOP "\n\t"                             \
                 "v_pk_fma_f32 %1, %0, %3, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t" NOP "\n\t"                             \
                 "v_pk_fma_f32 %0, %1, %4, %0 op_sel_hi:[1,0,1]\n\t" NOP "\n\t" : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d))
            if (VAR == 6) OPSEL_CHAIN("s_nop 1");      // as 3 with 2 / 4 / 8 / 16 wait states between dependent operations
            if (VAR == 7) OPSEL_CHAIN("s_nop 3");
            if (VAR == 8) OPSEL_CHAIN("s_nop 7");
            if (VAR == 9) OPSEL_CHAIN("s_nop 15");
            if (VAR == 10)      // as 3, the dependent operand is a SOURCE WITHOUT op_sel (src0), op_sel only on the constant
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %1, %0, %3, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %0, %1, %4, %0\n\t" : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d));
            if (VAR == 11)      // op_sel applied to the operand that was just written (src1 = previous result)
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %1, %3, %0, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                             "v_pk_fma_f32 %0, %4, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t" : "+v"(acc), "+v"(w) : "v"(z), "v"(c), "v"(d));
            z = (acc + w) * 1e-2f + z * 0.5f;
        }
    } else {
        float zr = z.x, zi = z.y;
        for (int k = 0; k < steps; ++k) {
            float tr = fmaf(zr, c.x, d.x), ti = fmaf(zi, c.x, d.y);
            asm volatile("" : "+v"(tr), "+v"(ti));          // keeps the two halves apart: no v_pk_* here
            const float nr = fmaf(-zi, c.y, tr), ni = fmaf(zr, c.y, ti);
            zr = nr; zi = ni;
            asm volatile("" : "+v"(zr), "+v"(zi));
        }
        z.x = zr; z.y = zi;
    }
    out[i] = z;
}

int main()
{
    const int n = 256 * 256, steps = 4000, reps = 100;
    std::vector<f2> hc(n), hd(n);
    for (int i = 0; i < n; ++i) {
        hc[i] = f2{0.9f * cosf(0.001f * i), 0.9f * sinf(0.001f * i)};      // |c| < 1: the iteration converges, no overflow
        hd[i] = f2{0.3f + 1e-5f * i, -0.2f};
    }
    f2 *c, *d, *o; float *ao;
    hipMalloc(&c, n * 8); hipMalloc(&d, n * 8); hipMalloc(&o, (size_t)reps * n * 8); hipMalloc(&ao, 1024 * 256 * 4);
    hipMemcpy(c, hc.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), n * 8, hipMemcpyHostToDevice);
    hipStream_t sA, sB;
    hipStreamCreateWithFlags(&sA, hipStreamNonBlocking); hipStreamCreateWithFlags(&sB, hipStreamNonBlocking);
    std::vector<f2> h((size_t)reps * n);
    const char *names[12] = {"scalar fp32", "full packed mix", "v_pk_fma_f32", "v_pk_fma_f32 op_sel", "v_pk_mul/add_f32",
                             "v_pk_fma + v_pk_mov_b32", "op_sel, s_nop 1", "op_sel, s_nop 3", "op_sel, s_nop 7", "op_sel, s_nop 15",
                             "op_sel on constants only", "op_sel on fresh result"};
    for (int var = 0; var < 12; ++var)
        for (int beside = 0; beside <= 1; ++beside) {
            for (int r = 0; r < reps; ++r) {
                if (beside) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sB, ao, 3000);
                f2 *dst = o + (size_t)r * n;
                switch (var) {
                case 0: hipLaunchKernelGGL(victim<0>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 1: hipLaunchKernelGGL(victim<1>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 2: hipLaunchKernelGGL(victim<2>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 3: hipLaunchKernelGGL(victim<3>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 4: hipLaunchKernelGGL(victim<4>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 5: hipLaunchKernelGGL(victim<5>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 6: hipLaunchKernelGGL(victim<6>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 7: hipLaunchKernelGGL(victim<7>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 8: hipLaunchKernelGGL(victim<8>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 9: hipLaunchKernelGGL(victim<9>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                case 10: hipLaunchKernelGGL(victim<10>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                default: hipLaunchKernelGGL(victim<11>, dim3(n / 256), dim3(256), 0, sA, c, d, dst, steps); break;
                }
            }
            hipDeviceSynchronize();
            hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
            long bad = 0;
            for (int r = 1; r < reps; ++r)
                for (int i = 0; i < n; ++i) bad += memcmp(&h[(size_t)r * n + i], &h[i], 8) != 0;
            printf("%-26s %-22s: %ld of %ld results differ from the first repetition\n", names[var],
                   beside ? "beside the MFMA kernel" : "alone", bad, (long)(reps - 1) * n);
        }
    return 0;
}
