// HBM streaming micro-benchmark: what does a plain element-wise kernel reach on this part, for the read/write mixes
// of the solver's fused updates?   hipcc --offload-arch=gfx950 -O3 stream.cpp -o stream && ./stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NR, int NW, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const f4 *const *rd, f4 *const *wr, long long n)
{
    const long long stride = (long long)gridDim.x * 256 * UNROLL;
    for (long long i = ((long long)blockIdx.x * 256) * UNROLL + threadIdx.x; i < n; i += stride) {
        f4 acc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc[u] = (f4)(0.f);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const long long j = i + 256 * u;
                if (j < n) acc[u] += NT ? __builtin_nontemporal_load(&rd[r][j]) : rd[r][j];
            }
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const long long j = i + 256 * u;
                if (j < n) {
                    if (NT) __builtin_nontemporal_store(acc[u] * (float)(w + 1), &wr[w][j]);
                    else wr[w][j] = acc[u] * (float)(w + 1);
                }
            }
    }
}

template <int NR, int NW, bool NT, int UNROLL>
static int run(const char *name, f4 **bufs, const f4 *const *drd, f4 *const *dwr, long long n, int grid)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) stream_kernel<NR, NW, NT, UNROLL><<<grid, 256>>>(drd, dwr, n);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) stream_kernel<NR, NW, NT, UNROLL><<<grid, 256>>>(drd, dwr, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)(NR + NW) * n * 16.0 * reps;
    printf("%-28s %dR+%dW %s unroll %d grid %6d : %7.1f GB/s\n", name, NR, NW, NT ? "nt " : "   ", UNROLL, grid,
           bytes / (ms * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const long long n = (long long)256 * 64 * 4096 / 2;       // one N x M x batch complex array = 0.54 GB, in float4
    f4 *bufs[12];
    for (int i = 0; i < 12; ++i) { CK(hipMalloc(&bufs[i], n * 16)); CK(hipMemset(bufs[i], 0, n * 16)); }
    const f4 **drd; f4 **dwr;
    CK(hipMalloc(&drd, 8 * sizeof(f4 *))); CK(hipMalloc(&dwr, 8 * sizeof(f4 *)));
    CK(hipMemcpy(drd, bufs, 6 * sizeof(f4 *), hipMemcpyHostToDevice));
    CK(hipMemcpy(dwr, bufs + 6, 6 * sizeof(f4 *), hipMemcpyHostToDevice));
    for (int grid : {2048, 8192, 32768}) {
        run<1, 1, false, 1>("copy", bufs, drd, dwr, n, grid);
        run<1, 1, false, 4>("copy", bufs, drd, dwr, n, grid);
        run<1, 1, true, 4>("copy", bufs, drd, dwr, n, grid);
        run<2, 1, false, 4>("form_z-like", bufs, drd, dwr, n, grid);
        run<2, 1, true, 4>("form_z-like", bufs, drd, dwr, n, grid);
        run<5, 4, false, 2>("update_x-like", bufs, drd, dwr, n, grid);
        run<5, 4, true, 2>("update_x-like", bufs, drd, dwr, n, grid);
        run<1, 0, false, 4>("read only", bufs, drd, dwr, n, grid);
        run<6, 0, false, 2>("read only", bufs, drd, dwr, n, grid);
    }
    return 0;
}
