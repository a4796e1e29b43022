// Does global_load_lds_dwordx4 (LDS-direct load, gfx950) reach LDS addresses above 64 KiB through M0, and in which lane order?
// build + run:  hipcc --offload-arch=gfx950 -O2 tools/probe/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(const float4 *src, float4 *dst, unsigned base)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + base + 1024u * w;
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(src), "s"(ldsaddr) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dst[threadIdx.x] = *reinterpret_cast<const float4 *>(lds + base + threadIdx.x * 16);
}
int main()
{
    std::vector<float4> h(512), o(512);
    for (int i = 0; i < 512; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *s, *d;
    hipMalloc(&s, 8192); hipMalloc(&d, 8192);
    hipMemcpy(s, h.data(), 8192, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad_total = 0;
    for (unsigned base : {0u, 40960u, 65536u, 100000u - 100000u % 16, 131072u, 155648u}) {
        hipMemset(d, 0xff, 8192);
        k<<<1, 512, 160 * 1024>>>(s, d, base);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 512; ++i) bad += !(o[i].x == h[i].x && o[i].y == h[i].y && o[i].z == h[i].z && o[i].w == h[i].w);
        printf("LDS base %6u: %s, %d of 512 lanes wrong (first values %.2f %.2f)\n", base, hipGetErrorString(e), bad, o[0].x, o[1].x);
        bad_total += bad;
    }
    return bad_total != 0;
}
