// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds halves whose value is their index; lane l supplies the byte address 8 l.
// Prints, per lane, the four half indices it receives.  Build: hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(short4v *out)
{
    __shared__ __align__(16) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) short4v *)((__attribute__((address_space(3))) char *)lds + threadIdx.x * 8);
    out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
}
int main()
{
    short4v *d, h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l].x, h[l].y, h[l].z, h[l].w);
    return 0;
}
