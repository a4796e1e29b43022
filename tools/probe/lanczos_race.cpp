// Is lambda_max (Lanczos kernel) bit-reproducible when other kernels run beside it?  One stream repeats launch_lmax on the SAME
// Gram partials; a second stream runs (0) nothing, (1) the three-Gram pass, (2) a dummy LDS-heavy kernel, (3) the single-Gram
// kernel, (4) a register-only MFMA loop, (5) the MFMA loop with a barrier + LDS access every 16 products.  argv[2] = 0: the
// Householder + Sturm kernel instead.  Result (round 2): with packed-fp32 VALU instructions (v_pk_fma_f32 ...) in the Lanczos
// kernel, modes 1 / 4 / 5 corrupt 5 / 4 / 3041 of 3184 results (a single Lanczos step is enough); compiled without them
// (-target-feature -packed-fp32-ops, what jstsp19_amd/build.py does) every mode gives 0.  Build (in jstsp19_amd/csrc):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I../../include -I. ../../tools/probe/lanczos_race.cpp -L. -ljstsp_mi355x -Wl,-rpath,'$ORIGIN/../../jstsp19_amd/csrc' -o ../../tools/probe/lanczos_race.bin
#include "solver_common.h"
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>
using namespace jstsp;

__global__ void dummy_lds(float *out, int iters)
{
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 12288; i += blockDim.x) sm[i] = (float)i;
    __syncthreads();
    float a = 0.f;
    for (int k = 0; k < iters; ++k) {
        a += sm[(threadIdx.x * 17 + k * 31) % 12288];
        sm[(threadIdx.x * 13 + k * 7) % 12288] = a;
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

// a register-only MFMA loop: matrix-pipe load without memory traffic (lds_touch: plus a barrier and an LDS round trip)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void dummy_mfma(float *out, int iters, int lds_touch)
{
    extern __shared__ float sm[];
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f16v acc0 = {0}, acc1 = {0};
    for (int k = 0; k < iters; ++k) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        if (lds_touch && (k & 15) == 0) { sm[threadIdx.x] = acc0[0]; __syncthreads(); acc1[1] += sm[(threadIdx.x + 64) & 255]; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[3];
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 1, reps = 200;
    const bool lanczos = argc > 2 ? atoi(argv[2]) != 0 : true;
    const int n = 64, batch = 16, nsplit = 4, M = 4096;
    jstsp_ctx *ctx = nullptr;
    if (jstsp_create(0, &ctx)) { printf("create failed\n"); return 1; }
    hipStream_t sA = ctx->stream, sB;
    hipStreamCreateWithFlags(&sB, hipStreamNonBlocking);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    // state-like matrices X, V1 (64 x 4096) per problem: low rank + noise; V2-like matrix for the Gram whose lambda_max is taken
    const size_t nm = (size_t)n * M;
    std::vector<float2> hX(batch * nm), hV(batch * nm), hW(batch * nm);
    for (auto *h : {&hX, &hV, &hW})
        for (int t = 0; t < batch; ++t) {
            std::vector<float2> u(n * 3), w(M * 3);
            for (auto &e : u) e = make_float2(nd(rng), nd(rng));
            for (auto &e : w) e = make_float2(nd(rng), nd(rng));
            for (size_t m = 0; m < (size_t)M; ++m)
                for (int i = 0; i < n; ++i) {
                    float2 s = make_float2(1e-3f * nd(rng), 1e-3f * nd(rng));
                    for (int r = 0; r < 3; ++r) {
                        const float2 a = u[i * 3 + r], b = w[m * 3 + r];
                        s.x += a.x * b.x + a.y * b.y; s.y += a.y * b.x - a.x * b.y;
                    }
                    (*h)[t * nm + i + n * m] = s;
                }
        }
    float2 *X, *V, *W, *Gz, *Gx, *Gv, *Gw; uint32_t *amax; TrialParams *prm; float *lam, *dout;
    hipMalloc(&X, batch * nm * 8); hipMalloc(&V, batch * nm * 8); hipMalloc(&W, batch * nm * 8);
    const size_t gsz = (size_t)batch * nsplit * n * n;
    hipMalloc(&Gz, gsz * 8); hipMalloc(&Gx, gsz * 8); hipMalloc(&Gv, gsz * 8); hipMalloc(&Gw, gsz * 8);
    hipMalloc(&amax, 4 * batch * 4); hipMalloc(&prm, batch * sizeof(TrialParams)); hipMalloc(&lam, (size_t)reps * batch * 4);
    hipMalloc(&dout, 256 * 1024 * 4);
    hipMemcpy(X, hX.data(), batch * nm * 8, hipMemcpyHostToDevice); hipMemcpy(V, hV.data(), batch * nm * 8, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), batch * nm * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> hm(4 * batch); float big = 64.f; for (auto &e : hm) memcpy(&e, &big, 4);
    hipMemcpy(amax, hm.data(), 4 * batch * 4, hipMemcpyHostToDevice);
    std::vector<TrialParams> hp(batch); for (auto &p : hp) { p.rho = 0.5f; p.irho = 2.f; p.tauY_rho = 1e-3f; p.tauS_rho = 1e-3f; p.c_coef = 1.f / 3; }
    hipMemcpy(prm, hp.data(), batch * sizeof(TrialParams), hipMemcpyHostToDevice);
    // Gram partials of W (the matrix whose lambda_max is repeated)
    if (launch_hgram(ctx, W, (long long)nm, n, M, batch, nsplit, amax + 3 * batch, Gw)) { printf("hgram failed: %s\n", jstsp_last_error()); return 1; }
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) {
        ctx->stream = sB;
        if (mode == 1) launch_hgram3(ctx, X, V, (long long)nm, n, M, batch, nsplit, amax, amax + batch, amax + 2 * batch, prm, Gz, Gx, Gv);
        if (mode == 2) hipLaunchKernelGGL(dummy_lds, dim3(512), dim3(512), 49152, sB, dout, 200);
        if (mode == 3) launch_hgram(ctx, X, (long long)nm, n, M, batch, nsplit, amax, Gx);
        if (mode == 4) hipLaunchKernelGGL(dummy_mfma, dim3(1024), dim3(256), 16384, sB, dout, 3000, 0);
        if (mode == 5) hipLaunchKernelGGL(dummy_mfma, dim3(1024), dim3(256), 16384, sB, dout, 3000, 1);
        ctx->stream = sA;
        launch_lmax(ctx, n, batch, Gw, (long long)n * n * nsplit, nsplit, (long long)n * n, lam + (size_t)r * batch, lanczos);
    }
    hipDeviceSynchronize();
    std::vector<float> hl((size_t)reps * batch);
    hipMemcpy(hl.data(), lam, hl.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 1; r < reps; ++r)
        for (int t = 0; t < batch; ++t)
            if (memcmp(&hl[r * batch + t], &hl[t], 4)) { if (bad < 8) printf("rep %d problem %d: %.9g vs %.9g\n", r, t, hl[r * batch + t], hl[t]); ++bad; }
    printf("mode %d lanczos %d: %d of %d lambda values differ from the first repetition (lambda[0] = %.6g)\n", mode, (int)lanczos, bad, (reps - 1) * batch, hl[0]);
    return 0;
}
