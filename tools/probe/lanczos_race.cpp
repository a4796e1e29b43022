// Is lambda_max (Lanczos kernel) bit-reproducible when other kernels run beside it?  One stream repeats launch_lmax on the SAME
// Gram partials; a second stream runs (a) nothing, (b) the three-Gram pass, (c) a dummy LDS-heavy kernel.  Build (in jstsp19_amd/csrc):
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

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 1, reps = 200;
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
        ctx->stream = sA;
        launch_lmax(ctx, n, batch, Gw, (long long)n * n * nsplit, nsplit, (long long)n * n, lam + (size_t)r * batch, true);
    }
    hipDeviceSynchronize();
    std::vector<float> hl((size_t)reps * batch);
    hipMemcpy(hl.data(), lam, hl.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 1; r < reps; ++r)
        for (int t = 0; t < batch; ++t)
            if (memcmp(&hl[r * batch + t], &hl[t], 4)) { if (bad < 8) printf("rep %d problem %d: %.9g vs %.9g\n", r, t, hl[r * batch + t], hl[t]); ++bad; }
    printf("mode %d: %d of %d lambda values differ from the first repetition (lambda[0] = %.6g)\n", mode, bad, (reps - 1) * batch, hl[0]);
    return 0;
}
