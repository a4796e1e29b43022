// gram64.hip - Gram matrices of the dictionary factors with float64 accumulation.
//
// R = K2'*K2 = (B*B').' (x) (A'*A)  (proposed_algorithm.m:25) enters `R*v` in EVERY iteration (:47), next to `K2'*k`
// computed through A and B themselves: an error of G_A = A'A or G_B = B B' is not rounding noise that averages out over
// the iterations, it is a constant bias of the gradient.  Measured (tools/precision_study.py, float64 port with ONE
// quantity perturbed, 16 full-size trials): entries of G_A off by 1.7e-7 relative (rms) - what a 64-term fp32 chain leaves -
// move the NMSE by 3.0e-7 rms (7e-7 max), the same on G_B by 2.0e-7; the fp32 STORAGE of every array of the iteration
// together by 0.9e-7, the accumulation error of the three big per-iteration products by 0.2e-7 each.  So the two Grams are
// formed here from the fp32 inputs with float64 products and sums (exact to 1e-16, then rounded ONCE to fp32).
#include "solver_common.h"

namespace jstsp {

// G[t][i + n j] = sum_k a(i,k) conj(a(j,k)),  a(i,k) = X[t sXt + i si + k sk] (conjugated if cj)
// 16 x 16 outputs per workgroup, k in panels of 16 staged through LDS as float64.
__global__ __launch_bounds__(256) void gram64_kernel(const float2 *X, long long sXt, long long si, long long sk, int cj, int n,
                                                     int kdim, float2 *G, long long sGt, float2 *Glo)
{
    __shared__ double ar[16][17], ai[16][17], br[16][17], bi[16][17];
    const int t = blockIdx.z, i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    if (j0 < i0) return;                                   // Hermitian: tiles on and above the diagonal, mirrored below
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const float2 *x = X + (long long)t * sXt;
    // the faster-varying thread index runs along the contiguous direction of X
    const bool k_fast = sk < si;
    double cr = 0.0, ci = 0.0;
    for (int k0 = 0; k0 < kdim; k0 += 16) {
        {
            const int ii = k_fast ? ty : tx, kk = k_fast ? tx : ty;
            float2 va = make_float2(0.f, 0.f), vb = make_float2(0.f, 0.f);
            if (k0 + kk < kdim) {
                if (i0 + ii < n) va = x[(long long)(i0 + ii) * si + (long long)(k0 + kk) * sk];
                if (j0 + ii < n) vb = x[(long long)(j0 + ii) * si + (long long)(k0 + kk) * sk];
            }
            ar[ii][kk] = va.x; ai[ii][kk] = cj ? -(double)va.y : (double)va.y;
            br[ii][kk] = vb.x; bi[ii][kk] = cj ? -(double)vb.y : (double)vb.y;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const double pr = ar[tx][kk], pi = ai[tx][kk], qr = br[ty][kk], qi = bi[ty][kk];
            cr += pr * qr + pi * qi;                       // a conj(b)
            ci += pi * qr - pr * qi;
        }
        __syncthreads();
    }
    const int i = i0 + tx, j = j0 + ty;
    if (i < n && j < n) {
        float2 *g = G + (long long)t * sGt;
        if (i == j) ci = 0.0;
        const float hr = (float)cr, hi = (float)ci;
        g[i + (long long)n * j] = make_float2(hr, hi);
        if (i0 != j0) g[j + (long long)n * i] = make_float2(hr, -hi);
        if (Glo) {                                         // G + Glo = the float64 sum to 2^-48
            float2 *gl = Glo + (long long)t * sGt;
            const float lr = (float)(cr - (double)hr), li = (float)(ci - (double)hi);
            gl[i + (long long)n * j] = make_float2(lr, li);
            if (i0 != j0) gl[j + (long long)n * i] = make_float2(lr, -li);
        }
    }
}

// side 'L': G = X^H X (cols x cols);  side 'R': G = X X^H (rows x rows).  X: rows x cols column-major, ld = rows.
int gram_f64(jstsp_ctx *ctx, char side, const float2 *X, long long sXt, int rows, int cols, int count, float2 *G, long long sGt,
             float2 *Glo)
{
    const int n = side == 'L' ? cols : rows, kdim = side == 'L' ? rows : cols;
    const long long si = side == 'L' ? rows : 1, sk = side == 'L' ? 1 : rows;
    const dim3 grid((n + 15) / 16, (n + 15) / 16, count);
    hipLaunchKernelGGL(gram64_kernel, grid, dim3(256), 0, ctx->stream, X, sXt, si, sk, side == 'L' ? 1 : 0, n, kdim, G, sGt, Glo);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
