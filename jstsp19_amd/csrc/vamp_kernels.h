// Kernels of the VAMP-GLM iteration (VampGlmEst.m:354-495), shared by the fp32-storage path (vamp.hip: C2 = float2) and the float64
// path (vamp64.hip: C2 = double2).  The scalar recurrences (gam1x, gam1z, gam2x, gam2z, alf), the denoiser and the likelihood are
// computed in double in both; C2 / R are the storage types of the vectors and of q, d q, the factor eigenvalues.
#pragma once
#include "solver_common.h"
#include <cfloat>

namespace jstsp {

template <class C2> __device__ __forceinline__ C2 vmk(double x, double y);
template <> __device__ __forceinline__ float2 vmk<float2>(double x, double y) { return make_float2((float)x, (float)y); }
template <> __device__ __forceinline__ double2 vmk<double2>(double x, double y) { return make_double2(x, y); }

struct VampScal {
    double gam1x, gam1z, gam2x, gam2z, alf;
    double pad[3];
};

__device__ __forceinline__ double bsum(double v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// Bernoulli-Gaussian posterior of one real coordinate with the complex-branch likelihoods.
__device__ __forceinline__ void bg_denoise(double r, double rvar, double var0, double p1, double &xhat, double &xvar)
{
    const double PI = 3.14159265358979323846;
    const double r2 = r * r;
    const double ll1 = -(log(PI) + log(var0 + rvar) + r2 / (var0 + rvar));     // CAwgnEstimIn.m:181-184
    const double rv = fmax(rvar, DBL_EPSILON);                                  // SparseScaEstim.m:96
    const double ll0 = -(log(PI) + log(rv) + r2 / rv);                          // :100-103
    double ex = ll0 - ll1 + log(1.0 - p1) - log(p1);                            // :107
    ex = fmax(fmin(ex, 500.0), -500.0);                                         // :108-109
    const double py1 = 1.0 / (1.0 + exp(ex));                                   // :110
    const double gain = var0 / (var0 + rv);                                     // CAwgnEstimIn.m:100-102
    const double xh1 = gain * r, xv1 = gain * rv;
    xhat = py1 * xh1;                                                           // :160
    xvar = py1 * (xh1 * xh1 - xhat * xhat) + py1 * xv1 + (1.0 - py1) * (0.0 - xhat * xhat);   // :163-165
}

__device__ __forceinline__ double clipg(double g) { return fmin(fmax(g, 1e-8), 1e14); }   // VampGlmOpt.m:7-8

// First half of an iteration (VampGlmEst.m:354-398): one workgroup per problem.
// Dc, Da: number of entries of d and the order of its A-side factor - (Mc, Na) with d = eig(A A') for M <= N (:402-406),
// (Nc, Gr) with d = eig(A'A) for M > N (:407-411; vamp.m passes no opt.V, so VampGlmEst.m:196-218 recomputes V and d)
template <class C2, class R>
__global__ __launch_bounds__(256) void vamp_first_half_kernel(int Nc, int Mc, int Dc, int Da, int G2, int it, double damp,
                                                              double sigma, double Lnz, const C2 *y,
                                                              const C2 *r1, const C2 *p1, C2 *x1,
                                                              C2 *r2, C2 *p2, const R *lamA,
                                                              long long sLa, const R *lamB, long long sLb,
                                                              R *q, R *dq, VampScal *sc)
{
    __shared__ double sh[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    const long long bn = (long long)t * Nc, bm = (long long)t * Mc;
    VampScal s = sc[t];
    const double N = 2.0 * Nc;
    const double beta = Lnz / N, var0 = 1.0 / beta;                              // vamp.m:23-24 (xvar0 = 1)
    // ---- denoiser (:361) + damping (:363-365)
    double sv = 0;
    for (int e = tid; e < Nc; e += 256) {
        const C2 r = r1[bn + e];
        double xr, vr, xi, vi;
        bg_denoise((double)r.x, 1.0 / s.gam1x, var0, beta, xr, vr);
        bg_denoise((double)r.y, 1.0 / s.gam1x, var0, beta, xi, vi);
        sv += vr + vi;
        if (it > 0) {
            const C2 xo = x1[bn + e];
            xr = damp * xr + (1.0 - damp) * xo.x;
            xi = damp * xi + (1.0 - damp) * xo.y;
        }
        x1[bn + e] = vmk<C2>((R)xr, (R)xi);
    }
    sv = bsum(sv, sh);
    const double eta1x = 1.0 / (sv / N);                                         // :362
    const double g2x = eta1x - s.gam1x;                                          // :366
    for (int e = tid; e < Nc; e += 256) {                                        // :367 (unclipped gam2x)
        const C2 x = x1[bn + e], r = r1[bn + e];
        r2[bn + e] = vmk<C2>((R)((x.x * eta1x - r.x * s.gam1x) / g2x), (R)((x.y * eta1x - r.y * s.gam1x) / g2x));
    }
    const double gam2x = clipg(g2x);                                             // :376
    // ---- likelihood (:378-393): CAwgnEstimOut with scale 1
    const double pvar = 1.0 / s.gam1z, gain = pvar / (pvar + sigma);
    const double eta1z = 1.0 / (sigma * gain);
    const double g2z = eta1z - s.gam1z;
    for (int e = tid; e < Mc; e += 256) {
        const C2 p = p1[bm + e], yy = y[bm + e];
        const double zr = gain * (yy.x - p.x) + p.x, zi = gain * (yy.y - p.y) + p.y;
        p2[bm + e] = vmk<C2>((R)((zr * eta1z - p.x * s.gam1z) / g2z), (R)((zi * eta1z - p.y * s.gam1z) / g2z));
    }
    double gam2z = clipg(g2z);
    if (it > 0) gam2z = damp * gam2z + (1.0 - damp) * s.gam2z;                   // :391-393
    // ---- q = 1/(d + gam2x/gam2z), alf = (1/N) d'q - eps (:397-398); d_ij = sa_i^2 lb_j^2, each counted twice
    const double ratio = gam2x / gam2z;
    double acc = 0;
    const long long bd = (long long)t * Dc;
    for (int e = tid; e < Dc; e += 256) {
        const int i = e % Da, j = e / Da;
        const double lb = lamB[(long long)t * sLb + j];
        const double d = fmax((double)lamA[(long long)t * sLa + i], 0.0) * lb * lb;
        const double qq = 1.0 / (d + ratio);
        q[bd + e] = (R)qq;
        dq[bd + e] = (R)(d * qq);
        acc += d * qq;
    }
    acc = bsum(acc, sh);
    if (tid == 0) {
        s.gam2x = gam2x; s.gam2z = gam2z;
        s.alf = (2.0 / N) * acc - DBL_EPSILON;
        sc[t] = s;
    }
}

// t = Tm .* q ; dt = Tm .* (d q)
template <class C2, class R>
__global__ __launch_bounds__(256) void vamp_scale_kernel(long long n, const C2 *Tm, const R *q, const R *dq,
                                                         C2 *tq, C2 *tdq)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const C2 v = Tm[i];
        tq[i] = vmk<C2>(v.x * q[i], v.y * q[i]);
        tdq[i] = vmk<C2>(v.x * dq[i], v.y * dq[i]);
    }
}

// W += (gam2x / gam2z) r2   (the argument of V' in VampGlmEst.m:408; the ratio is a per-problem device scalar)
template <class C2, class R>
__global__ __launch_bounds__(256) void vamp_add_ratio_kernel(int Nc, C2 *W, const C2 *r2, const VampScal *sc)
{
    const int t = blockIdx.y;
    const R ratio = (R)(sc[t].gam2x / sc[t].gam2z);
    const long long b = (long long)t * Nc;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < Nc; e += gridDim.x * 256) {
        const C2 r = r2[b + e];
        C2 w = W[b + e];
        w.x += ratio * r.x; w.y += ratio * r.y;
        W[b + e] = w;
    }
}

template <class C2, class R>
__global__ __launch_bounds__(256) void vamp_sub_kernel(long long n, const C2 *a, const C2 *b, C2 *o)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        o[i] = vmk<C2>(a[i].x - b[i].x, a[i].y - b[i].y);
}

// Second half (:411-413, :464-492): damping of z2, new r1, p1, gam1x, gam1z.
template <class C2, class R>
__global__ __launch_bounds__(256) void vamp_second_half_kernel(int Nc, int Mc, int it, double damp, const C2 *x2,
                                                               const C2 *r2, C2 *z2, C2 *z2old,
                                                               const C2 *p2, C2 *r1, C2 *p1, VampScal *sc)
{
    const int t = blockIdx.x, tid = threadIdx.x;
    const long long bn = (long long)t * Nc, bm = (long long)t * Mc;
    VampScal s = sc[t];
    const double alf = s.alf, dl = (double)Mc / (double)Nc;                      // del = M/N (:257)
    for (int e = tid; e < Nc; e += 256) {                                        // :464
        const C2 a = x2[bn + e], b = r2[bn + e];
        r1[bn + e] = vmk<C2>((R)((a.x - b.x * (1.0 - alf)) / alf), (R)((a.y - b.y * (1.0 - alf)) / alf));
    }
    for (int e = tid; e < Mc; e += 256) {
        C2 z = z2[bm + e];
        if (it > 0) {                                                            // :411-413
            const C2 zo = z2old[bm + e];
            z = vmk<C2>((R)(damp * z.x + (1.0 - damp) * zo.x), (R)(damp * z.y + (1.0 - damp) * zo.y));
        }
        z2old[bm + e] = z;
        const C2 pp = p2[bm + e];                                            // :465
        p1[bm + e] = vmk<C2>((R)((dl * z.x - pp.x * alf) / (dl - alf)), (R)((dl * z.y - pp.y * alf) / (dl - alf)));
    }
    if (tid == 0) {
        double g1x = clipg(s.gam2x * alf / (1.0 - alf));                         // :469,:479
        const double g1z = clipg(s.gam2z * (dl - alf) / alf);                    // :480,:489
        if (it > 0) g1x = damp * g1x + (1.0 - damp) * s.gam1x;                   // :490-492
        s.gam1x = g1x; s.gam1z = g1z;
        sc[t] = s;
    }
}


template <int DUMMY = 0> __global__ void vamp_init_kernel(int batch, VampScal *sc)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < batch) {
        VampScal s;
        s.gam1x = 1e-8; s.gam1z = 1e-8;            // VampGlmOpt.m:25,27
        s.gam2x = 0; s.gam2z = 0; s.alf = 0;
        sc[t] = s;
    }
}

}  // namespace jstsp
