// Fused element-wise updates and wave64-shuffle reductions of the ADMM solvers
// (proposed_algorithm.m:35-69, proposed_algorithm_angles.m:36,68).  All HBM-bound: each
// kernel makes one pass over its operands with 16-byte (two complex) accesses per lane.
#include "common.h"
#include <algorithm>

namespace jstsp {

struct c2 { float2 a, b; };   // two interleaved complex values = one 16-byte access

__device__ __forceinline__ c2 ld2(const float2 *p, long long i)
{
    const float4 v = *reinterpret_cast<const float4 *>(p + i);
    c2 r; r.a = make_float2(v.x, v.y); r.b = make_float2(v.z, v.w); return r;
}
__device__ __forceinline__ void st2(float2 *p, long long i, const c2 &v)
{
    *reinterpret_cast<float4 *>(p + i) = make_float4(v.a.x, v.a.y, v.b.x, v.b.y);
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// blockDim = 256: returns the block-wide sum in every thread.  `sh` = 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- Z = X - V1/rho  (argument of svt, proposed_algorithm.m:35) --------------------------
__global__ __launch_bounds__(256) void form_z_kernel(long long nm, const float2 *X, const float2 *V1,
                                                     const TrialParams *prm, float2 *Z)
{
    const int t = blockIdx.y;
    const TrialParams p = prm[t];
    const long long base = (long long)t * nm;
    const long long stride = (long long)gridDim.x * 256 * 2;
    const bool vec = (nm & 1) == 0;     // per-problem bases stay 16-byte aligned only for even nm
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; i < nm; i += stride) {
        if (vec && i + 1 < nm) {
            const c2 x = ld2(X, base + i), v = ld2(V1, base + i);
            c2 z;
            z.a = make_float2(admm_z(p, x.a.x, v.a.x), admm_z(p, x.a.y, v.a.y));
            z.b = make_float2(admm_z(p, x.b.x, v.b.x), admm_z(p, x.b.y, v.b.y));
            st2(Z, base + i, z);
        } else {
            for (long long j = i; j < nm && j < i + 2; ++j) {
                const float2 x = X[base + j], v = V1[base + j];
                Z[base + j] = make_float2(admm_z(p, x.x, v.x), admm_z(p, x.y, v.y));
            }
        }
    }
}

// ---- sub-problem 2 + first half of sub-problem 3 + V1 dual update -------------------------
//   X  = (V1 + rho Y + subY + V2 + rho C + rho Xs) ./ (Omega + 2 rho)      (:38-40)
//   K  = X - V2/rho - C                                                      (:43)
//   V1 = V1 + rho (Y - X)                                                    (:64; depends only on Y, X)
__device__ __forceinline__ void upd_x_one(float2 &x, float2 &v1, const float2 v2, const float2 c,
                                          const float2 xs, const float2 y, const float2 sy,
                                          const float id, const TrialParams &p, float2 &k)
{
    // (rho and 1/rho as two floats each: common.h)
    const float tx = (y.x + c.x) + xs.x, ty = (y.y + c.y) + xs.y;
    const float bx = (v1.x + sy.x) + v2.x + mul2(p.rho, p.rho_lo, tx);
    const float by = (v1.y + sy.y) + v2.y + mul2(p.rho, p.rho_lo, ty);
    x = make_float2(bx * id, by * id);
    k = make_float2((fmaf(-p.irho, v2.x, x.x) - p.irho_lo * v2.x) - c.x, (fmaf(-p.irho, v2.y, x.y) - p.irho_lo * v2.y) - c.y);
    v1 = make_float2(admm_v1(p, v1.x, y.x, x.x), admm_v1(p, v1.y, y.y, x.y));
}

__global__ __launch_bounds__(256) void update_x_kernel(long long nm, float2 *X, float2 *V1,
                                                       const float2 *V2, const float2 *C,
                                                       const float2 *Xs, const float2 *Y,
                                                       const float2 *subY, const float *invD,
                                                       const TrialParams *prm, float2 *K)
{
    const int t = blockIdx.y;
    const TrialParams p = prm[t];
    const long long base = (long long)t * nm;
    const long long stride = (long long)gridDim.x * 256 * 2;
    const bool vec = (nm & 1) == 0;     // per-problem bases stay 16-byte aligned only for even nm
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; i < nm; i += stride) {
        if (vec && i + 1 < nm) {
            c2 x, k, v1 = ld2(V1, base + i);
            const c2 v2 = ld2(V2, base + i), c = ld2(C, base + i), xs = ld2(Xs, base + i),
                     y = ld2(Y, base + i), sy = ld2(subY, base + i);
            const float2 id = *reinterpret_cast<const float2 *>(invD + base + i);
            upd_x_one(x.a, v1.a, v2.a, c.a, xs.a, y.a, sy.a, id.x, p, k.a);
            upd_x_one(x.b, v1.b, v2.b, c.b, xs.b, y.b, sy.b, id.y, p, k.b);
            st2(X, base + i, x); st2(V1, base + i, v1); st2(K, base + i, k);
        } else {
            for (long long j = i; j < nm && j < i + 2; ++j) {
                float2 x, k, v1 = V1[base + j];
                upd_x_one(x, v1, V2[base + j], C[base + j], Xs[base + j], Y[base + j], subY[base + j],
                          invD[base + j], p, k);
                X[base + j] = x; V1[base + j] = v1; K[base + j] = k;
            }
        }
    }
}

// ---- sub-problem 4 + V2 dual update --------------------------------------------------------
//   C  = rho/(rho+1) (X - Xs - V2/rho)        (:61)
//   V2 = V2 + rho (C - X + Xs)                (:65)
__device__ __forceinline__ void upd_c_one(const float2 x, const float2 xs, float2 &v2, float2 &c, const TrialParams &p)
{
    // (rho, 1/rho and rho/(rho+1) as two floats each: common.h)
    const float dx = x.x - xs.x, dy = x.y - xs.y;
    const float tx = fmaf(-p.irho, v2.x, dx) - p.irho_lo * v2.x, ty = fmaf(-p.irho, v2.y, dy) - p.irho_lo * v2.y;
    c = make_float2(mul2(p.c_coef, p.c_lo, tx), mul2(p.c_coef, p.c_lo, ty));
    const float ux = c.x - dx, uy = c.y - dy;
    v2 = make_float2(fmaf(p.rho, ux, v2.x) + p.rho_lo * ux, fmaf(p.rho, uy, v2.y) + p.rho_lo * uy);
}

__global__ __launch_bounds__(256) void update_c_kernel(long long nm, const float2 *X, const float2 *Xs,
                                                       float2 *V2, float2 *C, const TrialParams *prm)
{
    const int t = blockIdx.y;
    const TrialParams p = prm[t];
    const long long base = (long long)t * nm;
    const long long stride = (long long)gridDim.x * 256 * 2;
    const bool vec = (nm & 1) == 0;     // per-problem bases stay 16-byte aligned only for even nm
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; i < nm; i += stride) {
        if (vec && i + 1 < nm) {
            const c2 x = ld2(X, base + i), xs = ld2(Xs, base + i);
            c2 v2 = ld2(V2, base + i), c;
            upd_c_one(x.a, xs.a, v2.a, c.a, p);
            upd_c_one(x.b, xs.b, v2.b, c.b, p);
            st2(C, base + i, c); st2(V2, base + i, v2);
        } else {
            for (long long j = i; j < nm && j < i + 2; ++j) {
                float2 v2 = V2[base + j], c;
                upd_c_one(X[base + j], Xs[base + j], v2, c, p);
                C[base + j] = c; V2[base + j] = v2;
            }
        }
    }
}

// ---- invD = 1 ./ (Omega + scale*rho)   (iK1 of :14-20 with scale 2; mc_admm.m:11-17 with 1) --
__global__ __launch_bounds__(256) void inv_d_kernel(long long nm, const float *Omega, float scale,
                                                    const TrialParams *prm, float *invD)
{
    const int t = blockIdx.y;
    const double add = (double)scale * ((double)prm[t].rho + (double)prm[t].rho_lo);      // (rho as two floats: common.h)
    const long long base = (long long)t * nm;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nm; i += stride)
        invD[base + i] = (float)(1.0 / ((double)Omega[base + i] + add));
}

__device__ __forceinline__ float soft1(float v, float t)
{
    // max(|v| - t, 0) * sign(v), sign(0) = 0   (proposed_algorithm.m:56)
    const float m = fmaxf(fabsf(v) - t, 0.f);
    return (v > 0.f) ? m : ((v < 0.f) ? -m : 0.f);
}

// ---- steepest-descent step of the 'approximate' branch + soft threshold (:48-56) ---------
// One workgroup per problem: alpha = <res,res> / <res, R res> by wave64 shuffle reductions
// (fp64 accumulation), then v += alpha res, ce(i,3) = |dv|^2/|v_prev|^2, s = soft(v) (.* mask).
// (NT = 1024 for long vectors: one 256-thread workgroup per problem keeps too few loads in flight for 256 KiB arrays)
template <int NT>
__device__ __forceinline__ double block_sum_nt(double v, double *sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += sh[w];
    return s;
}
template <int NT>
__global__ __launch_bounds__(NT) void step_v_kernel(int g, const float2 *Res, const float2 *RRes,
                                                    float2 *V, float2 *S, const int32_t *rank,
                                                    int cnt, const TrialParams *prm, double *ce3,
                                                    int Imax, int it, float2 *RV, float2 *Vlo, float2 *RVlo, int vlo_reset)
{
    // Vlo / RVlo != NULL: v and R v are carried as two floats each (hi in V / RV): the sums of alpha res and alpha R res are
    // accumulated to about 48 bits (in float64 here, split again on the way out), so that R v stays R times THE v that was
    // accumulated instead of drifting from it by one fp32 rounding of each per iteration.  s = soft(v) sees the leading part.
    // vlo_reset: R v has just been recomputed from the leading part of v - the stored low-order part of v is dropped.
    __shared__ double sh[NT / 64];
    const int t = blockIdx.x;
    const long long base = (long long)t * g;
    double num = 0, den_re = 0, den_im = 0, vprev = 0;
    // Eight strides at a time with all their loads issued first: written as a plain loop, every trip waited for its own three
    // loads (64 trips x ~1 us at the headline shape: the kernel took 150 us alone whatever else ran).  Same sums in the same order.
    constexpr int UN = 8;
    for (int i0 = threadIdx.x; i0 < g; i0 += NT * UN) {
        float2 r[UN], rr[UN], v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = min(i0 + u * NT, g - 1);
            r[u] = Res[base + i]; rr[u] = RRes[base + i]; v[u] = V[base + i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (i0 + u * NT < g) {
                num += (double)r[u].x * r[u].x + (double)r[u].y * r[u].y;
                den_re += (double)r[u].x * rr[u].x + (double)r[u].y * rr[u].y;      // Re(conj(r) * rr)
                den_im += (double)r[u].x * rr[u].y - (double)r[u].y * rr[u].x;      // Im(conj(r) * rr)
                vprev += (double)v[u].x * v[u].x + (double)v[u].y * v[u].y;
            }
    }
    num = block_sum_nt<NT>(num, sh);
    den_re = block_sum_nt<NT>(den_re, sh);
    den_im = block_sum_nt<NT>(den_im, sh);
    vprev = block_sum_nt<NT>(vprev, sh);
    // alpha = num / (den_re + i den_im): complex scalar exactly as `res'*res/(res'*R*res)` (:48)
    const double dd = den_re * den_re + den_im * den_im;
    const float ax = (float)(num * den_re / dd);
    const float ay = (float)(-num * den_im / dd);
    const float thr = prm[t].tauS_rho;
    if (!Vlo) {
        for (int i0 = threadIdx.x; i0 < g; i0 += NT * UN) {
            float2 r[UN], v[UN], rr[UN], rv[UN];
            int rk[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = min(i0 + u * NT, g - 1);
                r[u] = Res[base + i]; v[u] = V[base + i];
                if (RV) { rr[u] = RRes[base + i]; rv[u] = RV[base + i]; }
                rk[u] = rank ? rank[base + i] : 0;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * NT;
                if (i >= g) break;
                float2 vn = v[u];
                vn.x += ax * r[u].x - ay * r[u].y;
                vn.y += ax * r[u].y + ay * r[u].x;
                V[base + i] = vn;
                if (RV) {       // R v_new = R v + alpha R res: carried between the periodic recomputations of R v (proposed.hip)
                    float2 rn = rv[u];
                    rn.x += ax * rr[u].x - ay * rr[u].y;
                    rn.y += ax * rr[u].y + ay * rr[u].x;
                    RV[base + i] = rn;
                }
                float2 sv = make_float2(soft1(vn.x, thr), soft1(vn.y, thr));
                if (rank && rk[u] >= cnt) sv = make_float2(0.f, 0.f);
                S[base + i] = sv;
            }
        }
    } else
    for (int i = threadIdx.x; i < g; i += NT) {
        const float2 r = Res[base + i];
        float2 v = V[base + i];
        if (Vlo) {
            const float2 vl = vlo_reset ? make_float2(0.f, 0.f) : Vlo[base + i];
            const double sx = ((double)v.x + (double)vl.x) + ((double)ax * r.x - (double)ay * r.y);
            const double sy = ((double)v.y + (double)vl.y) + ((double)ax * r.y + (double)ay * r.x);
            v = make_float2((float)sx, (float)sy);
            Vlo[base + i] = make_float2((float)(sx - (double)v.x), (float)(sy - (double)v.y));
            const float2 rr = RRes[base + i], rv = RV[base + i], rl = RVlo[base + i];
            const double tx = ((double)rv.x + (double)rl.x) + ((double)ax * rr.x - (double)ay * rr.y);
            const double ty = ((double)rv.y + (double)rl.y) + ((double)ax * rr.y + (double)ay * rr.x);
            const float2 rn = make_float2((float)tx, (float)ty);
            RV[base + i] = rn;
            RVlo[base + i] = make_float2((float)(tx - (double)rn.x), (float)(ty - (double)rn.y));
            V[base + i] = v;
        } else {
        v.x += ax * r.x - ay * r.y;
        v.y += ax * r.y + ay * r.x;
        V[base + i] = v;
        if (RV) {       // R v_new = R v + alpha R res: carried between the periodic recomputations of R v (proposed.hip)
            const float2 rr = RRes[base + i];
            float2 rv = RV[base + i];
            rv.x += ax * rr.x - ay * rr.y;
            rv.y += ax * rr.y + ay * rr.x;
            RV[base + i] = rv;
        }
        }
        float2 s = make_float2(soft1(v.x, thr), soft1(v.y, thr));
        if (rank && rank[base + i] >= cnt) s = make_float2(0.f, 0.f);
        S[base + i] = s;
    }
    if (ce3 && threadIdx.x == 0) {
        // |v - v_prev|^2 = |alpha|^2 |res|^2 ; 0-divide at i = 1 gives Inf (NaN if res = 0) as in :51
        const double a2 = (double)ax * ax + (double)ay * ay;
        ce3[(long long)t * 3 * Imax + 2 * Imax + it] = a2 * num / vprev;
    }
}

// ---- s = soft(v) (.* mask) only ('std' branch, :53-56) --------------------------------------
__global__ __launch_bounds__(256) void soft_kernel(int g, const float2 *V, float2 *S,
                                                   const int32_t *rank, int cnt, const TrialParams *prm)
{
    const int t = blockIdx.y;
    const long long base = (long long)t * g;
    const float thr = prm[t].tauS_rho;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < g; i += gridDim.x * 256) {
        const float2 v = V[base + i];
        float2 s = make_float2(soft1(v.x, thr), soft1(v.y, thr));
        if (rank && rank[base + i] >= cnt) s = make_float2(0.f, 0.f);
        S[base + i] = s;
    }
}

// ---- rank[pos] = first position of linear index pos+1 in indx_S (angles :36) ---------------
__global__ void rank_init_kernel(long long n, int32_t *rank)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rank[i] = 0x7fffffff;
}
__global__ void rank_scatter_kernel(int g, const int32_t *indx, int32_t *rank)
{
    const int t = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < g) {
        const int pos = indx[(long long)t * g + p] - 1;
        if (pos >= 0 && pos < g) atomicMin(&rank[(long long)t * g + pos], p);
    }
}

// ---- ce(i,1:2) = lambda_max(V1 V1^H)/lambda_max(X X^H), lambda_max(V2 V2^H)/... (:67,69) ---
__global__ void ce_ratio_kernel(int batch, const float *lamV1, const float *lamV2, const float *lamX,
                                double *ce, int Imax, int it)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < batch) {
        ce[(long long)t * 3 * Imax + it] = (double)lamV1[t] / (double)lamX[t];
        ce[(long long)t * 3 * Imax + Imax + it] = (double)lamV2[t] / (double)lamX[t];
    }
}

static inline dim3 ew_grid(long long nm, int batch)
{
    long long blocks = (nm / 2 + 255) / 256;
    if (blocks < 1) blocks = 1;
    // enough workgroups to fill the chip (256 CUs x 8) without a tail
    const long long cap = (2048 + batch - 1) / batch;
    if (blocks > cap) blocks = cap < 1 ? 1 : cap;
    return dim3((unsigned)blocks, (unsigned)batch);
}

int launch_form_z(jstsp_ctx *ctx, long long nm, int batch, const float2 *X, const float2 *V1,
                  const TrialParams *prm, float2 *Z)
{
    hipLaunchKernelGGL(form_z_kernel, ew_grid(nm, batch), dim3(256), 0, ctx->stream, nm, X, V1, prm, Z);
    JSTSP_HIP(hipGetLastError());
    return 0;
}
int launch_update_x(jstsp_ctx *ctx, long long nm, int batch, float2 *X, float2 *V1, const float2 *V2,
                    const float2 *C, const float2 *Xs, const float2 *Y, const float2 *subY,
                    const float *invD, const TrialParams *prm, float2 *K)
{
    hipLaunchKernelGGL(update_x_kernel, ew_grid(nm, batch), dim3(256), 0, ctx->stream, nm, X, V1, V2, C,
                       Xs, Y, subY, invD, prm, K);
    JSTSP_HIP(hipGetLastError());
    return 0;
}
int launch_update_c(jstsp_ctx *ctx, long long nm, int batch, const float2 *X, const float2 *Xs,
                    float2 *V2, float2 *C, const TrialParams *prm)
{
    hipLaunchKernelGGL(update_c_kernel, ew_grid(nm, batch), dim3(256), 0, ctx->stream, nm, X, Xs, V2, C,
                       prm);
    JSTSP_HIP(hipGetLastError());
    return 0;
}
int launch_step_v(jstsp_ctx *ctx, int g, int batch, const float2 *Res, const float2 *RRes, float2 *V,
                  float2 *S, const int32_t *rank, int cnt, const TrialParams *prm, double *ce3,
                  int Imax, int it, float2 *RV, int waves8, float2 *Vlo, float2 *RVlo, int vlo_reset)
{
    // waves8 (the caller runs an eigen-decomposition beside this kernel): eight waves per problem - a workgroup that fits on a
    // CU beside a resident Jacobi, 16 waves x 106 registers do not; alone, the 16-wave form is 25 % faster
    if (g >= 8192 && waves8)
        hipLaunchKernelGGL(step_v_kernel<512>, dim3(batch), dim3(512), 0, ctx->stream, g, Res, RRes, V, S, rank,
                           cnt, prm, ce3, Imax, it, RV, Vlo, RVlo, vlo_reset);
    else if (g >= 8192)
        hipLaunchKernelGGL(step_v_kernel<1024>, dim3(batch), dim3(1024), 0, ctx->stream, g, Res, RRes, V, S, rank,
                           cnt, prm, ce3, Imax, it, RV, Vlo, RVlo, vlo_reset);
    else
        hipLaunchKernelGGL(step_v_kernel<256>, dim3(batch), dim3(256), 0, ctx->stream, g, Res, RRes, V, S, rank,
                           cnt, prm, ce3, Imax, it, RV, Vlo, RVlo, vlo_reset);
    JSTSP_HIP(hipGetLastError());
    return 0;
}
int launch_soft(jstsp_ctx *ctx, int g, int batch, const float2 *V, float2 *S, const int32_t *rank,
                int cnt, const TrialParams *prm)
{
    hipLaunchKernelGGL(soft_kernel, dim3((g + 255) / 256, batch), dim3(256), 0, ctx->stream, g, V, S,
                       rank, cnt, prm);
    JSTSP_HIP(hipGetLastError());
    return 0;
}
int launch_inv_d(jstsp_ctx *ctx, long long nm, int batch, const float *Omega, float scale,
                 const TrialParams *prm, float *invD)
{
    hipLaunchKernelGGL(inv_d_kernel, ew_grid(nm * 2, batch), dim3(256), 0, ctx->stream, nm, Omega, scale,
                       prm, invD);
    JSTSP_HIP(hipGetLastError());
    return 0;
}
int launch_rank_from_index(jstsp_ctx *ctx, int g, int batch, const int32_t *indx, int32_t *rank)
{
    const long long n = (long long)g * batch;
    hipLaunchKernelGGL(rank_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       rank);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3((g + 255) / 256, batch), dim3(256), 0, ctx->stream, g,
                       indx, rank);
    JSTSP_HIP(hipGetLastError());
    return 0;
}
// P = I - Q for `count` n x n matrices (svt(Z) = Z - Q Z = P Z: one read of Z instead of two in the fused update)
__global__ __launch_bounds__(256) void eye_minus_kernel(int n, long long total, const float2 *Q, float2 *P)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int r = (int)(e % ((long long)n * n));
        const float2 q = Q[e];
        P[e] = make_float2(((r % n) == (r / n) ? 1.f : 0.f) - q.x, -q.y);
    }
}
int launch_eye_minus(jstsp_ctx *ctx, int n, int count, const float2 *Q, float2 *P)
{
    const long long total = (long long)count * n * n;
    hipLaunchKernelGGL(eye_minus_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0,
                       ctx->stream, n, total, Q, P);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

int launch_ce_ratio(jstsp_ctx *ctx, int batch, const float *lamV1, const float *lamV2, const float *lamX,
                    double *ce, int Imax, int it)
{
    hipLaunchKernelGGL(ce_ratio_kernel, dim3((batch + 255) / 256), dim3(256), 0, ctx->stream, batch,
                       lamV1, lamV2, lamX, ce, Imax, it);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
