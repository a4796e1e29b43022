// Device-side construction of the solver inputs for a batch of Monte-Carlo trials — the caller
// side of the hot path (plot_errorVSsnr.m:57-136): channel, pilots, noise, the random spatial
// sampling measurement, the dictionary factors and the hyper-parameters, all born in HBM.
//
//   wideband_mmwave_channel.m:1-39  -> channel_kernel (+ two GEMMs for Zbar = Dr' H_l Dt)
//   qam4mod.m:7-8, plot_errorVSsnr.m:63-67, proposed_hbf.m:15-18 -> pilots_kernel (rows of the Hermitian Toeplitz)
//   proposed_hbf.m:19-22            -> R = [H_1 .. H_L] Psi + sqrt(var/2) noise      (one GEMM, noise as the beta term)
//   proposed_hbf.m:36-42            -> omega_kernel (Mr smallest of Mr_e uniform keys per column), mask_kernel
//   plot_errorVSsnr.m:127-130       -> hyper_kernel (tau_Y, tau_Z, rho from the 6th largest eigenvalue of Y'Y)
//   plot_errorVSsnr.m:132-136       -> A = W_e' Dr, B_l = Dt' Psi_l                   (GEMMs)
//   plot_errorVSsnr.m:143           -> indx_S: stable descending sort of |vec(Zbar)| (64-bit keys, segmented radix sort)
//
// Random numbers: Philox4x32-10, key = mix(seed, sweep index, global trial index), counter =
// (element index, stream id) — a trial's inputs do not depend on the batch it is drawn in or
// on how trials are sharded over GPUs.
#include "solver_common.h"
#include <hipcub/hipcub.hpp>

using namespace jstsp;

namespace {

enum { ST_GAIN = 0, ST_UR = 1, ST_UT = 2, ST_NOISE = 3, ST_QAM = 4, ST_OMEGA = 5, ST_PILOT = 6 };

__host__ __device__ inline uint64_t mix_key(uint64_t seed, uint64_t sweep, uint64_t trial)
{
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + sweep * 0xBF58476D1CE4E5B9ull + trial * 0x94D049BB133111EBull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ uint4 philox(uint64_t elem, uint32_t stream, uint64_t key64)
{
    uint32_t c0 = (uint32_t)elem, c1 = (uint32_t)(elem >> 32), c2 = stream, c3 = 0u;
    uint32_t k0 = (uint32_t)key64, k1 = (uint32_t)(key64 >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float u01(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// two independent N(0,1) from two words (Box-Muller)
__device__ __forceinline__ float2 normal2(uint32_t w0, uint32_t w1)
{
    const float r = sqrtf(-2.0f * logf(u01(w0)));
    float s, c;
    sincospif(2.0f * u01(w1), &s, &c);
    return make_float2(r * c, r * s);
}

struct Model {
    int Nt, Nr, L, Tp, Mr, Mr_e, Gr, Gt, clusters, rays, Np, NtL, G2;
};

// ---- gains / angle draws (wideband_mmwave_channel.m:19-22; only tap 1's angles are used, :24) ----
__global__ void draw_small_kernel(Model m, uint64_t seed, uint64_t sweep, long long trial0, float2 *gains,
                                  float *u_r, float *u_t)
{
    const int t = blockIdx.x;
    const uint64_t key = mix_key(seed, sweep, (uint64_t)(trial0 + t));
    for (int i = threadIdx.x; i < m.L * m.Np; i += blockDim.x) {
        const uint4 w = philox((uint64_t)i, ST_GAIN, key);
        const float2 g = normal2(w.x, w.y);
        gains[(size_t)t * m.L * m.Np + i] = make_float2(g.x * 0.70710678f, g.y * 0.70710678f);     // :19
    }
    for (int i = threadIdx.x; i < m.Np; i += blockDim.x) {
        u_r[(size_t)t * m.Np + i] = u01(philox((uint64_t)i, ST_UR, key).x);                          // :20
        u_t[(size_t)t * m.Np + i] = u01(philox((uint64_t)i, ST_UT, key).x);                          // :22
    }
}

// noise = randn + 1j*randn (plot_errorVSsnr.m:60, unscaled) and the 4-QAM symbol indices (qam4mod.m:8)
// gauss != 0: Gaussian pilot draws randn + 1j*randn into psym instead (wideband_hybBF_comm_system_training.m:20, before its 1/sqrt(2))
__global__ __launch_bounds__(256) void draw_noise_qam_kernel(Model m, uint64_t seed, uint64_t sweep, long long trial0,
                                                             float2 *noise, uint8_t *qam, int shared_pilots, int gauss,
                                                             float2 *psym)
{
    const int t = blockIdx.y;
    const uint64_t key = mix_key(seed, sweep, (uint64_t)(trial0 + t));
    const uint64_t qkey = shared_pilots ? mix_key(seed, sweep, ~0ull) : key;     // one pilot set per sweep point
    const long long nn = (long long)m.Nr * m.Tp, nq = (long long)m.Nt * m.Tp;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nn; i += stride) {
        const uint4 w = philox((uint64_t)i, ST_NOISE, key);
        noise[(size_t)t * nn + i] = normal2(w.x, w.y);
    }
    const float a = 0.70710678f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += stride) {
        if (gauss) {
            const uint4 w = philox((uint64_t)i, ST_PILOT, qkey);
            psym[(size_t)t * nq + i] = normal2(w.x, w.y);
            qam[(size_t)t * nq + i] = 0;
        } else {
            const uint8_t q = (uint8_t)(philox((uint64_t)i, ST_QAM, qkey).x & 3u);
            qam[(size_t)t * nq + i] = q;
            // alphabet order of qam4mod.m:7: (1+j), (-1+j), (1-j), (-1-j), all / sqrt(2)
            psym[(size_t)t * nq + i] = make_float2((q & 1) ? -a : a, (q & 2) ? -a : a);
        }
    }
}

// Omega(:, j): ones on the Mr rows with the smallest of Mr_e uniform keys (= randperm(Mr_e)(1:Mr), proposed_hbf.m:37-40)
__global__ __launch_bounds__(256) void omega_kernel(Model m, uint64_t seed, uint64_t sweep, long long trial0,
                                                    float *Omega)
{
    extern __shared__ uint32_t keys[];
    const int t = blockIdx.y, j = blockIdx.x;
    const uint64_t key = mix_key(seed, sweep, (uint64_t)(trial0 + t));
    for (int i = threadIdx.x; i < m.Mr_e; i += 256)
        keys[i] = philox((uint64_t)j * m.Mr_e + i, ST_OMEGA, key).x;
    __syncthreads();
    for (int i = threadIdx.x; i < m.Mr_e; i += 256) {
        const uint32_t ki = keys[i];
        int rank = 0;
        for (int k = 0; k < m.Mr_e; ++k) {
            const uint32_t kk = keys[k];
            rank += (kk < ki) || (kk == ki && k < i);
        }
        Omega[((size_t)t * m.Tp + j) * m.Mr_e + i] = rank < m.Mr ? 1.f : 0.f;
    }
}

// ---- dictionaries (trial-independent): Dr, Dt (wideband_mmwave_channel.m:9-10), ZC beamformer (createBeamformer.m:15-16)
__global__ void dict_kernel(int rows, int cols, int kind, float2 *D)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    const int n = (int)(i % rows), g = (int)(i / rows);
    float s, c;
    if (kind == 0) {            // 1/sqrt(rows) exp(-j 2 pi n g / cols)
        const long long r = ((long long)n * g) % cols;
        sincospif(-2.0f * (float)r / (float)cols, &s, &c);
    } else {                    // 1/sqrt(N) exp(-j 11 pi n (g+1) / N),  N = rows
        const long long r = (11ll * n * (g + 1)) % (2ll * rows);
        sincospif(-(float)r / (float)rows, &s, &c);
    }
    const float sc = rsqrtf((float)rows);
    D[i] = make_float2(c * sc, s * sc);
}

// ---- channel: Hmat[t] = [H_1 ... H_L]  (Nr x Nt*L), H_l = 1/sqrt(Np) sum_p w_p g[l,p] a_r(p) a_t(p)^H
//      with tap 1's steering vectors for every l (:24) and w_p = clusters - cluster(p) (:29)
__global__ __launch_bounds__(256) void channel_kernel(Model m, const float2 *gains, const float *u_r,
                                                      const float *u_t, float2 *Hmat)
{
    extern __shared__ float2 sh[];
    float2 *ar = sh, *at = sh + (size_t)m.Nr * m.Np, *cf = at + (size_t)m.Nt * m.Np;
    const int t = blockIdx.y;
    const double beta = 1.0 / (1.0 - exp(-sqrt(2.0) * M_PI / 50.0));
    const double e0 = exp(-sqrt(2.0) / 50.0 * M_PI);
    for (int i = threadIdx.x; i < (m.Nr + m.Nt) * m.Np; i += 256) {
        const bool rx = i < m.Nr * m.Np;
        const int ii = rx ? i : i - m.Nr * m.Np;
        const int dim = rx ? m.Nr : m.Nt;
        const int n = ii % dim, p = ii / dim;
        const double u = rx ? u_r[(size_t)t * m.Np + p] : u_t[(size_t)t * m.Np + p];
        const double phi = beta * (e0 - cosh(u));                      // :56-62
        double s, c;
        sincos(-M_PI * sin(-phi) * (double)n, &s, &c);                 // :42-52
        (rx ? ar : at)[ii] = make_float2((float)c, (float)s);
    }
    const float isq = rsqrtf((float)m.Np);
    for (int i = threadIdx.x; i < m.L * m.Np; i += 256) {
        const int p = i % m.Np;
        const float w = (float)(m.clusters - p / m.rays) * isq;
        const float2 g = gains[(size_t)t * m.L * m.Np + i];
        cf[i] = make_float2(g.x * w, g.y * w);
    }
    __syncthreads();
    const long long n_el = (long long)m.Nr * m.NtL;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n_el; e += (long long)gridDim.x * 256) {
        const int r = (int)(e % m.Nr);
        const int sl = (int)(e / m.Nr);
        const int s = sl % m.Nt, l = sl / m.Nt;
        float hx = 0.f, hy = 0.f;
        for (int p = 0; p < m.Np; ++p) {
            const float2 a = ar[p * m.Nr + r], b = at[p * m.Nt + s], c = cf[l * m.Np + p];
            const float qx = a.x * b.x + a.y * b.y, qy = a.y * b.x - a.x * b.y;     // a * conj(b)
            hx += c.x * qx - c.y * qy;
            hy += c.x * qy + c.y * qx;
        }
        Hmat[(size_t)t * n_el + e] = make_float2(hx, hy);
    }
}

// ---- pilots: Psi[t] (Nt*L x Tp), row (s + Nt*l), column j = toeplitz(s_s)(l, j): s(|j-l|), conjugated below the diagonal
//      sym: the pilot symbols [t][s][Tp]; scale: 1 (4-QAM values) or 1/sqrt(2) (Gaussian draws, ...training.m:20)
__global__ __launch_bounds__(256) void pilots_kernel(Model m, const float2 *sym, float scale, float2 *Psi)
{
    const int t = blockIdx.y;
    const long long n_el = (long long)m.NtL * m.Tp;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n_el; e += (long long)gridDim.x * 256) {
        const int row = (int)(e % m.NtL), j = (int)(e / m.NtL);
        const int s = row % m.Nt, l = row / m.Nt;
        const int d = j - l;
        const float2 v = sym[((size_t)t * m.Nt + s) * m.Tp + (d < 0 ? -d : d)];
        Psi[(size_t)t * n_el + e] = make_float2(v.x * scale, (d < 0 ? -v.y : v.y) * scale);
    }
}

__global__ __launch_bounds__(256) void mask_kernel(long long nm, const float *Omega, const float2 *WR, float2 *subY)
{
    const long long base = (long long)blockIdx.y * nm;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nm; i += (long long)gridDim.x * 256) {
        const float o = Omega[base + i];
        const float2 w = WR[base + i];
        subY[base + i] = make_float2(o * w.x, o * w.y);                 // proposed_hbf.m:42
    }
}

__device__ __forceinline__ double block_sum256(double v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// tau_Y = 1/||Y||_F^2, tau_Z = 1/(2 ||Zbar||_F^2), rho = sqrt(lambda_6(Y'Y) / ||Y||_F^2)   (plot_errorVSsnr.m:127-130)
// lam: the n = min(N, M) non-zero-capable eigenvalues of the Gram; eigs() returns the 6 largest of the M x M matrix.
__global__ __launch_bounds__(256) void hyper_kernel(long long nm, long long nz, int n, int Mcols, const float2 *subY,
                                                    const float2 *Zbar, const float *lam, double *hyp, int rho_max,
                                                    double rho_scale)
{
    __shared__ double sh[4];
    __shared__ float sl[128];
    const int t = blockIdx.x;
    double fy = 0, fz = 0;
    for (long long i = threadIdx.x; i < nm; i += 256) {
        const float2 v = subY[(long long)t * nm + i];
        fy += (double)v.x * v.x + (double)v.y * v.y;
    }
    for (long long i = threadIdx.x; i < nz; i += 256) {
        const float2 v = Zbar[(long long)t * nz + i];
        fz += (double)v.x * v.x + (double)v.y * v.y;
    }
    fy = block_sum256(fy, sh);
    fz = block_sum256(fz, sh);
    for (int i = threadIdx.x; i < n; i += 256) sl[i] = lam[(size_t)t * n + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        // 0-based position in the descending list of Mcols eigenvalues: min(eigs(.)) = the 6th, max(eigs(.)) = the 1st
        const int want = rho_max ? 0 : min(5, Mcols - 1);
        double l6 = 0.0;
        if (want < n) {
            // the (want+1)-th largest: rank by counting (n <= 128)
            for (int i = 0; i < n; ++i) {
                int rank = 0;
                for (int k = 0; k < n; ++k) rank += (sl[k] > sl[i]) || (sl[k] == sl[i] && k < i);
                if (rank == want) l6 = fmax((double)sl[i], 0.0);
            }
        }
        hyp[3 * t + 0] = 1.0 / fy;
        hyp[3 * t + 1] = 0.5 / fz;
        hyp[3 * t + 2] = rho_scale * sqrt(l6 / fy);
    }
}

// 64-bit sort keys: high word = ~bits(|z|^2) (ascending key = descending magnitude), low word = index (stable)
__global__ __launch_bounds__(256) void sortkey_kernel(long long total, long long nz, const float2 *Zbar, uint64_t *keys)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float2 z = Zbar[i];
        const float m2 = z.x * z.x + z.y * z.y;
        keys[i] = ((uint64_t)(~__float_as_uint(m2)) << 32) | (uint64_t)(uint32_t)(i % nz);
    }
}
__global__ __launch_bounds__(256) void sortidx_kernel(long long total, const uint64_t *keys, int32_t *indx)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        indx[i] = (int32_t)(uint32_t)keys[i] + 1;                       // 1-based as MATLAB
}
__global__ void offsets_kernel(int batch, long long nz, int *off)
{
    for (int i = threadIdx.x; i <= batch; i += blockDim.x) off[i] = (int)(i * nz);
}

inline int grid_for(long long n, int cap = 4096) { return (int)std::min<long long>((n + 255) / 256, cap); }

template <class T> T *out_or_tmp(jstsp_ctx *ctx, T *user, size_t n, int memspace)
{
    if (user && memspace == JSTSP_DEVICE) return user;
    return ctx->arena.get<T>(n);
}

}  // namespace

extern "C" int jstsp_build_trials_c32(jstsp_ctx *ctx, const jstsp_model *mp, uint64_t seed, int sweep_idx,
                                      long long trial0, int batch, const jstsp_trials *out, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "build_trials: NULL context");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "build_trials: bad memspace");
    JSTSP_REQUIRE(mp && out, JSTSP_E_NULL, "build_trials: NULL argument");
    JSTSP_ENTER(ctx);
    Model m;
    m.Nt = mp->Nt; m.Nr = mp->Nr; m.L = mp->L; m.Tp = mp->T_prop; m.Mr = mp->Mr; m.Mr_e = mp->Mr_e;
    m.Gr = mp->Gr; m.Gt = mp->Gt; m.clusters = mp->clusters; m.rays = mp->rays;
    JSTSP_REQUIRE(m.Nt > 0 && m.Nr > 0 && m.L > 0 && m.Tp > 0 && m.Gr > 0 && m.Gt > 0 && m.clusters > 0 && m.rays > 0 &&
                      batch > 0 && trial0 >= 0 && sweep_idx >= 0,
                  JSTSP_E_SHAPE, "build_trials: bad model dimensions");
    JSTSP_REQUIRE(m.Mr_e >= 1 && m.Mr_e <= m.Nr && m.Mr >= 1 && m.Mr <= m.Mr_e, JSTSP_E_SHAPE,
                  "build_trials: need 1 <= Mr <= Mr_e <= Nr");
    JSTSP_REQUIRE(m.L <= m.Tp, JSTSP_E_SHAPE, "build_trials: L > T_prop");
    JSTSP_REQUIRE((mp->beamformer == JSTSP_BF_ZC || mp->beamformer == JSTSP_BF_DFT) &&
                      (mp->rho_rule == JSTSP_RHO_MIN6 || mp->rho_rule == JSTSP_RHO_MAX) && mp->rho_scale >= 0.0,
                  JSTSP_E_ARG, "build_trials: bad beamformer / rho_rule / rho_scale");
    JSTSP_REQUIRE(mp->noise_var >= 0.0, JSTSP_E_ARG, "build_trials: negative noise variance");
    JSTSP_REQUIRE(mp->pilots == JSTSP_PILOTS_QAM4 || mp->pilots == JSTSP_PILOTS_GAUSS, JSTSP_E_ARG, "build_trials: bad pilots kind");
    const int gauss = mp->pilots == JSTSP_PILOTS_GAUSS;
    JSTSP_REQUIRE(mp->T_hbf >= 0 && mp->T_hbf <= m.Tp, JSTSP_E_SHAPE, "build_trials: T_hbf outside [0, T_prop]");
    m.Np = m.clusters * m.rays; m.NtL = m.Nt * m.L; m.G2 = m.L * m.Gt;
    const int N = m.Mr_e, M = m.Tp, nG = std::min(N, M), Th = mp->T_hbf;
    const bool want_hyp = out->tau_Y || out->tau_Z || out->rho;
    JSTSP_REQUIRE(!want_hyp || nG <= 128, JSTSP_E_UNSUPPORTED, "build_trials: rho needs min(Mr_e, T_prop) <= 128");
    const size_t lds_ch = ((size_t)(m.Nr + m.Nt) * m.Np + (size_t)m.L * m.Np) * sizeof(float2);
    JSTSP_REQUIRE(lds_ch <= 150 * 1024 && (size_t)m.Mr_e * 4 <= 64 * 1024, JSTSP_E_UNSUPPORTED,
                  "build_trials: steering tables exceed the LDS");
    const bool want_hbf = Th > 0 && (out->Y_hbf || out->A_hbf || out->B_hbf);

    const size_t b = (size_t)batch;
    const size_t nH = (size_t)m.Nr * m.NtL, nPsi = (size_t)m.NtL * m.Tp, nR = (size_t)m.Nr * m.Tp, nY = (size_t)N * M,
                 nB = (size_t)m.G2 * M, nZ = (size_t)m.Gr * m.G2, nA = (size_t)N * m.Gr, nQ = (size_t)m.Nt * m.Tp;
    // ---- workspace -------------------------------------------------------------------------
    JSTSP_REQUIRE(b * nZ < (1ull << 31), JSTSP_E_UNSUPPORTED, "build_trials: batch * Gr * G2 exceeds 2^31");
    size_t sort_tmp = 0;
    if (out->indx_S)
        JSTSP_HIP(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, sort_tmp, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                                   (int)(b * nZ), batch, (const int *)nullptr,
                                                   (const int *)nullptr, 0, 64, ctx->stream));
    size_t need = 0;
    auto acc = [&](size_t bytes) { need += rnd256(bytes); };
    acc(b * m.L * m.Np * 8); acc(b * m.Np * 4); acc(b * m.Np * 4); acc(b * nR * 8); acc(b * nQ); acc(b * nQ * 8);       // draws
    acc((size_t)m.Nr * m.Gr * 8); acc((size_t)m.Nt * m.Gt * 8); acc((size_t)m.Nr * m.Nr * 8);         // Dr, Dt, W
    acc(b * nH * 8); acc(b * nPsi * 8); acc(b * nR * 8); acc(b * nY * 8); acc(b * nY * 8); acc(b * nY * 4);
    acc(nA * 8); acc(b * nB * 8); acc(b * nZ * 8); acc(b * (size_t)m.Gr * m.NtL * 8);
    acc(b * 3 * sizeof(double)); acc(b * nG * 4);
    need += GramWS::bytes(N, M, batch, true);
    if (out->indx_S) { acc(b * nZ * 8); acc(b * nZ * 8); acc(sort_tmp); acc((b + 1) * 8); acc(b * nZ * 4); }
    if (want_hbf) { acc(b * (size_t)m.Nr * Th * 8); acc((size_t)m.Nr * m.Gr * 8); acc(b * (size_t)m.G2 * Th * 8); }
    JSTSP_TRY(ctx->arena.reserve(need + 4096));
    ctx->arena.reset();
    Arena &ar = ctx->arena;
    float2 *gains = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->gains), b * m.L * m.Np, memspace);
    float *u_r = out_or_tmp(ctx, out->u_r, b * m.Np, memspace), *u_t = out_or_tmp(ctx, out->u_t, b * m.Np, memspace);
    float2 *noise = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->noise), b * nR, memspace);
    uint8_t *qam = out_or_tmp(ctx, out->qam_idx, b * nQ, memspace);
    float2 *psym = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->pilot_sym), b * nQ, memspace);
    float2 *Dr = ar.get<float2>((size_t)m.Nr * m.Gr), *Dt = ar.get<float2>((size_t)m.Nt * m.Gt),
           *W = ar.get<float2>((size_t)m.Nr * m.Nr);
    float2 *Hmat = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->H), b * nH, memspace);
    float2 *Psi = ar.get<float2>(b * nPsi), *R = ar.get<float2>(b * nR), *WR = ar.get<float2>(b * nY);
    float2 *subY = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->subY), b * nY, memspace);
    float *Omega = out_or_tmp(ctx, out->Omega, b * nY, memspace);
    float2 *A = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->A), nA, memspace);
    float2 *B = out->B ? out_or_tmp(ctx, reinterpret_cast<float2 *>(out->B), b * nB, memspace) : nullptr;
    float2 *Zbar = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->Zbar), b * nZ, memspace);
    float2 *T1 = ar.get<float2>(b * (size_t)m.Gr * m.NtL);
    double *hyp = ar.get<double>(b * 3);
    float *lam = ar.get<float>(b * nG);
    JSTSP_REQUIRE(gains && u_r && u_t && noise && qam && psym && Dr && Dt && W && Hmat && Psi && R && WR && subY && Omega &&
                      A && (B || !out->B) && Zbar && T1 && hyp && lam,
                  JSTSP_E_NOMEM, "build_trials: workspace exhausted");
    hipStream_t st = ctx->stream;
    const uint64_t sw = (uint64_t)sweep_idx;

    // ---- draws ------------------------------------------------------------------------------
    draw_small_kernel<<<batch, 64, 0, st>>>(m, seed, sw, trial0, gains, u_r, u_t);
    draw_noise_qam_kernel<<<dim3(grid_for((long long)std::max(nR, nQ), 1024), batch), 256, 0, st>>>(m, seed, sw, trial0,
                                                                                                      noise, qam, mp->shared_pilots, gauss, psym);
    omega_kernel<<<dim3(m.Tp, batch), 256, (size_t)m.Mr_e * 4, st>>>(m, seed, sw, trial0, Omega);
    // ---- dictionaries -------------------------------------------------------------------------
    dict_kernel<<<grid_for((long long)m.Nr * m.Gr), 256, 0, st>>>(m.Nr, m.Gr, 0, Dr);
    dict_kernel<<<grid_for((long long)m.Nt * m.Gt), 256, 0, st>>>(m.Nt, m.Gt, 0, Dt);
    // createBeamformer.m: 'ZC' (:15-16, plot_errorVSsnr.m:124) or 'fft' / 'ps' (:5,:12-13 - the same unitary DFT matrix)
    dict_kernel<<<grid_for((long long)m.Nr * m.Nr), 256, 0, st>>>(m.Nr, m.Nr, mp->beamformer == JSTSP_BF_ZC ? 1 : 0, W);
    // ---- channel, pilots ------------------------------------------------------------------------
    channel_kernel<<<dim3(grid_for((long long)nH, 64), batch), 256, lds_ch, st>>>(m, gains, u_r, u_t, Hmat);
    pilots_kernel<<<dim3(grid_for((long long)nPsi, 1024), batch), 256, 0, st>>>(m, psym, gauss ? 0.70710678f : 1.f, Psi);
    JSTSP_HIP(hipGetLastError());
    // R = [H_1..H_L] Psi + sqrt(var/2) noise                                     proposed_hbf.m:19-22
    JSTSP_TRY(gemm(ctx, 'N', 'N', m.Nr, m.Tp, m.NtL, batch, Mat{Hmat, (long long)nH, m.Nr}, Mat{Psi, (long long)nPsi, m.NtL},
                   R, (long long)nR, m.Nr, 1.f, noise, (long long)nR, m.Nr, (float)std::sqrt(mp->noise_var / 2.0)));
    // subY = Omega .* (W_e' R)                                                   proposed_hbf.m:42
    JSTSP_TRY(gemm(ctx, 'C', 'N', N, M, m.Nr, batch, Mat{W, 0, m.Nr}, Mat{R, (long long)nR, m.Nr}, WR, (long long)nY, N));
    mask_kernel<<<dim3(grid_for((long long)nY, 1024), batch), 256, 0, st>>>((long long)nY, Omega, WR, subY);
    // Zbar = [Dr' H_1 Dt ... Dr' H_L Dt]                                         wideband_mmwave_channel.m:35,38
    JSTSP_TRY(gemm(ctx, 'C', 'N', m.Gr, m.NtL, m.Nr, batch, Mat{Dr, 0, m.Nr}, Mat{Hmat, (long long)nH, m.Nr}, T1,
                   (long long)m.Gr * m.NtL, m.Gr));
    JSTSP_TRY(gemm(ctx, 'N', 'N', m.Gr, m.Gt, m.Nt, batch * m.L, Mat{T1, (long long)m.Gr * m.Nt, m.Gr}, Mat{Dt, 0, m.Nt},
                   Zbar, (long long)m.Gr * m.Gt, m.Gr));
    // A = W_e' Dr ; B_l = Dt' Psi_l                                              plot_errorVSsnr.m:132-136
    JSTSP_TRY(gemm(ctx, 'C', 'N', N, m.Gr, m.Nr, 1, Mat{W, 0, m.Nr}, Mat{Dr, 0, m.Nr}, A, 0, N));
    if (B)
        for (int l = 0; l < m.L; ++l)
            JSTSP_TRY(gemm(ctx, 'C', 'N', m.Gt, M, m.Nt, batch, Mat{Dt, 0, m.Nt},
                           Mat{Psi + (size_t)l * m.Nt, (long long)nPsi, m.NtL}, B + (size_t)l * m.Gt, (long long)nB, m.G2));
    // ---- hyper-parameters ---------------------------------------------------------------------------
    if (want_hyp) {
        GramWS w;
        JSTSP_TRY(w.alloc(ar, N, M, batch, true));
        JSTSP_TRY(gram_partials(ctx, w, subY, (long long)nY));
        JSTSP_TRY(launch_eig(ctx, EIG_VECS, w.n, batch, w.Gpart, (long long)w.n * w.n * w.nsplit, w.nsplit,
                             (long long)w.n * w.n, nullptr, nullptr, w.Q, lam, w.Vg));
        hyper_kernel<<<batch, 256, 0, st>>>((long long)nY, (long long)nZ, w.n, M, subY, Zbar, lam, hyp,
                                            mp->rho_rule == JSTSP_RHO_MAX, mp->rho_scale > 0.0 ? mp->rho_scale : 1.0);
        JSTSP_HIP(hipGetLastError());
    }
    // ---- support ordering -------------------------------------------------------------------------------
    int32_t *indx = nullptr;
    if (out->indx_S) {
        uint64_t *k0 = ar.get<uint64_t>(b * nZ), *k1 = ar.get<uint64_t>(b * nZ);
        void *tmp = ar.get<char>(sort_tmp ? sort_tmp : 1);
        int *off = ar.get<int>(b + 1);
        indx = out_or_tmp(ctx, out->indx_S, b * nZ, memspace);
        JSTSP_REQUIRE(k0 && k1 && tmp && off && indx, JSTSP_E_NOMEM, "build_trials: workspace exhausted (sort)");
        offsets_kernel<<<1, 256, 0, st>>>(batch, (long long)nZ, off);
        sortkey_kernel<<<grid_for((long long)(b * nZ)), 256, 0, st>>>((long long)(b * nZ), (long long)nZ, Zbar, k0);
        JSTSP_HIP(hipcub::DeviceSegmentedRadixSort::SortKeys(tmp, sort_tmp, k0, k1, (int)(b * nZ), batch, off, off + 1, 0,
                                                             64, st));
        sortidx_kernel<<<grid_for((long long)(b * nZ)), 256, 0, st>>>((long long)(b * nZ), k1, indx);
        JSTSP_HIP(hipGetLastError());
    }
    // ---- conventional HBF measurement for the LS / VAMP baselines (plot_errorVSsnr.m:73-80, hbf.m:17-24) ----
    float2 *Yh = nullptr, *Ah = nullptr, *Bh = nullptr;
    if (want_hbf) {
        if (out->Y_hbf) {
            Yh = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->Y_hbf), b * (size_t)m.Nr * Th, memspace);
            JSTSP_REQUIRE(Yh, JSTSP_E_NOMEM, "build_trials: workspace exhausted (Y_hbf)");
            JSTSP_TRY(gemm(ctx, 'C', 'N', m.Nr, Th, m.Nr, batch, Mat{W, 0, m.Nr}, Mat{R, (long long)nR, m.Nr}, Yh,
                           (long long)m.Nr * Th, m.Nr));
        }
        if (out->A_hbf) {
            Ah = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->A_hbf), (size_t)m.Nr * m.Gr, memspace);
            JSTSP_REQUIRE(Ah, JSTSP_E_NOMEM, "build_trials: workspace exhausted (A_hbf)");
            JSTSP_TRY(gemm(ctx, 'C', 'N', m.Nr, m.Gr, m.Nr, 1, Mat{W, 0, m.Nr}, Mat{Dr, 0, m.Nr}, Ah, 0, m.Nr));
        }
        if (out->B_hbf) {
            JSTSP_REQUIRE(B, JSTSP_E_ARG, "build_trials: B_hbf requires B");
            Bh = out_or_tmp(ctx, reinterpret_cast<float2 *>(out->B_hbf), b * (size_t)m.G2 * Th, memspace);
            JSTSP_REQUIRE(Bh, JSTSP_E_NOMEM, "build_trials: workspace exhausted (B_hbf)");
            JSTSP_HIP(hipMemcpy2DAsync(Bh, (size_t)m.G2 * Th * 8, B, nB * 8, (size_t)m.G2 * Th * 8, batch,
                                       hipMemcpyDeviceToDevice, st));
        }
    }
    // ---- hand the arrays over ------------------------------------------------------------------------------
    if (memspace == JSTSP_HOST) {
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->gains), gains, b * m.L * m.Np, memspace));
        JSTSP_TRY(stage_out(ctx, out->u_r, u_r, b * m.Np, memspace));
        JSTSP_TRY(stage_out(ctx, out->u_t, u_t, b * m.Np, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->noise), noise, b * nR, memspace));
        JSTSP_TRY(stage_out(ctx, out->qam_idx, qam, b * nQ, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->pilot_sym), psym, b * nQ, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->H), Hmat, b * nH, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->subY), subY, b * nY, memspace));
        JSTSP_TRY(stage_out(ctx, out->Omega, Omega, b * nY, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->A), A, nA, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->B), B, b * nB, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->Zbar), Zbar, b * nZ, memspace));
        JSTSP_TRY(stage_out(ctx, out->indx_S, indx, b * nZ, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->Y_hbf), Yh, b * (size_t)m.Nr * Th, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->A_hbf), Ah, (size_t)m.Nr * m.Gr, memspace));
        JSTSP_TRY(stage_out(ctx, reinterpret_cast<float2 *>(out->B_hbf), Bh, b * (size_t)m.G2 * Th, memspace));
    }
    // tau_Y, tau_Z, rho are host arrays in either memspace: the solver entry points take them from the host
    if (want_hyp) {
        std::vector<double> h(3 * b);
        JSTSP_HIP(hipMemcpyAsync(h.data(), hyp, 3 * b * sizeof(double), hipMemcpyDeviceToHost, st));
        JSTSP_HIP(hipStreamSynchronize(st));
        for (size_t t = 0; t < b; ++t) {
            if (out->tau_Y) out->tau_Y[t] = h[3 * t];
            if (out->tau_Z) out->tau_Z[t] = h[3 * t + 1];
            if (out->rho) out->rho[t] = h[3 * t + 2];
        }
    } else if (memspace == JSTSP_HOST) {
        JSTSP_HIP(hipStreamSynchronize(st));
    }
    return 0;
}
