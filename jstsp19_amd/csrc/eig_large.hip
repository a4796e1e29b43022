// Hermitian eigen-decomposition for orders above 128 - beyond what the LDS-resident Jacobi kernels hold (eig.hip,
// eig2.hip, eig3.hip).  Nothing on the timed paths gets here: these are the one-off decompositions of large factors
// (VAMP's `svd` of vamp.m:32 when G2 = L*Gt > 128, BASELINE configs[4]) and the SVT of inputs whose BOTH dimensions exceed
// 128.  The decomposition itself is rocSOLVER's cheevd (a vendor LAPACK routine, like a library GEMM), loaded lazily
// with dlopen so that the library has no link-time dependency on it and every other entry point works without it; the
// sum of the split-K partials, the symmetrisation and the projector Q = U diag(q) U^H stay here.
#include "common.h"
#include "solver_common.h"

#include <dlfcn.h>
#include <rocsolver/rocsolver.h>

namespace jstsp {
namespace {

struct RocSolver {
    bool tried = false, ok = false;
    char why[256] = {0};
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocsolver_cheevd_strided_batched) cheevd = nullptr;
    rocblas_handle handle[16] = {nullptr};
};

RocSolver &rocsolver()
{
    static RocSolver r;
    if (r.tried) return r;
    r.tried = true;
    // by soname first: a process that already holds a copy (PyTorch-ROCm bundles one) keeps using that one
    void *hs = nullptr;
    for (const char *name : {"librocsolver.so.0", "/opt/rocm/lib/librocsolver.so.0", "librocsolver.so"}) {
        hs = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (hs) break;
    }
    if (!hs) { snprintf(r.why, sizeof(r.why), "dlopen(librocsolver): %s", dlerror()); return r; }
    // rocBLAS is a dependency of rocSOLVER: dlsym on the handle searches the object and what it depends on
    r.create = reinterpret_cast<decltype(r.create)>(dlsym(hs, "rocblas_create_handle"));
    r.set_stream = reinterpret_cast<decltype(r.set_stream)>(dlsym(hs, "rocblas_set_stream"));
    r.cheevd = reinterpret_cast<decltype(r.cheevd)>(dlsym(hs, "rocsolver_cheevd_strided_batched"));
    if (!r.create || !r.set_stream || !r.cheevd) { snprintf(r.why, sizeof(r.why), "rocsolver / rocblas symbols not found"); return r; }
    r.ok = true;
    return r;
}

// W[t] = Hermitian part of sum_s Gpart[t][s]; amax[t] = max |off-diagonal entry| (bits of a non-negative float)
__global__ void sum_sym_kernel(int n, const float2 *Gpart, long long sGt, int nsplit, long long sGs, float2 *W, uint32_t *amax)
{
    const int t = blockIdx.y;
    const float2 *g = Gpart + (long long)t * sGt;
    float2 *w = W + (size_t)t * n * n;
    float m = 0.f;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % n), j = (int)(e / n);
        float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
        for (int s = 0; s < nsplit; ++s) {
            const float2 x = g[(long long)s * sGs + i + (long long)n * j], y = g[(long long)s * sGs + j + (long long)n * i];
            a.x += x.x; a.y += x.y; b.x += y.x; b.y += y.y;
        }
        const float2 v = (i == j) ? make_float2(a.x, 0.f) : make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
        w[e] = v;
        if (i != j) m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(&amax[t], __float_as_uint(m));
}

// An exactly diagonal matrix - the all-zero svt argument of a first ADMM iteration above all (svt.m:7-12 returns zeros for
// it) - comes back from cheevd with the right eigenvalues and NaN eigenvectors (measured: rocSOLVER 3.32 / ROCm 7.2, any
// order): its decomposition is written here instead, eigenvectors I, eigenvalues = the diagonal in place.
__global__ void diagonal_matrix_kernel(int n, const float2 *Gpart, long long sGt, int nsplit, long long sGs, float2 *W, float *D,
                                       const uint32_t *amax)
{
    const int t = blockIdx.y;
    if (amax[t] != 0u) return;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % n), j = (int)(e / n);
        W[(size_t)t * n * n + e] = make_float2(i == j ? 1.f : 0.f, 0.f);
        if (i == j) {
            float d = 0.f;
            for (int s = 0; s < nsplit; ++s) d += Gpart[(long long)t * sGt + (long long)s * sGs + i + (long long)n * i].x;
            D[(size_t)t * n + i] = d;
        }
    }
}

__global__ void last_value_kernel(int n, int batch, const float *D, float *lam_out)     // cheevd returns ascending order
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < batch) lam_out[t] = D[(size_t)t * n + n - 1];
}

// T[t](:, i) = U[t](:, i) * q_i,  q_i = min(1, tau_t / sqrt(max(lambda_i, 0)))  (1 for a zero singular value: svt.m:7-12
// then yields Q = I for the all-zero input, as the small kernels do)
__global__ void scale_cols_kernel(int n, const float2 *U, const float *lam, const TrialParams *prm, const float *tau, float2 *T)
{
    const int t = blockIdx.y;
    const float tv = tau ? tau[t] : prm[t].tauY_rho;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / n);
        const float sig = sqrtf(fmaxf(lam[(size_t)t * n + i], 0.f));
        const float q = (sig > 0.f) ? fminf(1.f, tv / sig) : 1.f;
        const float2 u = U[(size_t)t * n * n + e];
        T[(size_t)t * n * n + e] = make_float2(q * u.x, q * u.y);
    }
}

}  // namespace

int launch_eig_large(jstsp_ctx *ctx, int mode, int n, int batch, const float2 *Gpart, long long sGt, int nsplit,
                     long long sGs, const TrialParams *prm, const float *tau, float2 *Q, float *lam_out)
{
    JSTSP_REQUIRE(mode == EIG_VECS || mode == EIG_SVT_Q || mode == EIG_LMAX, JSTSP_E_ARG, "eig (order %d): bad mode %d", n, mode);
    JSTSP_REQUIRE(n <= 8192, JSTSP_E_UNSUPPORTED, "eig: matrix order %d above 8192", n);
    RocSolver &rs = rocsolver();
    JSTSP_REQUIRE(rs.ok, JSTSP_E_UNSUPPORTED, "eig: order %d > 128 needs rocSOLVER, which could not be loaded (%s)", n, rs.why);
    const int dev = ctx->device;
    JSTSP_REQUIRE(dev >= 0 && dev < 16, JSTSP_E_UNSUPPORTED, "eig: device index %d", dev);
    if (!rs.handle[dev])
        JSTSP_REQUIRE(rs.create(&rs.handle[dev]) == rocblas_status_success, JSTSP_E_UNSUPPORTED, "rocblas_create_handle failed");
    JSTSP_REQUIRE(rs.set_stream(rs.handle[dev], ctx->stream) == rocblas_status_success, JSTSP_E_UNSUPPORTED, "rocblas_set_stream failed");
    hipStream_t st = ctx->stream;
    const size_t nn = (size_t)n * n;
    // stream-ordered temporaries outside the arena (the callers sized that for the small kernels)
    float2 *W = nullptr, *T = nullptr;
    float *D = nullptr, *E = nullptr;
    int *info = nullptr;
    uint32_t *amax = nullptr;
    const bool own_w = (mode != EIG_VECS), own_d = (mode != EIG_VECS);
    if (own_w) JSTSP_HIP(hipMallocAsync((void **)&W, batch * nn * sizeof(float2), st)); else W = Q;
    if (mode == EIG_SVT_Q) JSTSP_HIP(hipMallocAsync((void **)&T, batch * nn * sizeof(float2), st));
    if (own_d) JSTSP_HIP(hipMallocAsync((void **)&D, (size_t)batch * n * sizeof(float), st)); else D = lam_out;
    JSTSP_HIP(hipMallocAsync((void **)&E, (size_t)batch * n * sizeof(float), st));
    JSTSP_HIP(hipMallocAsync((void **)&info, (size_t)batch * sizeof(int), st));
    JSTSP_HIP(hipMallocAsync((void **)&amax, (size_t)batch * sizeof(uint32_t), st));
    JSTSP_HIP(hipMemsetAsync(amax, 0, (size_t)batch * sizeof(uint32_t), st));
    const dim3 grid((unsigned)std::min<size_t>((nn + 255) / 256, 2048), (unsigned)batch);
    hipLaunchKernelGGL(sum_sym_kernel, grid, dim3(256), 0, st, n, Gpart, sGt, nsplit, sGs, W, amax);
    const rocblas_status rc = rs.cheevd(rs.handle[dev], mode == EIG_LMAX ? rocblas_evect_none : rocblas_evect_original,
                                        rocblas_fill_lower, n, reinterpret_cast<rocblas_float_complex *>(W), n, (rocblas_stride)nn,
                                        D, n, E, n, info, batch);
    int rcode = 0;
    if (rc != rocblas_status_success) {
        set_error("rocsolver_cheevd_strided_batched failed with status %d (order %d, %d matrices)", (int)rc, n, batch);
        rcode = JSTSP_E_UNSUPPORTED;
    } else {
        if (mode != EIG_LMAX) hipLaunchKernelGGL(diagonal_matrix_kernel, grid, dim3(256), 0, st, n, Gpart, sGt, nsplit, sGs, W, D, amax);
        if (mode == EIG_LMAX) {
            hipLaunchKernelGGL(last_value_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, n, batch, D, lam_out);
        } else if (mode == EIG_SVT_Q) {
            hipLaunchKernelGGL(scale_cols_kernel, grid, dim3(256), 0, st, n, W, D, prm, tau, T);
            rcode = gemm(ctx, 'N', 'C', n, n, n, batch, Mat{T, (long long)nn, n}, Mat{W, (long long)nn, n}, Q, (long long)nn, n);
        }
    }
    if (own_w) (void)hipFreeAsync(W, st);
    if (T) (void)hipFreeAsync(T, st);
    if (own_d) (void)hipFreeAsync(D, st);
    (void)hipFreeAsync(E, st);
    (void)hipFreeAsync(info, st);
    (void)hipFreeAsync(amax, st);
    JSTSP_TRY(rcode);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace jstsp
