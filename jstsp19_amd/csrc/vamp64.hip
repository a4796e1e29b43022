// jstsp_vamp_c64 / jstsp_vamp_kron_c64 - benchmark_algorithms/vamp.m:1-55 with MPbased_solvers/VAMP/VampGlmEst.m:347-511 in FLOAT64
// on the device (round 6).
//
// Why a float64 path for this one solver: in the only configuration the reference uses (vamp.m:9,38: nitMax = 100, sigma = 1 at every
// call site, the stopping rule commented out, VampGlmEst.m:505-511) the iteration amplifies a rounding difference by about 1e9 over
// its 100 iterations (tests/test_oracle.py) - the fp32-storage path of vamp.hip follows the float64 recurrences per iteration only
// for the first ~12 and statistically afterwards.  float64 storage and arithmetic leave 1e-16 x 1e9 = 1e-7: the reference's ACTUAL
// output at nit = 100 is reproduced per trial (tests/test_gpu_vamp64.py: rel <= 1e-5 against oracle/vamp.py).  VAMP is element-wise
// work plus small products; its cost is irrelevant, so everything here is plain float64 VALU code - no matrix pipe, no workspace
// arena (stream-ordered allocations), no tuning:
//   * zgemm_kernel        batched complex float64 C = op(A) op(B) [+ D], 16 x 16 LDS tiles
//   * jacobi64_*          two-sided cyclic Jacobi of a batch of Hermitian matrices of any order in global memory: per round (circle
//                         ordering, n / 2 disjoint pairs) one kernel computes the n / 2 rotations, one applies J^H H J to every 2 x 2
//                         block and V J to the basis; sweeps until the off-diagonal norm is below 1e-14 of the whole (at most 30)
//   * the iteration       vamp_kernels.h with C2 = double2 - the kernels of the fp32-storage path, same expressions
// The structure is that of vamp.hip: the LMMSE stage in complex arithmetic on the Kronecker factors, Phi = kron(Gb.', Af) never formed
// (a dense dictionary is the case G2 = 1, Gb = 1); both branches M <= N (:402-406) and M > N (:407-411).
#include "vamp_kernels.h"
#include <algorithm>
#include <vector>

namespace jstsp {
namespace {

struct MatD {
    const double2 *p; long long st; int ld;
};

__device__ __forceinline__ double2 zmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 zconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 zadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }

// C[t] (m x n) = op(A[t]) op(B[t]) + beta D[t]; opX: 0 = as stored, 1 = conjugate transpose; column-major
__global__ __launch_bounds__(256) void zgemm_kernel(int opA, int opB, int m, int n, int k, MatD A, MatD B, double2 *C, long long sC, int ldc,
                                                    const double2 *D, long long sD, int ldd, double beta)
{
    __shared__ double2 sa[16][17], sb[16][17];
    const int t = blockIdx.z, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + tx, j = blockIdx.y * 16 + ty;
    const double2 *a = A.p + (long long)t * A.st, *b = B.p + (long long)t * B.st;
    double2 acc = make_double2(0.0, 0.0);
    for (int k0 = 0; k0 < k; k0 += 16) {
        {   // sa[kk][ii] = op(A)(i0 + ii, k0 + kk): loaded with (tx -> ii, ty -> kk)
            const int ii = blockIdx.x * 16 + tx, kk = k0 + ty;
            double2 v = make_double2(0.0, 0.0);
            if (ii < m && kk < k) v = opA ? zconj(a[kk + (long long)A.ld * ii]) : a[ii + (long long)A.ld * kk];
            sa[ty][tx] = v;
        }
        {   // sb[kk][jj] = op(B)(k0 + kk, j0 + jj): loaded with (tx -> kk, ty -> jj)
            const int kk = k0 + tx, jj = blockIdx.y * 16 + ty;
            double2 v = make_double2(0.0, 0.0);
            if (kk < k && jj < n) v = opB ? zconj(b[jj + (long long)B.ld * kk]) : b[kk + (long long)B.ld * jj];
            sb[tx][ty] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = zadd(acc, zmul(sa[kk][tx], sb[kk][ty]));
        __syncthreads();
    }
    if (i < m && j < n) {
        if (D) {
            const double2 d = D[(long long)t * sD + i + (long long)ldd * j];
            acc.x += beta * d.x; acc.y += beta * d.y;
        }
        C[(long long)t * sC + i + (long long)ldc * j] = acc;
    }
}

int zgemm(hipStream_t st, char opA, char opB, int m, int n, int k, int batch, MatD A, MatD B, double2 *C, long long sC, int ldc,
          const double2 *D = nullptr, long long sD = 0, int ldd = 0, double beta = 0.0)
{
    hipLaunchKernelGGL(zgemm_kernel, dim3((m + 15) / 16, (n + 15) / 16, batch), dim3(256), 0, st, opA == 'C' ? 1 : 0, opB == 'C' ? 1 : 0, m, n, k,
                       A, B, C, sC, ldc, D, sD, ldd, beta);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

// ---- two-sided cyclic Jacobi, float64, any order, batch of matrices in global memory -------------------------------------------
// circle ordering of n = 2 h players: round r pairs slot s with slot n - 1 - s of the ring [0, 1 + (r + 0 ..) ...]
__device__ __forceinline__ void pair_of(int n, int r, int s, int &p, int &q)
{
    // player 0 is fixed, players 1 .. n - 1 rotate by r
    const int nm = n - 1;
    auto at = [&](int slot) { return slot == 0 ? 0 : 1 + (slot - 1 + r) % nm; };
    const int a = at(s), b = at(n - 1 - s);
    p = min(a, b); q = max(a, b);
}

// rotation of pair s of round r: J = [c, s; -conj(s), c] (c real) with J^H [hpp, hpq; conj(hpq), hqq] J diagonal
__global__ __launch_bounds__(256) void jacobi64_rot_kernel(int n, int r, const double2 *H, double2 *rot)
{
    const int t = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n / 2) return;
    int p, q;
    pair_of(n, r, s, p, q);
    const double2 *h = H + (long long)t * n * n;
    const double hpp = h[p + (long long)n * p].x, hqq = h[q + (long long)n * q].x;
    const double2 hpq = h[p + (long long)n * q];
    const double a = hypot(hpq.x, hpq.y);
    double c = 1.0; double2 sn = make_double2(0.0, 0.0);
    if (a > 0.0 && a > 1e-300 * (fabs(hpp) + fabs(hqq))) {
        // real symmetric rotation for [hpp, a; a, hqq] after the phase e = hpq / |hpq| is pulled out
        const double tau = (hqq - hpp) / (2.0 * a);
        const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        c = 1.0 / sqrt(1.0 + tt * tt);
        const double sr = tt * c;
        sn = make_double2(sr * hpq.x / a, sr * hpq.y / a);          // s = sr e
    }
    rot[(long long)t * (n / 2) + s] = make_double2(c, 0.0);
    rot[(long long)(gridDim.y + t) * (n / 2) + s] = sn;
}

// H <- J^H H J on the 2 x 2 block (pair k rows, pair l columns); V <- V J on (row i, pair l).  J_k = [c, s; -conj(s), c] acting on
// the coordinates (p_k, q_k): x_p' = c x_p - conj(s) x_q ... written out below for columns (right factor) and rows (left, conjugated)
__global__ __launch_bounds__(256) void jacobi64_apply_kernel(int n, int r, double2 *H, double2 *V, const double2 *rot, int nb)
{
    const int t = blockIdx.z, h2 = n / 2;
    const int l = blockIdx.x * 16 + (threadIdx.x & 15);            // column pair
    const int k = blockIdx.y * 16 + (threadIdx.x >> 4);            // row pair (H), or row-pair index of V rows (2 k, 2 k + 1)
    if (l >= h2 || k >= h2) return;
    int pl, ql, pk, qk;
    pair_of(n, r, l, pl, ql);
    pair_of(n, r, k, pk, qk);
    const double cl = rot[(long long)t * h2 + l].x, ck = rot[(long long)t * h2 + k].x;
    const double2 sl = rot[(long long)(nb + t) * h2 + l], sk = rot[(long long)(nb + t) * h2 + k];
    double2 *h = H + (long long)t * n * n, *v = V + (long long)t * n * n;
    // columns: [x_p', x_q'] = [x_p, x_q] J_l = [c x_p - conj(s) x_q, s x_p + c x_q]
    auto colrot = [&](double2 xp, double2 xq, double2 &op, double2 &oq) {
        const double2 a = zmul(zconj(sl), xq), b = zmul(sl, xp);
        op = make_double2(cl * xp.x - a.x, cl * xp.y - a.y);
        oq = make_double2(b.x + cl * xq.x, b.y + cl * xq.y);
    };
    double2 a00 = h[pk + (long long)n * pl], a01 = h[pk + (long long)n * ql], a10 = h[qk + (long long)n * pl], a11 = h[qk + (long long)n * ql];
    double2 b00, b01, b10, b11;
    colrot(a00, a01, b00, b01);
    colrot(a10, a11, b10, b11);
    // rows: J_k^H [y_p; y_q] = [c y_p - s y_q; conj(s) y_p + c y_q]
    {
        const double2 u0 = zmul(sk, b10), u1 = zmul(sk, b11), w0 = zmul(zconj(sk), b00), w1 = zmul(zconj(sk), b01);
        a00 = make_double2(ck * b00.x - u0.x, ck * b00.y - u0.y); a01 = make_double2(ck * b01.x - u1.x, ck * b01.y - u1.y);
        a10 = make_double2(w0.x + ck * b10.x, w0.y + ck * b10.y); a11 = make_double2(w1.x + ck * b11.x, w1.y + ck * b11.y);
    }
    if (k == l) { a01 = make_double2(0.0, 0.0); a10 = a01; a00.y = 0.0; a11.y = 0.0; }      // the annihilated pair, exactly
    h[pk + (long long)n * pl] = a00; h[pk + (long long)n * ql] = a01; h[qk + (long long)n * pl] = a10; h[qk + (long long)n * ql] = a11;
    // basis: rows 2 k, 2 k + 1 of V, columns (p_l, q_l)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = 2 * k + u;
        double2 op, oq;
        colrot(v[i + (long long)n * pl], v[i + (long long)n * ql], op, oq);
        v[i + (long long)n * pl] = op; v[i + (long long)n * ql] = oq;
    }
}

// out[t] = (sum of |off-diagonal|^2, sum of |all|^2)
__global__ __launch_bounds__(256) void jacobi64_off_kernel(int n, const double2 *H, double *out)
{
    __shared__ double sh[4];
    const int t = blockIdx.x;
    const double2 *h = H + (long long)t * n * n;
    double off = 0.0, all = 0.0;
    for (long long e = threadIdx.x; e < (long long)n * n; e += 256) {
        const double2 x = h[e];
        const double m2 = x.x * x.x + x.y * x.y;
        all += m2;
        if (e % n != e / n) off += m2;
    }
    off = bsum(off, sh);
    all = bsum(all, sh);
    if (threadIdx.x == 0) { out[2 * t] = off; out[2 * t + 1] = all; }
}

__global__ __launch_bounds__(256) void jacobi64_init_kernel(int n, int n0, const double2 *G, long long sG, double2 *H, double2 *V)
{
    // H = G padded to the even order n with a decoupled diagonal entry (never rotated: its off-diagonals are zero), V = I
    const int t = blockIdx.y;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % n), j = (int)(e / n);
        double2 x = make_double2(0.0, 0.0);
        if (i < n0 && j < n0) x = G[(long long)t * sG + i + (long long)n0 * j];
        H[(long long)t * n * n + e] = x;
        V[(long long)t * n * n + e] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
}

__global__ __launch_bounds__(256) void jacobi64_out_kernel(int n, int n0, const double2 *H, const double2 *V, double2 *U, double *lam)
{
    const int t = blockIdx.y;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)n0 * n0; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % n0), j = (int)(e / n0);
        U[(long long)t * n0 * n0 + e] = V[(long long)t * n * n + i + (long long)n * j];
        if (i == j) lam[(long long)t * n0 + i] = H[(long long)t * n * n + i + (long long)n * i].x;
    }
}

__global__ __launch_bounds__(256) void eig64_order1_kernel(int nmat, const double2 *G, long long sG, double2 *U, double *lam)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < nmat) { U[t] = make_double2(1.0, 0.0); lam[t] = G[(long long)t * sG].x; }
}

struct Scratch {            // stream-ordered temporaries of one call
    hipStream_t st;
    std::vector<void *> held;
    int rc = 0;
    explicit Scratch(hipStream_t s) : st(s) {}
    ~Scratch() { for (void *p : held) (void)hipFreeAsync(p, st); }
    template <class T> T *get(size_t n)
    {
        if (rc) return nullptr;
        void *p = nullptr;
        const hipError_t e = hipMallocAsync(&p, std::max<size_t>(n * sizeof(T), 16), st);
        if (e != hipSuccess) { set_error("vamp (float64): hipMallocAsync(%zu) failed: %s", n * sizeof(T), hipGetErrorString(e)); rc = (int)e; return nullptr; }
        held.push_back(p);
        return static_cast<T *>(p);
    }
};

// Hermitian eigen-decomposition of nmat matrices of order n0 (G: column-major, ld = n0, stride sG): U (n0 x n0 each), lam (n0 each)
int eig64(hipStream_t st, Scratch &sc, int n0, int nmat, const double2 *G, long long sG, double2 *U, double *lam)
{
    if (n0 == 1) {          // (the Kronecker factor of a dense dictionary, Gb = 1: U = 1, lambda = the entry)
        hipLaunchKernelGGL(eig64_order1_kernel, dim3((nmat + 255) / 256), dim3(256), 0, st, nmat, G, sG, U, lam);
        JSTSP_HIP(hipGetLastError());
        return 0;
    }
    const int n = (n0 + 1) & ~1, h2 = n / 2;
    double2 *H = sc.get<double2>((size_t)nmat * n * n), *V = sc.get<double2>((size_t)nmat * n * n), *rot = sc.get<double2>((size_t)2 * nmat * h2);
    double *off = sc.get<double>((size_t)2 * nmat);
    JSTSP_TRY(sc.rc);
    const unsigned gi = (unsigned)std::min<long long>(((long long)n * n + 255) / 256, 4096);
    hipLaunchKernelGGL(jacobi64_init_kernel, dim3(gi, nmat), dim3(256), 0, st, n, n0, G, sG, H, V);
    std::vector<double> hoff(2 * (size_t)nmat);
    for (int sweep = 0; sweep < 30; ++sweep) {
        hipLaunchKernelGGL(jacobi64_off_kernel, dim3(nmat), dim3(256), 0, st, n, H, off);
        JSTSP_HIP(hipMemcpyAsync(hoff.data(), off, hoff.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        JSTSP_HIP(hipStreamSynchronize(st));
        bool done = true;
        for (int t = 0; t < nmat; ++t) done = done && !(hoff[2 * t] > 1e-28 * hoff[2 * t + 1]);      // off-norm <= 1e-14 of the whole (rounding leaves ~ n eps)
        if (done) break;
        for (int r = 0; r < n - 1; ++r) {
            hipLaunchKernelGGL(jacobi64_rot_kernel, dim3((h2 + 255) / 256, nmat), dim3(256), 0, st, n, r, H, rot);
            hipLaunchKernelGGL(jacobi64_apply_kernel, dim3((h2 + 15) / 16, (h2 + 15) / 16, nmat), dim3(256), 0, st, n, r, H, V, rot, nmat);
        }
    }
    hipLaunchKernelGGL(jacobi64_out_kernel, dim3(gi, nmat), dim3(256), 0, st, n, n0, H, V, U, lam);
    JSTSP_HIP(hipGetLastError());
    return 0;
}

inline dim3 gsz(long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096))); }

// the iteration of vamp.hip (vamp_run) on double2 arrays; all arrays in device memory
int vamp_run64(jstsp_ctx *ctx, int Na, int Gr, int G2, int batch, const double2 *Y, const double2 *Af, long long sA, const double2 *Gb,
               long long sG, double sigma, double Lnz, int nit, double damp, double2 *Xout)
{
    hipStream_t st = ctx->stream;
    Scratch a(st);
    const int Nc = Gr * G2, Mc = Na * G2;
    const bool tall = Na > Gr;                     // M > N: VampGlmEst.m:407-411 with V, d from eig(A'A) (:196-218)
    const int Da = tall ? Gr : Na, Dc = tall ? Nc : Mc;
    const int nA = sA ? batch : 1, nG = sG ? batch : 1;
    const size_t bN = (size_t)batch * Nc, bM = (size_t)batch * Mc;
    double2 *r1 = a.get<double2>(bN), *x1 = a.get<double2>(bN), *r2 = a.get<double2>(bN), *x2 = a.get<double2>(bN), *u3 = a.get<double2>(bN);
    double2 *p1 = a.get<double2>(bM), *p2 = a.get<double2>(bM), *z2 = a.get<double2>(bM), *z2o = a.get<double2>(bM), *Ar2 = a.get<double2>(bM),
            *E = a.get<double2>(bM), *T1 = a.get<double2>(std::max(bM, bN)), *T2 = a.get<double2>(std::max(bM, bN)), *tq = a.get<double2>(std::max(bM, bN)),
            *tdq = a.get<double2>(std::max(bM, bN));
    double *q = a.get<double>(std::max(bM, bN)), *dq = a.get<double>(std::max(bM, bN));
    double2 *AAh = a.get<double2>((size_t)nA * Da * Da), *Ua = a.get<double2>((size_t)nA * Da * Da), *Ub = a.get<double2>((size_t)nG * G2 * G2);
    double *lamA = a.get<double>((size_t)nA * Da), *lamB = a.get<double>((size_t)nG * G2);
    VampScal *sc = a.get<VampScal>(batch);
    JSTSP_TRY(a.rc);
    const MatD Am{Af, sA, Na}, Gm{Gb, sG, G2};
    // ---- decompositions (the `svd(B)` of vamp.m:32 in factored complex form): d = eig of the Gram = squares of the singular values
    if (tall) JSTSP_TRY(zgemm(st, 'C', 'N', Gr, Gr, Na, nA, Am, Am, AAh, (long long)Gr * Gr, Gr));          // A'A = Va diag(la) Va'
    else JSTSP_TRY(zgemm(st, 'N', 'C', Na, Na, Gr, nA, Am, Am, AAh, (long long)Na * Na, Na));               // A A' = Ua diag(sa^2) Ua'
    JSTSP_TRY(eig64(st, a, Da, nA, AAh, (long long)Da * Da, Ua, lamA));
    JSTSP_TRY(eig64(st, a, G2, nG, Gb, sG ? sG : (long long)G2 * G2, Ub, lamB));
    const MatD Uam{Ua, sA ? (long long)Da * Da : 0, Da}, Ubm{Ub, sG ? (long long)G2 * G2 : 0, G2};
    JSTSP_HIP(hipMemsetAsync(r1, 0, bN * sizeof(double2), st));           // r1init = eps*1i ~ 0 (vamp.m:45)
    JSTSP_HIP(hipMemsetAsync(p1, 0, bM * sizeof(double2), st));           // VampGlmEst.m:331
    JSTSP_HIP(hipMemsetAsync(x1, 0, bN * sizeof(double2), st));
    JSTSP_HIP(hipMemsetAsync(z2o, 0, bM * sizeof(double2), st));
    hipLaunchKernelGGL((vamp_init_kernel<0>), dim3((batch + 255) / 256), dim3(256), 0, st, batch, sc);
    const long long sN = Nc, sM = Mc;
    for (int it = 0; it < nit; ++it) {
        hipLaunchKernelGGL((vamp_first_half_kernel<double2, double>), dim3(batch), dim3(256), 0, st, Nc, Mc, Dc, Da, G2, it, damp, sigma, Lnz, Y,
                           r1, p1, x1, r2, p2, lamA, sA ? (long long)Da : 0, lamB, sG ? (long long)G2 : 0, q, dq, sc);
        if (tall) {
            // Vr2Ap2 = V'(r2 gam2x/gam2z + A'p2);  x2 = V(Vr2Ap2 .* q);  z2 = A x2                 (:408-410)
            JSTSP_TRY(zgemm(st, 'C', 'N', Gr, G2, Na, batch, Am, MatD{p2, sM, Na}, u3, sN, Gr));
            JSTSP_TRY(zgemm(st, 'N', 'N', Gr, G2, G2, batch, MatD{u3, sN, Gr}, Gm, x2, sN, Gr));
            hipLaunchKernelGGL((vamp_add_ratio_kernel<double2, double>), dim3((unsigned)std::min(64, (Nc + 255) / 256), batch), dim3(256), 0, st, Nc,
                               x2, r2, sc);
            JSTSP_TRY(zgemm(st, 'C', 'N', Gr, G2, Gr, batch, Uam, MatD{x2, sN, Gr}, u3, sN, Gr));
            JSTSP_TRY(zgemm(st, 'N', 'N', Gr, G2, G2, batch, MatD{u3, sN, Gr}, Ubm, x2, sN, Gr));
            hipLaunchKernelGGL((vamp_scale_kernel<double2, double>), gsz((long long)bN), dim3(256), 0, st, (long long)bN, x2, q, dq, u3, tdq);
            JSTSP_TRY(zgemm(st, 'N', 'N', Gr, G2, Gr, batch, Uam, MatD{u3, sN, Gr}, T2, sN, Gr));
            JSTSP_TRY(zgemm(st, 'N', 'C', Gr, G2, G2, batch, MatD{T2, sN, Gr}, Ubm, x2, sN, Gr));
            JSTSP_TRY(zgemm(st, 'N', 'N', Na, G2, Gr, batch, Am, MatD{x2, sN, Gr}, T1, sM, Na));
            JSTSP_TRY(zgemm(st, 'N', 'N', Na, G2, G2, batch, MatD{T1, sM, Na}, Gm, z2, sM, Na));
            hipLaunchKernelGGL((vamp_second_half_kernel<double2, double>), dim3(batch), dim3(256), 0, st, Nc, Mc, it, damp, x2, r2, z2, z2o, p2, r1,
                               p1, sc);
            continue;
        }
        // Ar2 = Af R2 Gb                                                              (:400)
        JSTSP_TRY(zgemm(st, 'N', 'N', Na, G2, Gr, batch, Am, MatD{r2, sN, Gr}, T1, sM, Na));
        JSTSP_TRY(zgemm(st, 'N', 'N', Na, G2, G2, batch, MatD{T1, sM, Na}, Gm, Ar2, sM, Na));
        // t = (U^H (p2 - Ar2)) .* q,  U^H vec(Z) = vec(Ua^H Z Ub)                      (:401)
        hipLaunchKernelGGL((vamp_sub_kernel<double2, double>), gsz((long long)bM), dim3(256), 0, st, (long long)bM, p2, Ar2, E);
        JSTSP_TRY(zgemm(st, 'C', 'N', Na, G2, Na, batch, Uam, MatD{E, sM, Na}, T1, sM, Na));
        JSTSP_TRY(zgemm(st, 'N', 'N', Na, G2, G2, batch, MatD{T1, sM, Na}, Ubm, T2, sM, Na));
        hipLaunchKernelGGL((vamp_scale_kernel<double2, double>), gsz((long long)bM), dim3(256), 0, st, (long long)bM, T2, q, dq, tq, tdq);
        // x2 = r2 + Phi^H U t,  U vec(T) = vec(Ua T Ub^H),  Phi^H vec(Z) = vec(Af^H Z Gb)   (:402)
        JSTSP_TRY(zgemm(st, 'N', 'N', Na, G2, Na, batch, Uam, MatD{tq, sM, Na}, T1, sM, Na));
        JSTSP_TRY(zgemm(st, 'N', 'C', Na, G2, G2, batch, MatD{T1, sM, Na}, Ubm, T2, sM, Na));
        JSTSP_TRY(zgemm(st, 'C', 'N', Gr, G2, Na, batch, Am, MatD{T2, sM, Na}, u3, sN, Gr));
        JSTSP_TRY(zgemm(st, 'N', 'N', Gr, G2, G2, batch, MatD{u3, sN, Gr}, Gm, x2, sN, Gr, r2, sN, Gr, 1.0));
        // z2 = Ar2 + U (d .* t)                                                       (:403)
        JSTSP_TRY(zgemm(st, 'N', 'N', Na, G2, Na, batch, Uam, MatD{tdq, sM, Na}, T1, sM, Na));
        JSTSP_TRY(zgemm(st, 'N', 'C', Na, G2, G2, batch, MatD{T1, sM, Na}, Ubm, z2, sM, Na, Ar2, sM, Na, 1.0));
        hipLaunchKernelGGL((vamp_second_half_kernel<double2, double>), dim3(batch), dim3(256), 0, st, Nc, Mc, it, damp, x2, r2, z2, z2o, p2, r1, p1,
                           sc);
    }
    JSTSP_HIP(hipGetLastError());
    JSTSP_HIP(hipMemcpyAsync(Xout, x1, bN * sizeof(double2), hipMemcpyDeviceToDevice, st));   // x = x1 (vamp.m:54)
    return 0;
}

__global__ __launch_bounds__(256) void sca_estim_kernel(long long n, const double *rhat, double rvar, double var0, double p1, double *xhat, double *xvar)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double xh, xv;
        bg_denoise(rhat[i], rvar, var0, p1, xh, xv);
        xhat[i] = xh; xvar[i] = xv;
    }
}
__global__ __launch_bounds__(256) void awgn_out_kernel(long long n, const double *y, const double *phat, double pvar, double wvar, double *zhat)
{
    const double gain = pvar / (pvar + wvar);                                    // CAwgnEstimOut.m:106
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        zhat[i] = gain * (y[i] - phat[i]) + phat[i];                             // :107
}

// a caller array in device memory: itself (JSTSP_DEVICE) or a stream-ordered copy (JSTSP_HOST)
const double2 *stage64(Scratch &sc, const jstsp_c64 *src, size_t n, int memspace)
{
    if (memspace == JSTSP_DEVICE) return reinterpret_cast<const double2 *>(src);
    double2 *d = sc.get<double2>(n);
    if (!d) return nullptr;
    const hipError_t e = hipMemcpyAsync(d, src, n * sizeof(double2), hipMemcpyHostToDevice, sc.st);
    if (e != hipSuccess) { set_error("vamp (float64): upload failed: %s", hipGetErrorString(e)); sc.rc = (int)e; return nullptr; }
    return d;
}

}  // namespace
}  // namespace jstsp

using namespace jstsp;

extern "C" {

int jstsp_vamp_kron_c64(jstsp_ctx *ctx, int Na, int Gr, int G2, int batch, const jstsp_c64 *Y_, const jstsp_c64 *Af_, long long strideA,
                        const jstsp_c64 *Gb_, long long strideG, double sigma, double Lnz, int nit, jstsp_c64 *X_out, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(Y_ && Af_ && Gb_ && X_out, JSTSP_E_NULL, "vamp_kron: NULL array argument");
    JSTSP_REQUIRE(Na > 0 && Gr > 0 && G2 > 0 && batch > 0 && nit >= 1 && strideA >= 0 && strideG >= 0, JSTSP_E_SHAPE, "vamp_kron: bad shape");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_REQUIRE(std::min(Na, Gr) <= 2048 && G2 <= 8192, JSTSP_E_UNSUPPORTED,
                  "vamp_kron: min(Na, Gr) = %d, G2 = %d: the factor eigenproblems are limited to orders 2048 and 8192", std::min(Na, Gr), G2);
    JSTSP_REQUIRE(sigma > 0 && Lnz > 0 && Lnz < 2.0 * Gr * G2, JSTSP_E_ARG, "vamp: need sigma > 0 and 0 < L < nx");
    JSTSP_ENTER(ctx);
    const size_t szA = strideA ? (size_t)strideA * (batch - 1) + (size_t)Na * Gr : (size_t)Na * Gr;
    const size_t szG = strideG ? (size_t)strideG * (batch - 1) + (size_t)G2 * G2 : (size_t)G2 * G2;
    const size_t bN = (size_t)batch * Gr * G2, bM = (size_t)batch * Na * G2;
    Scratch sc(ctx->stream);
    const double2 *Y = stage64(sc, Y_, bM, memspace), *Af = stage64(sc, Af_, szA, memspace), *Gb = stage64(sc, Gb_, szG, memspace);
    double2 *X = memspace == JSTSP_DEVICE ? reinterpret_cast<double2 *>(X_out) : sc.get<double2>(bN);
    JSTSP_TRY(sc.rc);
    JSTSP_TRY(vamp_run64(ctx, Na, Gr, G2, batch, Y, Af, strideA, Gb, strideG, sigma, Lnz, nit, 0.85, X));      // damp: vamp.m:11
    if (memspace == JSTSP_HOST) {
        JSTSP_HIP(hipMemcpyAsync(X_out, X, bN * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
        JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

// SparseScaEstim(CAwgnEstimIn(0, var0), p1).estim(rhat, rvar) and CAwgnEstimOut(y, wvar).estim(phat, pvar) on arrays: the device
// functions of the VAMP iteration, stand-alone (include/jstsp.h)
int jstsp_sparse_sca_estim_f64(jstsp_ctx *ctx, long long n, const double *rhat_, double rvar, double var0, double p1, double *xhat_, double *xvar_,
                               int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(rhat_ && xhat_ && xvar_, JSTSP_E_NULL, "sparse_sca_estim: NULL array argument");
    JSTSP_REQUIRE(n > 0, JSTSP_E_SHAPE, "sparse_sca_estim: n = %lld", n);
    JSTSP_REQUIRE(rvar >= 0 && var0 > 0 && p1 > 0 && p1 < 1, JSTSP_E_ARG, "sparse_sca_estim: need rvar >= 0, var0 > 0, 0 < p1 < 1");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_ENTER(ctx);
    Scratch sc(ctx->stream);
    const double *rhat = rhat_;
    double *xhat = xhat_, *xvar = xvar_;
    if (memspace == JSTSP_HOST) {
        double *d = sc.get<double>((size_t)3 * n);
        JSTSP_TRY(sc.rc);
        JSTSP_HIP(hipMemcpyAsync(d, rhat_, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        rhat = d; xhat = d + n; xvar = d + 2 * n;
    }
    hipLaunchKernelGGL(sca_estim_kernel, gsz(n), dim3(256), 0, ctx->stream, n, rhat, rvar, var0, p1, xhat, xvar);
    JSTSP_HIP(hipGetLastError());
    if (memspace == JSTSP_HOST) {
        JSTSP_HIP(hipMemcpyAsync(xhat_, xhat, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        JSTSP_HIP(hipMemcpyAsync(xvar_, xvar, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int jstsp_cawgn_estim_out_f64(jstsp_ctx *ctx, long long n, const double *y_, const double *phat_, double pvar, double wvar, double *zhat_,
                              double *zvar, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(y_ && phat_ && zhat_ && zvar, JSTSP_E_NULL, "cawgn_estim_out: NULL argument");
    JSTSP_REQUIRE(n > 0, JSTSP_E_SHAPE, "cawgn_estim_out: n = %lld", n);
    JSTSP_REQUIRE(pvar >= 0 && wvar > 0, JSTSP_E_ARG, "cawgn_estim_out: need pvar >= 0, wvar > 0");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_ENTER(ctx);
    Scratch sc(ctx->stream);
    const double *y = y_, *phat = phat_;
    double *zhat = zhat_;
    if (memspace == JSTSP_HOST) {
        double *d = sc.get<double>((size_t)3 * n);
        JSTSP_TRY(sc.rc);
        JSTSP_HIP(hipMemcpyAsync(d, y_, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        JSTSP_HIP(hipMemcpyAsync(d + n, phat_, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        y = d; phat = d + n; zhat = d + 2 * n;
    }
    hipLaunchKernelGGL(awgn_out_kernel, gsz(n), dim3(256), 0, ctx->stream, n, y, phat, pvar, wvar, zhat);
    JSTSP_HIP(hipGetLastError());
    *zvar = wvar * (pvar / (pvar + wvar));                                       // CAwgnEstimOut.m:108 (scale = 1)
    if (memspace == JSTSP_HOST) {
        JSTSP_HIP(hipMemcpyAsync(zhat_, zhat, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        JSTSP_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

// x = vamp(y, A, sigma, L) with a dense dictionary A (M x N): the Kronecker form with G2 = 1, Gb = 1 (vamp.m:1)
int jstsp_vamp_c64(jstsp_ctx *ctx, int M, int N, int batch, const jstsp_c64 *y, const jstsp_c64 *A, long long strideA, double sigma, double Lnz,
                   int nit, jstsp_c64 *x_out, int memspace)
{
    JSTSP_REQUIRE(ctx, JSTSP_E_NULL, "ctx is NULL");
    JSTSP_REQUIRE(memspace == JSTSP_HOST || memspace == JSTSP_DEVICE, JSTSP_E_ARG, "bad memspace %d", memspace);
    JSTSP_ENTER(ctx);
    static const jstsp_c64 one_h = {1.0, 0.0};
    if (memspace == JSTSP_HOST) return jstsp_vamp_kron_c64(ctx, M, N, 1, batch, y, A, strideA, &one_h, 0, sigma, Lnz, nit, x_out, memspace);
    if (!ctx->unit64) {     // device arrays: a device copy of the 1 x 1 identity factor, owned by the context
        JSTSP_HIP(hipMalloc((void **)&ctx->unit64, sizeof(double2)));
        JSTSP_HIP(hipMemcpy(ctx->unit64, &one_h, sizeof(double2), hipMemcpyHostToDevice));
    }
    return jstsp_vamp_kron_c64(ctx, M, N, 1, batch, y, A, strideA, reinterpret_cast<const jstsp_c64 *>(ctx->unit64), 0, sigma, Lnz, nit, x_out,
                               memspace);
}

}  // extern "C"
