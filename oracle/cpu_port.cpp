// oracle/cpu_port.cpp - TEST INFRASTRUCTURE, never linked into or called by the product.
//
// The timed CPU baseline of bench.py (`cpu_baseline`, kind "port"): a float64 C++ restatement of
//   basic_system_functions/proposed_algorithm.m:1-73 ('approximate')       [paths relative to /root/reference]
//   basic_system_functions/proposed_algorithm_angles.m:1-85                (indx_S != NULL)
//   benchmark_algorithms/svt.m:1-15
// in the structured form of oracle/solvers.py::proposed_algorithm (same operation order; the dense Kronecker operators
// K1, K2, R of :14-25 replaced by the exact identities of SURVEY.md section 0.5, which is the only way the BASELINE sizes
// fit a host), OpenMP over the Monte-Carlo trials - the reference's own parallelism (`parfor r`, plot_errorVSsnr_approx.m:41):
// one trial per thread, every trial single-threaded.  MATLAB cannot run here; this is what BASELINE.md section 3.2 names
// as the host baseline.  It is checked against the numpy oracle in tests/test_cpu_port.py.
//
// Differences from the numpy oracle that do not change results beyond float64 rounding:
//  * svt through the N x N Gram of its argument (eigen-decomposition by cyclic Jacobi) instead of a full SVD whose
//    right factor is M x M (svt.m:5); the guard of svt.m:8-12 fires for the all-zero argument of iteration 1.
//  * spectral norms of :67,:69 as sqrt(lambda_max) of the Grams (Lanczos with full re-orthogonalisation + bisection).
//
// Build: g++ -O3 -march=x86-64-v4 -fopenmp -shared -fPIC (oracle/build_cpu_port.py).  Arrays: column-major,
// complex interleaved double (MATLAB's own layout), trial index slowest.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef double v8d __attribute__((vector_size(64), aligned(8)));

struct Planar {                 // rows x cols, column-major, real and imaginary planes
    int rows = 0, cols = 0;
    std::vector<double> re, im;
    void init(int r, int c) { rows = r; cols = c; re.assign((size_t)r * c, 0.0); im.assign((size_t)r * c, 0.0); }
    void zero() { std::fill(re.begin(), re.end(), 0.0); std::fill(im.begin(), im.end(), 0.0); }
};

void load_interleaved(Planar &p, const double *src, int rows, int cols, bool conj_transpose = false)
{
    if (!conj_transpose) {
        p.init(rows, cols);
        const size_t n = (size_t)rows * cols;
        for (size_t i = 0; i < n; ++i) { p.re[i] = src[2 * i]; p.im[i] = src[2 * i + 1]; }
    } else {                    // p = src^H  (cols x rows)
        p.init(cols, rows);
        for (int c = 0; c < cols; ++c)
            for (int r = 0; r < rows; ++r) {
                const size_t s = (size_t)r + (size_t)rows * c, d = (size_t)c + (size_t)cols * r;
                p.re[d] = src[2 * s]; p.im[d] = -src[2 * s + 1];
            }
    }
}

// C (m x n) = [C +] op(A) (m x k) * B (k x n), all planar column-major; A given as m x k (no transposition here: callers
// keep explicit conjugate-transposed copies of the constant operands).  Register tile 16 rows x 4 columns, k in chunks
// that keep the A panel cache-resident.
void gemm(int m, int n, int k, const double *Ar, const double *Ai, int lda, const double *Br, const double *Bi, int ldb,
          double *Cr, double *Ci, int ldc, bool accumulate)
{
    if (!accumulate)
        for (int j = 0; j < n; ++j) {
            std::memset(Cr + (size_t)ldc * j, 0, sizeof(double) * m);
            std::memset(Ci + (size_t)ldc * j, 0, sizeof(double) * m);
        }
    const int KC = 256;
    const int m16 = m - m % 16, n4 = n - n % 4;
    for (int k0 = 0; k0 < k; k0 += KC) {
        const int kc = std::min(KC, k - k0);
        for (int j0 = 0; j0 < n4; j0 += 4) {
            const double *br[4], *bi[4];
            for (int jj = 0; jj < 4; ++jj) {
                br[jj] = Br + (size_t)ldb * (j0 + jj) + k0;
                bi[jj] = Bi + (size_t)ldb * (j0 + jj) + k0;
            }
            for (int i0 = 0; i0 < m16; i0 += 16) {
                v8d cr[2][4], ci[2][4];
                for (int jj = 0; jj < 4; ++jj)
                    for (int h = 0; h < 2; ++h) {
                        cr[h][jj] = *(const v8d *)(Cr + (size_t)ldc * (j0 + jj) + i0 + 8 * h);
                        ci[h][jj] = *(const v8d *)(Ci + (size_t)ldc * (j0 + jj) + i0 + 8 * h);
                    }
                const double *ar = Ar + (size_t)lda * k0 + i0, *ai = Ai + (size_t)lda * k0 + i0;
                for (int kk = 0; kk < kc; ++kk) {
                    const v8d ar0 = *(const v8d *)(ar + (size_t)lda * kk), ar1 = *(const v8d *)(ar + (size_t)lda * kk + 8);
                    const v8d ai0 = *(const v8d *)(ai + (size_t)lda * kk), ai1 = *(const v8d *)(ai + (size_t)lda * kk + 8);
#pragma GCC unroll 4
                    for (int jj = 0; jj < 4; ++jj) {
                        const double xr = br[jj][kk], xi = bi[jj][kk];
                        cr[0][jj] += ar0 * xr - ai0 * xi; ci[0][jj] += ar0 * xi + ai0 * xr;
                        cr[1][jj] += ar1 * xr - ai1 * xi; ci[1][jj] += ar1 * xi + ai1 * xr;
                    }
                }
                for (int jj = 0; jj < 4; ++jj)
                    for (int h = 0; h < 2; ++h) {
                        *(v8d *)(Cr + (size_t)ldc * (j0 + jj) + i0 + 8 * h) = cr[h][jj];
                        *(v8d *)(Ci + (size_t)ldc * (j0 + jj) + i0 + 8 * h) = ci[h][jj];
                    }
            }
        }
        // edges (rows beyond a multiple of 16, columns beyond a multiple of 4): plain loops
        for (int j = 0; j < n; ++j) {
            const int i_lo = j < n4 ? m16 : 0;
            if (i_lo >= m) continue;
            double *cr = Cr + (size_t)ldc * j, *ci = Ci + (size_t)ldc * j;
            for (int kk = 0; kk < kc; ++kk) {
                const double xr = Br[(size_t)ldb * j + k0 + kk], xi = Bi[(size_t)ldb * j + k0 + kk];
                const double *ar = Ar + (size_t)lda * (k0 + kk), *ai = Ai + (size_t)lda * (k0 + kk);
#pragma omp simd
                for (int i = i_lo; i < m; ++i) {
                    cr[i] += ar[i] * xr - ai[i] * xi;
                    ci[i] += ar[i] * xi + ai[i] * xr;
                }
            }
        }
    }
}

inline void gemm(const Planar &A, const Planar &B, Planar &C, bool accumulate = false)
{
    gemm(A.rows, B.cols, A.cols, A.re.data(), A.im.data(), A.rows, B.re.data(), B.im.data(), B.rows, C.re.data(), C.im.data(),
         C.rows, accumulate);
}

void conj_transpose(const Planar &X, Planar &Xh)
{
    Xh.init(X.cols, X.rows);
    for (int c = 0; c < X.cols; ++c)
        for (int r = 0; r < X.rows; ++r) {
            Xh.re[(size_t)c + (size_t)X.cols * r] = X.re[(size_t)r + (size_t)X.rows * c];
            Xh.im[(size_t)c + (size_t)X.cols * r] = -X.im[(size_t)r + (size_t)X.rows * c];
        }
}

// Hermitian eigen-decomposition G = U diag(lam) U^H by cyclic two-sided Jacobi (G is overwritten by its diagonal form)
void herm_eig_jacobi(int n, Planar &G, Planar &U, std::vector<double> &lam)
{
    U.init(n, n);
    for (int i = 0; i < n; ++i) U.re[(size_t)i + (size_t)n * i] = 1.0;
    double *gr = G.re.data(), *gi = G.im.data(), *ur = U.re.data(), *ui = U.im.data();
    auto at = [n](int r, int c) { return (size_t)r + (size_t)n * c; };
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) {
                const double a = gr[at(r, c)] * gr[at(r, c)] + gi[at(r, c)] * gi[at(r, c)];
                if (r == c) diag += a; else off += a;
            }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double hr = gr[at(p, q)], hi = gi[at(p, q)];
                const double habs = std::hypot(hr, hi);
                if (habs == 0.0) continue;
                const double app = gr[at(p, p)], aqq = gr[at(q, q)];
                // rotation J = [c, s e; -s conj(e), c] with e = h/|h| that zeroes G(p,q)
                const double theta = (aqq - app) / (2.0 * habs);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                const double er = hr / habs, ei = hi / habs;       // e
                // columns p, q:  G(:,p) <- c G(:,p) - s conj(e) G(:,q);  G(:,q) <- s e G(:,p) + c G(:,q)
                for (int r = 0; r < n; ++r) {
                    const double pr = gr[at(r, p)], pi = gi[at(r, p)], qr = gr[at(r, q)], qi = gi[at(r, q)];
                    gr[at(r, p)] = c * pr - s * (er * qr + ei * qi);
                    gi[at(r, p)] = c * pi - s * (er * qi - ei * qr);
                    gr[at(r, q)] = s * (er * pr - ei * pi) + c * qr;
                    gi[at(r, q)] = s * (er * pi + ei * pr) + c * qi;
                }
                // rows p, q:  G(p,:) <- c G(p,:) - s e G(q,:);  G(q,:) <- s conj(e) G(p,:) + c G(q,:)
                for (int k = 0; k < n; ++k) {
                    const double pr = gr[at(p, k)], pi = gi[at(p, k)], qr = gr[at(q, k)], qi = gi[at(q, k)];
                    gr[at(p, k)] = c * pr - s * (er * qr - ei * qi);
                    gi[at(p, k)] = c * pi - s * (er * qi + ei * qr);
                    gr[at(q, k)] = s * (er * pr + ei * pi) + c * qr;
                    gi[at(q, k)] = s * (er * pi - ei * pr) + c * qi;
                }
                gr[at(p, q)] = gi[at(p, q)] = gr[at(q, p)] = gi[at(q, p)] = 0.0;
                gi[at(p, p)] = gi[at(q, q)] = 0.0;
                for (int r = 0; r < n; ++r) {
                    const double pr = ur[at(r, p)], pi = ui[at(r, p)], qr = ur[at(r, q)], qi = ui[at(r, q)];
                    ur[at(r, p)] = c * pr - s * (er * qr + ei * qi);
                    ui[at(r, p)] = c * pi - s * (er * qi - ei * qr);
                    ur[at(r, q)] = s * (er * pr - ei * pi) + c * qr;
                    ui[at(r, q)] = s * (er * pi + ei * pr) + c * qi;
                }
            }
    }
    lam.resize(n);
    for (int i = 0; i < n; ++i) lam[i] = gr[at(i, i)];
}

// largest eigenvalue of a Hermitian positive semi-definite G (n x n): Lanczos with full re-orthogonalisation, then
// bisection on the tridiagonal matrix (Sturm counts)
double herm_lmax(int n, const Planar &G)
{
    std::vector<double> Vr((size_t)n * (n + 1)), Vi((size_t)n * (n + 1)), alpha(n), beta(n + 1, 0.0), wr(n), wi(n);
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) { Vr[i] = 1.0 + 0.37 * std::sin(1.0 + i); Vi[i] = 0.21 * std::cos(2.0 + 3.0 * i); nrm += Vr[i] * Vr[i] + Vi[i] * Vi[i]; }
    nrm = std::sqrt(nrm);
    for (int i = 0; i < n; ++i) { Vr[i] /= nrm; Vi[i] /= nrm; }
    int steps = 0;
    for (int j = 0; j < n; ++j) {
        const double *vr = &Vr[(size_t)n * j], *vi = &Vi[(size_t)n * j];
        std::fill(wr.begin(), wr.end(), 0.0); std::fill(wi.begin(), wi.end(), 0.0);
        for (int c = 0; c < n; ++c) {
            const double xr = vr[c], xi = vi[c];
            const double *gr = &G.re[(size_t)n * c], *gi = &G.im[(size_t)n * c];
            for (int r = 0; r < n; ++r) { wr[r] += gr[r] * xr - gi[r] * xi; wi[r] += gr[r] * xi + gi[r] * xr; }
        }
        double a = 0.0;
        for (int r = 0; r < n; ++r) a += vr[r] * wr[r] + vi[r] * wi[r];
        alpha[j] = a;
        steps = j + 1;
        for (int pass = 0; pass < 2; ++pass)                     // w -= V (V^H w), twice
            for (int q = 0; q <= j; ++q) {
                const double *qr = &Vr[(size_t)n * q], *qi = &Vi[(size_t)n * q];
                double dr = 0.0, di = 0.0;
                for (int r = 0; r < n; ++r) { dr += qr[r] * wr[r] + qi[r] * wi[r]; di += qr[r] * wi[r] - qi[r] * wr[r]; }
                for (int r = 0; r < n; ++r) { wr[r] -= qr[r] * dr - qi[r] * di; wi[r] -= qr[r] * di + qi[r] * dr; }
            }
        double b = 0.0;
        for (int r = 0; r < n; ++r) b += wr[r] * wr[r] + wi[r] * wi[r];
        b = std::sqrt(b);
        beta[j + 1] = b;
        if (j + 1 == n || b <= 1e-14 * std::fabs(alpha[0]) || b == 0.0) break;
        for (int r = 0; r < n; ++r) { Vr[(size_t)n * (j + 1) + r] = wr[r] / b; Vi[(size_t)n * (j + 1) + r] = wi[r] / b; }
    }
    double lo = 0.0, hi = 0.0;
    for (int j = 0; j < steps; ++j) hi = std::max(hi, std::fabs(alpha[j]) + beta[j] + (j + 1 < steps ? beta[j + 1] : 0.0));
    if (hi == 0.0) return 0.0;
    lo = -hi;
    for (int itb = 0; itb < 200 && hi - lo > 1e-16 * std::fabs(hi); ++itb) {
        const double x = 0.5 * (lo + hi);
        int below = 0;                                           // eigenvalues of T smaller than x
        double d = 1.0;
        for (int j = 0; j < steps; ++j) {
            d = alpha[j] - x - (j ? beta[j] * beta[j] / d : 0.0);
            if (d == 0.0) d = 1e-300;
            if (d < 0) ++below;
        }
        if (below == steps) hi = x; else lo = x;
    }
    return 0.5 * (lo + hi);
}

void gram(const Planar &Z, const Planar &Zh, Planar &G)          // G = Z Z^H
{
    G.init(Z.rows, Z.rows);
    gemm(Z, Zh, G);
}

double spectral_norm_sq(const Planar &X, Planar &Xh, Planar &G)   // norm(X)^2 = lambda_max(X X^H)   (MATLAB norm of a matrix)
{
    conj_transpose(X, Xh);
    gram(X, Xh, G);
    return herm_lmax(X.rows, G);
}

// Precision study (tools/precision_study.py): JSTSP_PORT_ROUND is a bit mask of intermediate arrays that are rounded to
// float32 where the HIP path stores or produces them in fp32 - which rounding the NMSE is sensitive to.  Unset / 0 (every
// test, the bench baseline and every parity statement): pure float64, nothing below executes.
inline void round32(Planar &p)
{
    for (size_t i = 0; i < p.re.size(); ++i) { p.re[i] = (double)(float)p.re[i]; p.im[i] = (double)(float)p.im[i]; }
}
// a product whose accumulation carried a relative error `rel` (uniform in [-rel, rel], fixed pseudo-random sequence)
inline void jitter(Planar &p, double rel, uint64_t &state)
{
    for (size_t i = 0; i < p.re.size(); ++i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const double u = ((double)(state >> 11) / 9007199254740992.0) * 2.0 - 1.0;
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const double v = ((double)(state >> 11) / 9007199254740992.0) * 2.0 - 1.0;
        p.re[i] *= 1.0 + rel * u; p.im[i] *= 1.0 + rel * v;
    }
}

// x rounded to `bits` significant bits (the split-f16 operand pack keeps 22)
inline void round_bits(Planar &p, int bits)
{
    auto r = [bits](double x) { if (x == 0.0) return 0.0; int e; const double m = std::frexp(x, &e); return std::ldexp(std::nearbyint(std::ldexp(m, bits)), e - bits); };
    for (size_t i = 0; i < p.re.size(); ++i) { p.re[i] = r(p.re[i]); p.im[i] = r(p.im[i]); }
}

inline double soft(double v, double t) { return v > t ? v - t : (v < -t ? v + t : 0.0); }
inline double ratio(double a, double b) { return b == 0.0 ? (a == 0.0 ? std::numeric_limits<double>::quiet_NaN() : std::numeric_limits<double>::infinity()) : a / b; }

void solve_one(int N, int M, int Gr, int G2, const double *subY_, const double *Omega, const double *A_, const double *B_,
               int Imax, double tau_Y, double tau_S, double rho, const int32_t *indx_S, bool want_ce, double *S_out,
               double *Y_out, double *ce)
{
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2;
    Planar subY, A, Ah, B, Bh, X, V1, V2, C, Xs, Y, Z, Zh, K, S, V, Res, RRes, T1, T2, W, GA, GB, G, U, Q;
    load_interleaved(subY, subY_, N, M);
    load_interleaved(A, A_, N, Gr);
    load_interleaved(Ah, A_, N, Gr, true);
    load_interleaved(B, B_, G2, M);
    load_interleaved(Bh, B_, G2, M, true);
    X.init(N, M); V1.init(N, M); V2.init(N, M); C.init(N, M); Xs.init(N, M); Y.init(N, M); Z.init(N, M); K.init(N, M);
    S.init(Gr, G2); V.init(Gr, G2); Res.init(Gr, G2); RRes.init(Gr, G2); T1.init(N, G2); T2.init(Gr, G2); W.init(N, G2);
    GA.init(Gr, Gr); GB.init(G2, G2);
    const char *rm_env0 = std::getenv("JSTSP_PORT_ROUND");
    const unsigned rm0 = rm_env0 ? (unsigned)std::strtoul(rm_env0, nullptr, 0) : 0u;
    if (rm0 & 131072u) { round_bits(B, 22); round_bits(Bh, 22); }          // the packed dictionary everywhere (consistent)
    gemm(Ah, A, GA);                                              // R = K2'*K2 = (B B')^T (x) (A'A)      (:25)
    gemm(B, Bh, GB);
    if (rm0 & 65536u) { round_bits(B, 22); round_bits(Bh, 22); }           // ... in the products only, G_B from the exact one
    if (rm0 & 262144u) round_bits(GB, 22);                                  // the packed G_B of the apply
    const char *rm_env = std::getenv("JSTSP_PORT_ROUND");
    const unsigned rm = rm_env ? (unsigned)std::strtoul(rm_env, nullptr, 0) : 0u;
    const char *rj_env = std::getenv("JSTSP_PORT_JITTER");
    const double rj = rj_env ? std::atof(rj_env) : 0.0;
    uint64_t jst = 0x9E3779B97F4A7C15ull ^ (uint64_t)(subY_[0] * 1e9);
    if (rm & 16384u) jitter(GA, rj, jst);
    if (rm & 32768u) jitter(GB, rj, jst);
    if (rm & 128u) { round32(GA); round32(GB); }
    if (rm & 67108864u) round32(GB);                                       // G_B alone in fp32 (the device: G_A as two floats, G_B as one)
    std::vector<double> inv_d(nm), omega_s(indx_S ? g : 0, 0.0), lam;
    for (size_t i = 0; i < nm; ++i) inv_d[i] = 1.0 / (Omega[i] + 2.0 * rho);     // iK1                        (:14-20)
    std::fill(ce, ce + (size_t)3 * Imax, 0.0);
    double ir = 1.0 / rho, cc = rho / (rho + 1.0), tY = tau_Y / rho, tS = tau_S / rho;
    if (rm & 1048576u) {        // (precision study) the per-problem scalars as the HIP path holds them: each one rounded to fp32 on its own
        ir = (double)(float)ir; cc = (double)(float)cc; tY = (double)(float)tY; tS = (double)(float)tS;
        rho = (double)(float)rho;
    }
    if (rm & 4194304u)          // (precision study) 1 / (Omega + 2 rho) stored as fp32
        for (size_t i = 0; i < nm; ++i) inv_d[i] = (double)(float)inv_d[i];
    if (rm & 8388608u) { tS = (double)(float)tS; tY = (double)(float)tY; }       // the two thresholds alone
    if (rm & 16777216u) { ir = (double)(float)ir; }                               // 1 / rho alone
    if (rm & 33554432u) { cc = (double)(float)cc; }                               // rho / (rho + 1) alone
    if (rm & 2097152u) {        // (precision study) ... or all derived consistently from the fp32 value of rho, in float64
        rho = (double)(float)rho;
        ir = 1.0 / rho; cc = rho / (rho + 1.0); tY = tau_Y / rho; tS = tau_S / rho;
        for (size_t i = 0; i < nm; ++i) inv_d[i] = 1.0 / (Omega[i] + 2.0 * rho);
    }
    for (int it = 1; it <= Imax; ++it) {
        if (indx_S) {                                             // angles :36 (cumulative support)
            const long long cnt = std::min<long long>(10 + 5ll * it, (long long)g);
            for (long long i = 0; i < cnt; ++i) omega_s[(size_t)indx_S[i] - 1] = 1.0;
        }
        // Y = svt(X - V1/rho, tau_Y/rho)                                                                    (:35, svt.m)
        for (size_t i = 0; i < nm; ++i) { Z.re[i] = X.re[i] - ir * V1.re[i]; Z.im[i] = X.im[i] - ir * V1.im[i]; }
        conj_transpose(Z, Zh);
        gram(Z, Zh, G);
        herm_eig_jacobi(N, G, U, lam);
        bool any_zero = false;
        for (int i = 0; i < N; ++i) any_zero = any_zero || !(lam[i] > 0.0);
        if (any_zero && *std::max_element(lam.begin(), lam.end()) <= 0.0) {
            Y.zero();                                             // 0/0 = NaN in svt.m:7 -> zeros (:12)
        } else {
            // Y = U diag(max(0, 1 - tau/sigma)) U^H Z ; a non-positive Gram eigenvalue of a non-zero Z is a singular value
            // at rounding level: removed (what the numpy oracle computes when its guard does not fire)
            Q.init(N, N);
            Planar Uf; Uf.init(N, N);
            for (int c = 0; c < N; ++c) {
                const double sg = lam[c] > 0.0 ? std::sqrt(lam[c]) : 0.0;
                const double f = sg > tY ? 1.0 - tY / sg : 0.0;
                for (int r = 0; r < N; ++r) { Uf.re[r + (size_t)N * c] = f * U.re[r + (size_t)N * c]; Uf.im[r + (size_t)N * c] = f * U.im[r + (size_t)N * c]; }
            }
            Planar Uh; conj_transpose(U, Uh);
            gemm(Uf, Uh, Q);
            if (rm & 134217728u) {
                // the device's svt operator I - Q as the pass holds it: every entry times 2^13 split into two f16 (11 + 11 significant
                // bits, fused.hip pack_wq_kernel) - a rounding that stays the same for as long as Q does
                auto r11 = [](double x) { if (x == 0.0) return 0.0; int e; const double m = std::frexp(x, &e); return std::ldexp(std::nearbyint(std::ldexp(m, 11)), e - 11); };
                auto sp = [&](double x) { const double h = r11(x * 8192.0), l = r11(x * 8192.0 - h); return (h + l) / 8192.0; };
                for (size_t i = 0; i < Q.re.size(); ++i) { Q.re[i] = sp(Q.re[i]); Q.im[i] = sp(Q.im[i]); }
            }
            gemm(Q, Z, Y);
        }
        if (rm & 256u) round32(Y);
        // X = (V1 + rho Y + subY + V2 + rho C + rho Xs) ./ (Omega + 2 rho);  K = X - V2/rho - C             (:38-43)
        for (size_t i = 0; i < nm; ++i) {
            X.re[i] = (V1.re[i] + rho * Y.re[i] + subY.re[i] + V2.re[i] + rho * C.re[i] + rho * Xs.re[i]) * inv_d[i];
            X.im[i] = (V1.im[i] + rho * Y.im[i] + subY.im[i] + V2.im[i] + rho * C.im[i] + rho * Xs.im[i]) * inv_d[i];
            K.re[i] = X.re[i] - ir * V2.re[i] - C.re[i];
            K.im[i] = X.im[i] - ir * V2.im[i] - C.im[i];
        }
        if (rm & 1u) { round32(X); round32(K); }
        // res = K2'*k - R*v = A^H K B^H - G_A V G_B ; alpha = res'res / res'R res ; v += alpha res        (:47-50)
        gemm(K, Bh, T1);
        if (rm & 2048u) jitter(T1, rj, jst);
        if (rm & 8u) round32(T1);
        gemm(Ah, T1, Res);
        if (rm & 16u) round32(Res);
        gemm(GA, V, T2);
        if (rm & 16u) round32(T2);
        gemm(T2, GB, RRes);
        if (rm & 4096u) jitter(RRes, rj, jst);
        if (rm & 16u) round32(RRes);
        for (size_t i = 0; i < g; ++i) { Res.re[i] -= RRes.re[i]; Res.im[i] -= RRes.im[i]; }
        if (rm & 16u) round32(Res);
        gemm(GA, Res, T2);
        if (rm & 32u) round32(T2);
        gemm(T2, GB, RRes);
        if (rm & 32u) round32(RRes);
        double nr = 0.0, dr = 0.0, di = 0.0, nv = 0.0;
        for (size_t i = 0; i < g; ++i) {
            nr += Res.re[i] * Res.re[i] + Res.im[i] * Res.im[i];
            dr += Res.re[i] * RRes.re[i] + Res.im[i] * RRes.im[i];
            di += Res.re[i] * RRes.im[i] - Res.im[i] * RRes.re[i];
            nv += V.re[i] * V.re[i] + V.im[i] * V.im[i];
        }
        const double den = dr * dr + di * di;
        const double ar = den == 0.0 ? ratio(nr, 0.0) : nr * dr / den, ai = den == 0.0 ? 0.0 : -nr * di / den;   // complex alpha
        double dv = 0.0;
        for (size_t i = 0; i < g; ++i) {
            const double sr = ar * Res.re[i] - ai * Res.im[i], si = ar * Res.im[i] + ai * Res.re[i];
            V.re[i] += sr; V.im[i] += si;
            dv += sr * sr + si * si;
        }
        ce[(size_t)(it - 1) + (size_t)Imax * 2] = ratio(dv, nv);                                            // (:51)
        if (rm & 4u) round32(V);
        // s = soft(re) + j soft(im) (.* Omega_S);  Xs = A S B                                              (:56-58, angles :68)
        for (size_t i = 0; i < g; ++i) {
            const double m = indx_S ? omega_s[i] : 1.0;
            S.re[i] = m * soft(V.re[i], tS); S.im[i] = m * soft(V.im[i], tS);
        }
        if (rm & 4u) round32(S);
        gemm(A, S, W);
        if (rm & 512u) round32(W);
        gemm(W, B, Xs);
        if (rm & 8192u) jitter(Xs, rj, jst);
        if (rm & 64u) round32(Xs);
        // C, V1, V2                                                                                         (:61-65)
        for (size_t i = 0; i < nm; ++i) {
            C.re[i] = cc * (X.re[i] - Xs.re[i] - ir * V2.re[i]);
            C.im[i] = cc * (X.im[i] - Xs.im[i] - ir * V2.im[i]);
            V1.re[i] += rho * (Y.re[i] - X.re[i]); V1.im[i] += rho * (Y.im[i] - X.im[i]);
            V2.re[i] += rho * (C.re[i] - X.re[i] + Xs.re[i]); V2.im[i] += rho * (C.im[i] - X.im[i] + Xs.im[i]);
        }
        if (rm & 2u) { round32(V1); round32(V2); round32(C); }
        if (want_ce) {                                            // norm(V1)^2/norm(X)^2, norm(V2)^2/norm(X)^2     (:67,:69)
            const double nx = spectral_norm_sq(X, Zh, G);
            ce[(size_t)(it - 1)] = ratio(spectral_norm_sq(V1, Zh, G), nx);
            ce[(size_t)(it - 1) + (size_t)Imax] = ratio(spectral_norm_sq(V2, Zh, G), nx);
        }
    }
    for (size_t i = 0; i < g; ++i) { S_out[2 * i] = S.re[i]; S_out[2 * i + 1] = S.im[i]; }
    if (Y_out)
        for (size_t i = 0; i < nm; ++i) { Y_out[2 * i] = Y.re[i]; Y_out[2 * i + 1] = Y.im[i]; }
}

}  // namespace

extern "C" {

// Returns the number of OpenMP threads used (>= 1), or a negative value for a bad argument.
// subY: N x M x batch; Omega: N x M x batch (double); A: N x Gr (shared); B: G2 x M, strideB complex elements between
// trials (0 = shared); indx_S: NULL or Gr*G2 x batch (1-based); S_out: Gr x G2 x batch; Y_out: NULL or N x M x batch;
// ce_out: Imax x 3 x batch.
int jstsp_cpu_proposed_algorithm(int N, int M, int Gr, int G2, int batch, const double *subY, const double *Omega,
                                 const double *A, const double *B, long long strideB, int Imax, const double *tau_Y,
                                 const double *tau_S, const double *rho, const int32_t *indx_S, int want_ce, double *S_out,
                                 double *Y_out, double *ce_out, int threads)
{
    if (N <= 0 || M <= 0 || Gr <= 0 || G2 <= 0 || batch <= 0 || Imax < 0 || !subY || !Omega || !A || !B || !S_out || !ce_out)
        return -1;
    const size_t nm = (size_t)N * M, g = (size_t)Gr * G2;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = std::min(threads > 0 ? threads : omp_get_max_threads(), batch);
#pragma omp parallel for schedule(dynamic, 1) num_threads(used)
#endif
    for (int t = 0; t < batch; ++t)
        solve_one(N, M, Gr, G2, subY + 2 * nm * t, Omega + nm * t, A, B + 2 * (size_t)strideB * t, Imax, tau_Y[t], tau_S[t],
                  rho[t], indx_S ? indx_S + g * t : nullptr, want_ce != 0, S_out + 2 * g * t,
                  Y_out ? Y_out + 2 * nm * t : nullptr, ce_out + (size_t)3 * Imax * t);
    return used;
}

}  // extern "C"
