// MEX gateway of libjstsp_mi355x.so — one entry point, dispatched by name, used by the .m wrappers
// in this directory (proposed_algorithm.m, OMP.m, ...), which carry the reference's exact
// signatures so that the plot_errorVS* drivers run unchanged.
//
//   mex -R2018a -I../include jstsp_mex.cpp -L../jstsp19_amd/csrc -ljstsp_mi355x
//
// Cannot be built in the build image (no MATLAB / mex.h): everything is guarded on
// MATLAB_MEX_FILE.  Conventions follow the only in-tree MEX exemplar of the reference,
// MPbased_solvers/BiGAMP/comparison_codes/ompbox10/private/ompmex.c:39-133 (nrhs/nlhs checks,
// mexErrMsgIdAndTxt, outputs via mxCreate*), with the interleaved-complex API (-R2018a).
#ifdef MATLAB_MEX_FILE
#include <cstring>
#include <string>
#include <vector>
#include "mex.h"
#include "jstsp.h"

static jstsp_ctx *g_ctx = nullptr;
static void at_exit() { if (g_ctx) { jstsp_destroy(g_ctx); g_ctx = nullptr; } }

static void fail(const char *what, int rc)
{
    // copy the message first: mexErrMsgIdAndTxt long-jumps, no C++ object may be live here
    static char msg[640];
    snprintf(msg, sizeof(msg), "%s failed (%d): %s", what, rc, jstsp_last_error());
    mexErrMsgIdAndTxt("jstsp:call", "%s", msg);
}

// double complex (interleaved) mxArray -> float2 vector (MATLAB arrays are column-major already)
static void to_c32(const mxArray *a, std::vector<jstsp_c32> &out)
{
    const size_t n = mxGetNumberOfElements(a);
    out.resize(n);
    if (mxIsComplex(a)) {
        const mxComplexDouble *p = mxGetComplexDoubles(a);
        for (size_t i = 0; i < n; ++i) { out[i].re = (float)p[i].real; out[i].im = (float)p[i].imag; }
    } else {
        const double *p = mxGetDoubles(a);
        for (size_t i = 0; i < n; ++i) { out[i].re = (float)p[i]; out[i].im = 0.f; }
    }
}
static mxArray *from_c32(const std::vector<jstsp_c32> &v, mwSize r, mwSize c)
{
    mxArray *a = mxCreateDoubleMatrix(r, c, mxCOMPLEX);
    mxComplexDouble *p = mxGetComplexDoubles(a);
    for (size_t i = 0; i < v.size(); ++i) { p[i].real = v[i].re; p[i].imag = v[i].im; }
    return a;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("jstsp:args", "first argument must be the function name");
    char name[64];
    mxGetString(prhs[0], name, sizeof(name));
    if (!g_ctx) {
        // one context per MATLAB process; parfor workers are separate processes: spread them over the GPUs
        int dev = 0;
        const char *env = getenv("JSTSP_DEVICE");
        if (env) dev = atoi(env);
        int rc = jstsp_create(dev, &g_ctx);
        if (rc) fail("jstsp_create", rc);
        mexAtExit(at_exit);
    }
    const std::string fn(name);
    if (fn == "proposed_algorithm") {
        // (subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type [, indx_S])  ->  [S, Y, convergence_error]
        if (nrhs < 10) mexErrMsgIdAndTxt("jstsp:args", "proposed_algorithm: 9 inputs expected");
        const int N = (int)mxGetM(prhs[1]), M = (int)mxGetN(prhs[1]);
        const int Gr = (int)mxGetN(prhs[3]), G2 = (int)mxGetM(prhs[4]);
        if ((int)mxGetM(prhs[3]) != N || (int)mxGetN(prhs[4]) != M || (int)mxGetM(prhs[2]) != N || (int)mxGetN(prhs[2]) != M)
            mexErrMsgIdAndTxt("jstsp:shape", "proposed_algorithm: inconsistent dimensions");
        std::vector<jstsp_c32> subY, A, B;
        to_c32(prhs[1], subY); to_c32(prhs[3], A); to_c32(prhs[4], B);
        std::vector<float> Om((size_t)N * M);
        const double *po = mxGetDoubles(prhs[2]);
        for (size_t i = 0; i < Om.size(); ++i) Om[i] = (float)po[i];
        const int Imax = (int)mxGetScalar(prhs[5]);
        const double tY = mxGetScalar(prhs[6]), tS = mxGetScalar(prhs[7]), rho = mxGetScalar(prhs[8]);
        char type[32] = "";
        mxGetString(prhs[9], type, sizeof(type));
        const int tcode = strcmp(type, "approximate") == 0 ? JSTSP_TYPE_APPROXIMATE : JSTSP_TYPE_STD;
        std::vector<int32_t> idx;
        if (nrhs >= 11 && !mxIsEmpty(prhs[10])) {            // proposed_algorithm_angles: 1-based linear indices
            const double *pi = mxGetDoubles(prhs[10]);
            idx.resize(mxGetNumberOfElements(prhs[10]));
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int32_t)pi[i];
        }
        std::vector<jstsp_c32> S((size_t)Gr * G2), Y((size_t)N * M);
        std::vector<double> ce(nlhs >= 3 ? (size_t)Imax * 3 : 0);   // spectral norms only if requested
        int rc = jstsp_proposed_algorithm_c32(g_ctx, N, M, Gr, G2, 1, subY.data(), Om.data(), A.data(), 0, B.data(), 0,
                                              Imax, &tY, &tS, &rho, tcode, idx.empty() ? nullptr : idx.data(), S.data(),
                                              nlhs >= 2 ? Y.data() : nullptr, nlhs >= 3 ? ce.data() : nullptr, JSTSP_HOST);
        if (rc) { std::vector<jstsp_c32>().swap(S); fail("jstsp_proposed_algorithm_c32", rc); }
        plhs[0] = from_c32(S, Gr, G2);
        if (nlhs >= 2) plhs[1] = from_c32(Y, N, M);
        if (nlhs >= 3) {
            plhs[2] = mxCreateDoubleMatrix(Imax, 3, mxREAL);
            memcpy(mxGetDoubles(plhs[2]), ce.data(), ce.size() * sizeof(double));
        }
    } else if (fn == "svt") {
        const int Mr = (int)mxGetM(prhs[1]), Mt = (int)mxGetN(prhs[1]);
        std::vector<jstsp_c32> Y, X((size_t)Mr * Mt);
        to_c32(prhs[1], Y);
        const double tau = mxGetScalar(prhs[2]);
        int rc = jstsp_svt_c32(g_ctx, Mr, Mt, 1, Y.data(), &tau, X.data(), JSTSP_HOST);
        if (rc) fail("jstsp_svt_c32", rc);
        plhs[0] = from_c32(X, Mr, Mt);
    } else if (fn == "ls") {
        // (Y, A, B) -> pinv(A)*Y*pinv(B)   (plot_errorVSsnr.m:83)
        const int N = (int)mxGetM(prhs[1]), M = (int)mxGetN(prhs[1]);
        const int Gr = (int)mxGetN(prhs[2]), G2 = (int)mxGetM(prhs[3]);
        if ((int)mxGetM(prhs[2]) != N || (int)mxGetN(prhs[3]) != M)
            mexErrMsgIdAndTxt("jstsp:shape", "ls: inconsistent dimensions");
        std::vector<jstsp_c32> Y, A, B, S((size_t)Gr * G2);
        to_c32(prhs[1], Y); to_c32(prhs[2], A); to_c32(prhs[3], B);
        int rc = jstsp_ls_c32(g_ctx, N, M, Gr, G2, 1, Y.data(), A.data(), 0, B.data(), 0, S.data(), JSTSP_HOST);
        if (rc) fail("jstsp_ls_c32", rc);
        plhs[0] = from_c32(S, Gr, G2);
    } else if (fn == "OMP") {
        // (A, v, m, snr) -> [x_hat, indexSet (1 x m cell), v, targetMatrix]
        const int meas = (int)mxGetM(prhs[1]), size_d = (int)mxGetN(prhs[1]);
        const int m = (int)mxGetScalar(prhs[3]);
        std::vector<jstsp_c32> A, v, x((size_t)size_d), T((size_t)meas * m);
        to_c32(prhs[1], A); to_c32(prhs[2], v);
        std::vector<int32_t> idx(m);
        int rc = jstsp_omp_c32(g_ctx, meas, size_d, 1, A.data(), 0, v.data(), m, x.data(), idx.data(),
                               nlhs >= 4 ? T.data() : nullptr, JSTSP_HOST);
        if (rc) fail("jstsp_omp_c32", rc);
        plhs[0] = from_c32(x, size_d, 1);
        if (nlhs >= 2) {
            plhs[1] = mxCreateCellMatrix(1, m);
            for (int i = 0; i < m; ++i) mxSetCell(plhs[1], i, mxCreateDoubleScalar(idx[i]));
        }
        if (nlhs >= 3) plhs[2] = mxDuplicateArray(prhs[2]);
        if (nlhs >= 4) plhs[3] = from_c32(T, meas, m);
    } else if (fn == "sparse_admm") {
        // (Htrue, OH, Dr, Dt, Imax) -> [S, convergence_error]
        const int Mr = (int)mxGetM(prhs[2]), Mt = (int)mxGetN(prhs[2]);
        const int Gr = (int)mxGetN(prhs[3]), Gt = (int)mxGetN(prhs[4]);
        const int Imax = (int)mxGetScalar(prhs[5]);
        std::vector<jstsp_c32> H, OH, Dr, Dt, S((size_t)Mr * Mt);
        to_c32(prhs[1], H); to_c32(prhs[2], OH); to_c32(prhs[3], Dr); to_c32(prhs[4], Dt);
        std::vector<double> ce(Imax);
        int rc = jstsp_sparse_admm_c32(g_ctx, Mr, Mt, Gr, Gt, 1, H.data(), OH.data(), Dr.data(), Dt.data(), Imax, S.data(),
                                       ce.data(), JSTSP_HOST);
        if (rc) fail("jstsp_sparse_admm_c32", rc);
        plhs[0] = from_c32(S, Mr, Mt);
        if (nlhs >= 2) {
            plhs[1] = mxCreateDoubleMatrix(Imax, 1, mxREAL);
            memcpy(mxGetDoubles(plhs[1]), ce.data(), ce.size() * sizeof(double));
        }
    } else if (fn == "mc_svt" || fn == "mc_admm") {
        const bool admm = fn == "mc_admm";
        const int o = admm ? 1 : 0;                         // mc_admm has Htrue first
        const int Mr = (int)mxGetM(prhs[1 + o]), Mt = (int)mxGetN(prhs[1 + o]);
        std::vector<jstsp_c32> H, OH, X((size_t)Mr * Mt);
        if (admm) to_c32(prhs[1], H);
        to_c32(prhs[1 + o], OH);
        std::vector<float> Om((size_t)Mr * Mt);
        const double *po = mxGetDoubles(prhs[2 + o]);
        for (size_t i = 0; i < Om.size(); ++i) Om[i] = (float)po[i];
        const int Imax = (int)mxGetScalar(prhs[3 + o]);
        const double tau = mxGetScalar(prhs[4 + o]), rho = mxGetScalar(prhs[5 + o]);
        std::vector<double> ce(Imax);
        int rc = admm ? jstsp_mc_admm_c32(g_ctx, Mr, Mt, 1, H.data(), OH.data(), Om.data(), Imax, &tau, &rho, X.data(),
                                          ce.data(), JSTSP_HOST)
                      : jstsp_mc_svt_c32(g_ctx, Mr, Mt, 1, OH.data(), Om.data(), Imax, &tau, &rho, X.data(), JSTSP_HOST);
        if (rc) fail(admm ? "jstsp_mc_admm_c32" : "jstsp_mc_svt_c32", rc);
        plhs[0] = from_c32(X, Mr, Mt);
        if (admm && nlhs >= 2) {
            plhs[1] = mxCreateDoubleMatrix(Imax, 1, mxREAL);
            memcpy(mxGetDoubles(plhs[1]), ce.data(), ce.size() * sizeof(double));
        }
    } else if (fn == "vamp") {
        // (y, A, sigma, L) -> x      benchmark_algorithms/vamp.m:1 (always 100 iterations, VampGlmEst.m:509-511)
        const int M = (int)mxGetM(prhs[2]), N = (int)mxGetN(prhs[2]);
        std::vector<jstsp_c32> y, A, x((size_t)N);
        to_c32(prhs[1], y); to_c32(prhs[2], A);
        int rc = jstsp_vamp_c32(g_ctx, M, N, 1, y.data(), A.data(), 0, mxGetScalar(prhs[3]), mxGetScalar(prhs[4]), 100,
                                x.data(), JSTSP_HOST);
        if (rc) fail("jstsp_vamp_c32", rc);
        plhs[0] = from_c32(x, N, 1);
    } else if (fn == "mmv_omp") {
        // (A, K, Y) -> [Z, support]   replaces spx.pursuit.joint.OrthogonalMatchingPursuit(A, K).solve(Y).Z
        // (plot_errorVSsnr.m:116-117; sparse-plex is not vendored: published simultaneous OMP, l2 row score)
        const int N = (int)mxGetM(prhs[1]), Gr = (int)mxGetN(prhs[1]);
        const int K = (int)mxGetScalar(prhs[2]);
        const int S = (int)mxGetN(prhs[3]);
        if ((int)mxGetM(prhs[3]) != N) mexErrMsgIdAndTxt("jstsp:shape", "mmv_omp: size(Y,1) must equal size(A,1)");
        std::vector<jstsp_c32> A, Y, Z((size_t)Gr * S);
        to_c32(prhs[1], A); to_c32(prhs[3], Y);
        std::vector<int32_t> idx(K), cnt(1);
        int rc = jstsp_mmv_omp_c32(g_ctx, N, Gr, S, 1, A.data(), 0, Y.data(), K, 2, Z.data(), idx.data(), cnt.data(), JSTSP_HOST);
        if (rc) fail("jstsp_mmv_omp_c32", rc);
        plhs[0] = from_c32(Z, Gr, S);
        if (nlhs >= 2) {
            plhs[1] = mxCreateDoubleMatrix(1, cnt[0], mxREAL);
            for (int i = 0; i < cnt[0]; ++i) mxGetDoubles(plhs[1])[i] = idx[i];
        }
    } else if (fn == "pinv") {
        // (A) -> pinv(A) through the float64 SVD kernel (small matrices; MATLAB's own pinv works as well)
        const int R = (int)mxGetM(prhs[1]), C = (int)mxGetN(prhs[1]);
        std::vector<jstsp_c32> A, P((size_t)R * C);
        to_c32(prhs[1], A);
        int rc = jstsp_pinv_c32(g_ctx, R, C, 1, A.data(), P.data(), JSTSP_HOST);
        if (rc) fail("jstsp_pinv_c32", rc);
        plhs[0] = from_c32(P, C, R);
    } else if (fn == "rate") {
        // (S, Zbar, noise_var) -> log2(real(det(eye(Nr) + 1/Nr*Zbar*Zbar'/(noise_var + nmse))))   plot_rateVSframelength.m:81
        const int R = (int)mxGetM(prhs[1]), C = (int)mxGetN(prhs[1]);
        std::vector<jstsp_c32> S, Zb;
        to_c32(prhs[1], S); to_c32(prhs[2], Zb);
        double r = 0.0;
        int rc = jstsp_rate_c32(g_ctx, R, C, 1, S.data(), Zb.data(), mxGetScalar(prhs[3]), &r, JSTSP_HOST);
        if (rc) fail("jstsp_rate_c32", rc);
        plhs[0] = mxCreateDoubleScalar(r);
    } else {
        mexErrMsgIdAndTxt("jstsp:args", "unknown function '%s'", name);
    }
}
#endif  // MATLAB_MEX_FILE
