// MEX gateway of libjstsp_mi355x.so - one entry point, dispatched by name, used by the .m wrappers in this directory
// (proposed_algorithm.m, OMP.m, ...), which carry the reference's exact signatures so that the plot_errorVS* drivers
// run unchanged.
//
//   mex -R2018a -I../include jstsp_mex.cpp -L../jstsp19_amd/csrc -ljstsp_mi355x
//
// MATLAB's arrays go to the library AS THEY ARE: column-major interleaved complex double is exactly the layout of the
// _c64 entry points of include/jstsp.h (mxGetComplexDoubles -> jstsp_c64*, no conversion loop on the host; the
// narrowing to the solvers' fp32 happens on the device).  A trailing third dimension is the batch of the C ABI:
// subY N x M x batch etc. -> ONE call for all realisations of a driver's `for r = 1:maxMCRealizations` loop
// (plot_errorVSsnr.m:51); per-problem scalars (tau_Y, tau_S, rho, tau) may be scalars or vectors of length batch;
// a 2-D dictionary (A, B) is shared by the batch, a 3-D one is per problem.
//
// Conventions follow the only in-tree MEX exemplar of the reference,
// MPbased_solvers/BiGAMP/comparison_codes/ompbox10/private/ompmex.c:39-133 (nrhs / nlhs checks up front,
// mexErrMsgIdAndTxt for every failure, outputs via mxCreate*), with the interleaved-complex API (-R2018a).
// mexErrMsgIdAndTxt long-jumps: no C++ object with a destructor is live where it is called (everything that owns
// memory is an mxArray, which MATLAB frees itself on error).
//
// No MATLAB in the build image: the file is compiled and driven in tests/test_mex_gateway.py against a first-party
// stand-in for mex.h / matrix.h (tests/mex_stub/) that implements the few API calls used here.
#ifdef MATLAB_MEX_FILE
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "mex.h"
#include "jstsp.h"

static jstsp_ctx *g_ctx = nullptr;
static void at_exit() { if (g_ctx) { jstsp_destroy(g_ctx); g_ctx = nullptr; } }

static void fail(const char *what, int rc)
{
    static char msg[768];
    snprintf(msg, sizeof(msg), "%s failed (%d): %s", what, rc, jstsp_last_error());
    mexErrMsgIdAndTxt("jstsp:call", "%s", msg);
}

struct Dims { int r, c, b; };

// rows, columns and batch (trailing dimensions folded) of a numeric array
static Dims dims_of(const mxArray *a)
{
    const mwSize nd = mxGetNumberOfDimensions(a);
    const mwSize *d = mxGetDimensions(a);
    Dims o{(int)d[0], nd > 1 ? (int)d[1] : 1, 1};
    for (mwSize i = 2; i < nd; ++i) o.b *= (int)d[i];
    return o;
}

static void need_double(const mxArray *a, const char *fn, const char *name)
{
    if (!mxIsDouble(a) || mxIsEmpty(a)) mexErrMsgIdAndTxt("jstsp:args", "%s: %s must be a non-empty double array", fn, name);
}

// interleaved complex view of an input; a REAL array (MATLAB keeps e.g. a real-valued dictionary real) is widened into a
// temporary complex mxArray, which MATLAB owns and frees when the MEX call ends
static const jstsp_c64 *cplx(const mxArray *a, const char *fn, const char *name)
{
    need_double(a, fn, name);
    if (mxIsComplex(a)) return reinterpret_cast<const jstsp_c64 *>(mxGetComplexDoubles(a));
    mxArray *t = mxCreateNumericArray(mxGetNumberOfDimensions(a), mxGetDimensions(a), mxDOUBLE_CLASS, mxCOMPLEX);
    mxComplexDouble *p = mxGetComplexDoubles(t);
    const double *s = mxGetDoubles(a);
    const size_t n = mxGetNumberOfElements(a);
    for (size_t i = 0; i < n; ++i) { p[i].real = s[i]; p[i].imag = 0.0; }
    return reinterpret_cast<const jstsp_c64 *>(p);
}

static const double *real_of(const mxArray *a, const char *fn, const char *name)
{
    need_double(a, fn, name);
    if (mxIsComplex(a)) mexErrMsgIdAndTxt("jstsp:args", "%s: %s must be real", fn, name);
    return mxGetDoubles(a);
}

// per-problem scalars: a scalar is repeated, a vector must have one entry per problem (temporary owned by MATLAB)
static const double *scalars(const mxArray *a, int batch, const char *fn, const char *name)
{
    const double *p = real_of(a, fn, name);
    const size_t n = mxGetNumberOfElements(a);
    if ((int)n == batch) return p;
    if (n != 1) mexErrMsgIdAndTxt("jstsp:args", "%s: %s must be a scalar or have one entry per problem (%d)", fn, name, batch);
    mxArray *t = mxCreateDoubleMatrix(batch, 1, mxREAL);
    double *q = mxGetDoubles(t);
    for (int i = 0; i < batch; ++i) q[i] = p[0];
    return q;
}

static mxArray *new_complex(int r, int c, int b)
{
    const mwSize d[3] = {(mwSize)r, (mwSize)c, (mwSize)b};
    return mxCreateNumericArray(b > 1 ? 3 : 2, d, mxDOUBLE_CLASS, mxCOMPLEX);
}
static mxArray *new_real(int r, int c, int b)
{
    const mwSize d[3] = {(mwSize)r, (mwSize)c, (mwSize)b};
    return mxCreateNumericArray(b > 1 ? 3 : 2, d, mxDOUBLE_CLASS, mxREAL);
}
static jstsp_c64 *c64(mxArray *a) { return reinterpret_cast<jstsp_c64 *>(mxGetComplexDoubles(a)); }

// stride (in elements) between the problems of a dictionary argument: 0 when it is 2-D (shared by the batch)
static long long dict_stride(const Dims &d, int batch, const char *fn, const char *name)
{
    if (d.b == 1) return 0;
    if (d.b != batch) mexErrMsgIdAndTxt("jstsp:shape", "%s: %s has %d pages, expected 1 or %d", fn, name, d.b, batch);
    return (long long)d.r * d.c;
}

static void check_nargs(const char *fn, int nrhs, int lo, int hi, int nlhs, int max_out)
{
    if (nrhs - 1 < lo || nrhs - 1 > hi)
        mexErrMsgIdAndTxt("jstsp:args", "%s: %d input arguments given, %d to %d expected", fn, nrhs - 1, lo, hi);
    if (nlhs > max_out) mexErrMsgIdAndTxt("jstsp:args", "%s: too many output arguments (%d, at most %d)", fn, nlhs, max_out);
}

static void ensure_ctx()
{
    if (g_ctx) return;
    // one context per MATLAB process; parfor workers are separate processes: JSTSP_DEVICE spreads them over the GPUs
    const char *env = getenv("JSTSP_DEVICE");
    const int rc = jstsp_create(env ? atoi(env) : 0, &g_ctx);
    if (rc) { g_ctx = nullptr; fail("jstsp_create", rc); }
    mexAtExit(at_exit);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("jstsp:args", "first argument must be the function name");
    char fn[64];
    if (mxGetString(prhs[0], fn, sizeof(fn))) mexErrMsgIdAndTxt("jstsp:args", "function name too long");
    const mxArray *const *in = prhs + 1;         // in[0] is the reference function's first argument

    if (!strcmp(fn, "proposed_algorithm")) {
        // [S, Y, convergence_error] = proposed_algorithm(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type)
        //   basic_system_functions/proposed_algorithm.m:1; a 10th argument indx_S makes it proposed_algorithm_angles.m:1
        check_nargs(fn, nrhs, 9, 10, nlhs, 3);
        const Dims dy = dims_of(in[0]), dom = dims_of(in[1]), da = dims_of(in[2]), db = dims_of(in[3]);
        const int N = dy.r, M = dy.c, batch = dy.b, Gr = da.c, G2 = db.r;
        if (dom.r != N || dom.c != M || dom.b != batch || da.r != N || db.c != M)
            mexErrMsgIdAndTxt("jstsp:shape", "proposed_algorithm: inconsistent dimensions");
        const jstsp_c64 *subY = cplx(in[0], fn, "subY"), *A = cplx(in[2], fn, "A"), *B = cplx(in[3], fn, "B");
        const double *Om = real_of(in[1], fn, "Omega");
        const long long sA = dict_stride(da, batch, fn, "A"), sB = dict_stride(db, batch, fn, "B");
        const int Imax = (int)mxGetScalar(in[4]);
        if (Imax < 1) mexErrMsgIdAndTxt("jstsp:args", "proposed_algorithm: Imax must be >= 1");
        const double *tY = scalars(in[5], batch, fn, "tau_Y"), *tS = scalars(in[6], batch, fn, "tau_S"),
                     *rho = scalars(in[7], batch, fn, "rho");
        char type[32] = "";
        if (!mxIsChar(in[8]) || mxGetString(in[8], type, sizeof(type)))
            mexErrMsgIdAndTxt("jstsp:args", "proposed_algorithm: type must be a short char array");
        const int tcode = strcmp(type, "approximate") == 0 ? JSTSP_TYPE_APPROXIMATE : JSTSP_TYPE_STD;      // :23,:45
        const int32_t *idx = nullptr;
        if (nrhs - 1 >= 10 && !mxIsEmpty(in[9])) {            // 1-based linear indices (plot_errorVSsnr.m:143), doubles in MATLAB
            const size_t n = mxGetNumberOfElements(in[9]);
            if (n != (size_t)Gr * G2 * batch) mexErrMsgIdAndTxt("jstsp:shape", "proposed_algorithm_angles: numel(indx_S) must be Gr*G2 per problem");
            const double *pi = real_of(in[9], fn, "indx_S");
            const mwSize d1[2] = {(mwSize)n, 1};
            mxArray *t = mxCreateNumericArray(2, d1, mxINT32_CLASS, mxREAL);
            int32_t *q = (int32_t *)mxGetData(t);
            for (size_t i = 0; i < n; ++i) q[i] = (int32_t)pi[i];
            idx = q;
        }
        ensure_ctx();
        plhs[0] = new_complex(Gr, G2, batch);
        mxArray *Y = nlhs >= 2 ? new_complex(N, M, batch) : nullptr;
        mxArray *ce = nlhs >= 3 ? new_real(Imax, 3, batch) : nullptr;          // spectral norms only if requested
        const int rc = jstsp_proposed_algorithm_c64(g_ctx, N, M, Gr, G2, batch, subY, Om, A, sA, B, sB, Imax, tY, tS, rho, tcode,
                                                    idx, c64(plhs[0]), Y ? c64(Y) : nullptr, ce ? mxGetDoubles(ce) : nullptr,
                                                    JSTSP_HOST);
        if (rc) fail("jstsp_proposed_algorithm_c64", rc);
        if (Y) plhs[1] = Y;
        if (ce) plhs[2] = ce;
    } else if (!strcmp(fn, "svt")) {
        // X = svt(Y, tau)     benchmark_algorithms/svt.m:1
        check_nargs(fn, nrhs, 2, 2, nlhs, 1);
        const Dims d = dims_of(in[0]);
        const jstsp_c64 *Y = cplx(in[0], fn, "Y");
        const double *tau = scalars(in[1], d.b, fn, "tau");
        ensure_ctx();
        plhs[0] = new_complex(d.r, d.c, d.b);
        const int rc = jstsp_svt_c64(g_ctx, d.r, d.c, d.b, Y, tau, c64(plhs[0]), JSTSP_HOST);
        if (rc) fail("jstsp_svt_c64", rc);
    } else if (!strcmp(fn, "ls")) {
        // S_ls = pinv(A)*Y*pinv(B)   (plot_errorVSsnr.m:83) as ls(Y, A, B)
        check_nargs(fn, nrhs, 3, 3, nlhs, 1);
        const Dims dy = dims_of(in[0]), da = dims_of(in[1]), db = dims_of(in[2]);
        if (da.r != dy.r || db.c != dy.c) mexErrMsgIdAndTxt("jstsp:shape", "ls: inconsistent dimensions");
        const jstsp_c64 *Y = cplx(in[0], fn, "Y"), *A = cplx(in[1], fn, "A"), *B = cplx(in[2], fn, "B");
        const long long sA = dict_stride(da, dy.b, fn, "A"), sB = dict_stride(db, dy.b, fn, "B");
        ensure_ctx();
        plhs[0] = new_complex(da.c, db.r, dy.b);
        const int rc = jstsp_ls_c64(g_ctx, dy.r, dy.c, da.c, db.r, dy.b, Y, A, sA, B, sB, c64(plhs[0]), JSTSP_HOST);
        if (rc) fail("jstsp_ls_c64", rc);
    } else if (!strcmp(fn, "OMP")) {
        // [x_hat, indexSet, v, targetMatrix] = OMP(A, v, m, snr)     benchmark_algorithms/OMP.m:1 (snr is unused there too)
        check_nargs(fn, nrhs, 3, 4, nlhs, 4);
        const Dims da = dims_of(in[0]), dv = dims_of(in[1]);
        const int meas = da.r, size_d = da.c, batch = dv.c * dv.b;        // v: measures x 1 (x batch columns)
        if (dv.r != meas) mexErrMsgIdAndTxt("jstsp:shape", "OMP: size(v,1) must equal size(A,1)");
        const int m = (int)mxGetScalar(in[2]);
        if (m < 1) mexErrMsgIdAndTxt("jstsp:args", "OMP: m must be >= 1");
        const jstsp_c64 *A = cplx(in[0], fn, "A"), *v = cplx(in[1], fn, "v");
        const long long sA = da.b == 1 ? 0 : dict_stride(da, batch, fn, "A");
        ensure_ctx();
        plhs[0] = new_complex(size_d, batch, 1);
        const mwSize di[2] = {(mwSize)m, (mwSize)batch};
        mxArray *ix = mxCreateNumericArray(2, di, mxINT32_CLASS, mxREAL);
        mxArray *T = nlhs >= 4 ? new_complex(meas, m, batch) : nullptr;
        const int rc = jstsp_omp_c64(g_ctx, meas, size_d, batch, A, sA, v, m, c64(plhs[0]), (int32_t *)mxGetData(ix),
                                     T ? c64(T) : nullptr, JSTSP_HOST);
        if (rc) fail("jstsp_omp_c64", rc);
        if (nlhs >= 2) {                                                  // indexSet: 1 x m cell (OMP.m:9,22), first problem
            plhs[1] = mxCreateCellMatrix(1, m);
            const int32_t *pi = (const int32_t *)mxGetData(ix);
            for (int i = 0; i < m; ++i) mxSetCell(plhs[1], i, mxCreateDoubleScalar((double)pi[i]));
        }
        if (nlhs >= 3) plhs[2] = mxDuplicateArray(in[1]);
        if (T) plhs[3] = T;
    } else if (!strcmp(fn, "sparse_admm")) {
        // [S, convergence_error] = sparse_admm(Htrue, OH, Dr, Dt, Imax)     benchmark_algorithms/sparse_admm.m:1
        check_nargs(fn, nrhs, 5, 5, nlhs, 2);
        const Dims d = dims_of(in[1]), dh = dims_of(in[0]), dr = dims_of(in[2]), dt = dims_of(in[3]);
        if (dh.r != d.r || dh.c != d.c || dh.b != d.b || dr.r != d.r || dt.r != d.c)
            mexErrMsgIdAndTxt("jstsp:shape", "sparse_admm: inconsistent dimensions");
        const int Imax = (int)mxGetScalar(in[4]);
        if (Imax < 1) mexErrMsgIdAndTxt("jstsp:args", "sparse_admm: Imax must be >= 1");
        const jstsp_c64 *H = cplx(in[0], fn, "Htrue"), *OH = cplx(in[1], fn, "OH"), *Dr = cplx(in[2], fn, "Dr"),
                        *Dt = cplx(in[3], fn, "Dt");
        ensure_ctx();
        plhs[0] = new_complex(d.r, d.c, d.b);
        mxArray *ce = new_real(Imax, 1, d.b);
        const int rc = jstsp_sparse_admm_c64(g_ctx, d.r, d.c, dr.c, dt.c, d.b, H, OH, Dr, Dt, Imax, c64(plhs[0]), mxGetDoubles(ce),
                                             JSTSP_HOST);
        if (rc) fail("jstsp_sparse_admm_c64", rc);
        if (nlhs >= 2) plhs[1] = ce;
    } else if (!strcmp(fn, "mc_svt")) {
        // X = mc_svt(OH, Omega, Imax, tau, rho)     benchmark_algorithms/mc_svt.m:1
        check_nargs(fn, nrhs, 5, 5, nlhs, 1);
        const Dims d = dims_of(in[0]), dom = dims_of(in[1]);
        if (dom.r != d.r || dom.c != d.c || dom.b != d.b) mexErrMsgIdAndTxt("jstsp:shape", "mc_svt: Omega must have the shape of OH");
        const jstsp_c64 *OH = cplx(in[0], fn, "OH");
        const double *Om = real_of(in[1], fn, "Omega");
        const int Imax = (int)mxGetScalar(in[2]);
        const double *tau = scalars(in[3], d.b, fn, "tau"), *rho = scalars(in[4], d.b, fn, "rho");
        ensure_ctx();
        plhs[0] = new_complex(d.r, d.c, d.b);
        const int rc = jstsp_mc_svt_c64(g_ctx, d.r, d.c, d.b, OH, Om, Imax, tau, rho, c64(plhs[0]), JSTSP_HOST);
        if (rc) fail("jstsp_mc_svt_c64", rc);
    } else if (!strcmp(fn, "mc_admm")) {
        // [X, convergence_error] = mc_admm(Htrue, OH, Omega, Imax, tau, rho)     benchmark_algorithms/mc_admm.m:1
        check_nargs(fn, nrhs, 6, 6, nlhs, 2);
        const Dims d = dims_of(in[1]), dh = dims_of(in[0]), dom = dims_of(in[2]);
        if (dom.r != d.r || dom.c != d.c || dom.b != d.b || dh.r != d.r || dh.c != d.c || dh.b != d.b)
            mexErrMsgIdAndTxt("jstsp:shape", "mc_admm: inconsistent dimensions");
        const jstsp_c64 *H = cplx(in[0], fn, "Htrue"), *OH = cplx(in[1], fn, "OH");
        const double *Om = real_of(in[2], fn, "Omega");
        const int Imax = (int)mxGetScalar(in[3]);
        if (Imax < 1) mexErrMsgIdAndTxt("jstsp:args", "mc_admm: Imax must be >= 1");
        const double *tau = scalars(in[4], d.b, fn, "tau"), *rho = scalars(in[5], d.b, fn, "rho");
        ensure_ctx();
        plhs[0] = new_complex(d.r, d.c, d.b);
        mxArray *ce = new_real(Imax, 1, d.b);
        const int rc = jstsp_mc_admm_c64(g_ctx, d.r, d.c, d.b, H, OH, Om, Imax, tau, rho, c64(plhs[0]), mxGetDoubles(ce), JSTSP_HOST);
        if (rc) fail("jstsp_mc_admm_c64", rc);
        if (nlhs >= 2) plhs[1] = ce;
    } else if (!strcmp(fn, "vamp")) {
        // x = vamp(y, A, sigma, L)     benchmark_algorithms/vamp.m:1 (always 100 iterations, VampGlmEst.m:509-511)
        check_nargs(fn, nrhs, 4, 4, nlhs, 1);
        const Dims da = dims_of(in[1]), dy = dims_of(in[0]);
        const int batch = dy.c * dy.b;
        if (dy.r != da.r) mexErrMsgIdAndTxt("jstsp:shape", "vamp: length(y) must equal size(A,1)");
        const jstsp_c64 *y = cplx(in[0], fn, "y"), *A = cplx(in[1], fn, "A");
        const long long sA = da.b == 1 ? 0 : dict_stride(da, batch, fn, "A");
        ensure_ctx();
        plhs[0] = new_complex(da.c, batch, 1);
        const int rc = jstsp_vamp_c64(g_ctx, da.r, da.c, batch, y, A, sA, mxGetScalar(in[2]), mxGetScalar(in[3]), 100, c64(plhs[0]),
                                      JSTSP_HOST);
        if (rc) fail("jstsp_vamp_c64", rc);
    } else if (!strcmp(fn, "vamp_kron")) {
        // X = vamp_kron(Y, A, Gb, sigma, L): the drivers' vamp(vec(Y), kron(Gb.', A), sigma, L) (plot_errorVSsnr.m:79-80,100)
        // without forming the Kronecker matrix; X is Gr x G2 (x = vec(X))
        check_nargs(fn, nrhs, 5, 5, nlhs, 1);
        const Dims dy = dims_of(in[0]), da = dims_of(in[1]), dg = dims_of(in[2]);
        if (da.r != dy.r || dg.r != dy.c || dg.c != dy.c) mexErrMsgIdAndTxt("jstsp:shape", "vamp_kron: inconsistent dimensions");
        const jstsp_c64 *Y = cplx(in[0], fn, "Y"), *A = cplx(in[1], fn, "A"), *G = cplx(in[2], fn, "Gb");
        const long long sA = dict_stride(da, dy.b, fn, "A"), sG = dict_stride(dg, dy.b, fn, "Gb");
        ensure_ctx();
        plhs[0] = new_complex(da.c, dy.c, dy.b);
        const int rc = jstsp_vamp_kron_c64(g_ctx, dy.r, da.c, dy.c, dy.b, Y, A, sA, G, sG, mxGetScalar(in[3]), mxGetScalar(in[4]), 100,
                                           c64(plhs[0]), JSTSP_HOST);
        if (rc) fail("jstsp_vamp_kron_c64", rc);
    } else if (!strcmp(fn, "mmv_omp")) {
        // [Z, support] = mmv_omp(A, K, Y)   replaces spx.pursuit.joint.OrthogonalMatchingPursuit(A, K).solve(Y).Z
        // (plot_errorVSsnr.m:116-117; sparse-plex is not vendored: published simultaneous OMP, l2 row score)
        check_nargs(fn, nrhs, 3, 3, nlhs, 2);
        const Dims da = dims_of(in[0]), dy = dims_of(in[2]);
        const int K = (int)mxGetScalar(in[1]);
        if (dy.r != da.r) mexErrMsgIdAndTxt("jstsp:shape", "mmv_omp: size(Y,1) must equal size(A,1)");
        if (K < 1) mexErrMsgIdAndTxt("jstsp:args", "mmv_omp: K must be >= 1");
        const jstsp_c64 *A = cplx(in[0], fn, "A"), *Y = cplx(in[2], fn, "Y");
        const long long sA = dict_stride(da, dy.b, fn, "A");
        ensure_ctx();
        plhs[0] = new_complex(da.c, dy.c, dy.b);
        const mwSize di[2] = {(mwSize)K, (mwSize)dy.b}, dc[2] = {(mwSize)dy.b, 1};
        mxArray *ix = mxCreateNumericArray(2, di, mxINT32_CLASS, mxREAL), *cn = mxCreateNumericArray(2, dc, mxINT32_CLASS, mxREAL);
        const int rc = jstsp_mmv_omp_c64(g_ctx, da.r, da.c, dy.c, dy.b, A, sA, Y, K, 2, c64(plhs[0]), (int32_t *)mxGetData(ix),
                                         (int32_t *)mxGetData(cn), JSTSP_HOST);
        if (rc) fail("jstsp_mmv_omp_c64", rc);
        if (nlhs >= 2) {                                                  // support of the first problem, selection order
            const int cnt = ((const int32_t *)mxGetData(cn))[0];
            plhs[1] = mxCreateDoubleMatrix(1, cnt, mxREAL);
            for (int i = 0; i < cnt; ++i) mxGetDoubles(plhs[1])[i] = (double)((const int32_t *)mxGetData(ix))[i];
        }
    } else if (!strcmp(fn, "pinv")) {
        // P = pinv(A) through the float64 SVD kernel (small matrices; MATLAB's own pinv works as well)
        check_nargs(fn, nrhs, 1, 1, nlhs, 1);
        const Dims d = dims_of(in[0]);
        const jstsp_c64 *A = cplx(in[0], fn, "A");
        ensure_ctx();
        plhs[0] = new_complex(d.c, d.r, d.b);
        const int rc = jstsp_pinv_c64(g_ctx, d.r, d.c, d.b, A, c64(plhs[0]), JSTSP_HOST);
        if (rc) fail("jstsp_pinv_c64", rc);
    } else if (!strcmp(fn, "nmse") || !strcmp(fn, "rate")) {
        // nmse(S, Zbar): min(1, norm(S-Zbar)^2/norm(Zbar)^2)  plot_errorVSsnr.m:138-141 (one value per page)
        // rate(S, Zbar, noise_var): log2(real(det(eye(Nr) + 1/Nr*Zbar*Zbar'/(noise_var + nmse))))  plot_rateVSframelength.m:81
        const bool is_rate = fn[0] == 'r';
        check_nargs(fn, nrhs, is_rate ? 3 : 2, is_rate ? 3 : 2, nlhs, 1);
        const Dims d = dims_of(in[0]), dz = dims_of(in[1]);
        if (dz.r != d.r || dz.c != d.c || dz.b != d.b) mexErrMsgIdAndTxt("jstsp:shape", "%s: S and Zbar must have the same size", fn);
        const jstsp_c64 *S = cplx(in[0], fn, "S"), *Zb = cplx(in[1], fn, "Zbar");
        ensure_ctx();
        plhs[0] = mxCreateDoubleMatrix(d.b, 1, mxREAL);
        const int rc = is_rate ? jstsp_rate_c64(g_ctx, d.r, d.c, d.b, S, Zb, mxGetScalar(in[2]), mxGetDoubles(plhs[0]), JSTSP_HOST)
                               : jstsp_nmse_spectral_c64(g_ctx, d.r, d.c, d.b, S, Zb, mxGetDoubles(plhs[0]), JSTSP_HOST);
        if (rc) fail(is_rate ? "jstsp_rate_c64" : "jstsp_nmse_spectral_c64", rc);
    } else {
        mexErrMsgIdAndTxt("jstsp:args", "unknown function '%s'", fn);
    }
}
#endif  // MATLAB_MEX_FILE
