// tests/mex_stub/stub.cpp - implementation of the stand-in MEX API (TEST INFRASTRUCTURE, see matrix.h).
#include <csetjmp>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "mex.h"

struct mxArray_tag {
    mxClassID cls;
    int complex_;
    mwSize ndim;
    mwSize dims[8];
    void *data;              // doubles / interleaved complex doubles / int32 / chars (one byte each) / mxArray* cells
};

static size_t numel(const mxArray *a)
{
    size_t n = 1;
    for (mwSize i = 0; i < a->ndim; ++i) n *= a->dims[i];
    return n;
}
static size_t elem_size(mxClassID c, int cplx)
{
    switch (c) {
    case mxDOUBLE_CLASS: return cplx ? 16 : 8;
    case mxINT32_CLASS: return 4;
    case mxCHAR_CLASS: return 1;
    case mxCELL_CLASS: return sizeof(mxArray *);
    default: return 0;
    }
}

extern "C" {

mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity flag)
{
    if (ndim > 8) return nullptr;
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    a->cls = cls; a->complex_ = flag == mxCOMPLEX;
    a->ndim = ndim < 2 ? 2 : ndim;
    a->dims[0] = a->dims[1] = 1;
    for (mwSize i = 0; i < ndim; ++i) a->dims[i] = dims[i];
    while (a->ndim > 2 && a->dims[a->ndim - 1] == 1) --a->ndim;          // MATLAB drops trailing singleton dimensions
    const size_t bytes = numel(a) * elem_size(cls, a->complex_);
    a->data = calloc(bytes ? bytes : 1, 1);                              // MATLAB zero-initialises
    return a;
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag)
{
    const mwSize d[2] = {m, n};
    return mxCreateNumericArray(2, d, mxDOUBLE_CLASS, flag);
}
mxArray *mxCreateDoubleScalar(double v)
{
    mxArray *a = mxCreateDoubleMatrix(1, 1, mxREAL);
    ((double *)a->data)[0] = v;
    return a;
}
mxArray *mxCreateCellMatrix(mwSize m, mwSize n)
{
    const mwSize d[2] = {m, n};
    return mxCreateNumericArray(2, d, mxCELL_CLASS, mxREAL);
}
mxArray *mxCreateString(const char *s)
{
    const mwSize d[2] = {1, strlen(s)};
    mxArray *a = mxCreateNumericArray(2, d, mxCHAR_CLASS, mxREAL);
    memcpy(a->data, s, d[1]);
    return a;
}
mxArray *mxDuplicateArray(const mxArray *in)
{
    mxArray *a = mxCreateNumericArray(in->ndim, in->dims, in->cls, in->complex_ ? mxCOMPLEX : mxREAL);
    const size_t n = numel(in);
    if (in->cls == mxCELL_CLASS)
        for (size_t i = 0; i < n; ++i) {
            mxArray *c = ((mxArray **)in->data)[i];
            ((mxArray **)a->data)[i] = c ? mxDuplicateArray(c) : nullptr;
        }
    else
        memcpy(a->data, in->data, n * elem_size(in->cls, in->complex_));
    return a;
}
void mxDestroyArray(mxArray *a)
{
    if (!a) return;
    if (a->cls == mxCELL_CLASS)
        for (size_t i = 0, n = numel(a); i < n; ++i) mxDestroyArray(((mxArray **)a->data)[i]);
    free(a->data);
    free(a);
}
void mxSetCell(mxArray *cell, mwIndex i, mxArray *v) { ((mxArray **)cell->data)[i] = v; }
mxArray *mxGetCell(const mxArray *cell, mwIndex i) { return ((mxArray **)cell->data)[i]; }

int mxIsChar(const mxArray *a) { return a->cls == mxCHAR_CLASS; }
int mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS; }
int mxIsComplex(const mxArray *a) { return a->complex_; }
int mxIsEmpty(const mxArray *a) { return numel(a) == 0; }
mxClassID mxGetClassID(const mxArray *a) { return a->cls; }
size_t mxGetM(const mxArray *a) { return a->dims[0]; }
size_t mxGetN(const mxArray *a)
{
    size_t n = 1;
    for (mwSize i = 1; i < a->ndim; ++i) n *= a->dims[i];
    return n;
}
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->ndim; }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims; }
size_t mxGetNumberOfElements(const mxArray *a) { return numel(a); }
double mxGetScalar(const mxArray *a)
{
    if (numel(a) == 0) return 0.0;
    if (a->cls == mxINT32_CLASS) return (double)((int32_t *)a->data)[0];
    if (a->cls == mxCHAR_CLASS) return (double)((unsigned char *)a->data)[0];
    return ((double *)a->data)[0];                                       // real part of the first element
}
double *mxGetDoubles(const mxArray *a) { return (a->cls == mxDOUBLE_CLASS && !a->complex_) ? (double *)a->data : nullptr; }
mxComplexDouble *mxGetComplexDoubles(const mxArray *a)
{
    return (a->cls == mxDOUBLE_CLASS && a->complex_) ? (mxComplexDouble *)a->data : nullptr;
}
void *mxGetData(const mxArray *a) { return a->data; }
int mxGetString(const mxArray *a, char *buf, mwSize buflen)
{
    if (a->cls != mxCHAR_CLASS || buflen == 0) return 1;
    const size_t n = numel(a);
    const size_t c = n < buflen - 1 ? n : buflen - 1;
    memcpy(buf, a->data, c);
    buf[c] = 0;
    return n > buflen - 1;
}

// ---- error routing and the driver's side ---------------------------------------------------------------------------
static jmp_buf g_jmp;
static int g_armed = 0;
static char g_id[128], g_msg[1024];
static void (*g_at_exit)(void) = nullptr;

void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...)
{
    snprintf(g_id, sizeof(g_id), "%s", id ? id : "");
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_msg, sizeof(g_msg), fmt, ap);
    va_end(ap);
    if (g_armed) longjmp(g_jmp, 1);
    fprintf(stderr, "mexErrMsgIdAndTxt outside stub_call: %s: %s\n", g_id, g_msg);
    abort();
}
int mexAtExit(void (*fn)(void)) { g_at_exit = fn; return 0; }

int stub_call(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    g_id[0] = g_msg[0] = 0;
    g_armed = 1;
    if (setjmp(g_jmp)) { g_armed = 0; return 1; }
    mexFunction(nlhs, plhs, nrhs, prhs);
    g_armed = 0;
    return 0;
}
const char *stub_error_id(void) { return g_id; }
const char *stub_error_message(void) { return g_msg; }
void stub_run_at_exit(void) { if (g_at_exit) g_at_exit(); }

}  // extern "C"
