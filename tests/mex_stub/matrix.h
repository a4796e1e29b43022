/* tests/mex_stub/matrix.h - FIRST-PARTY stand-in for MATLAB's matrix.h (TEST INFRASTRUCTURE).
 *
 * MATLAB is not in the build image, so mex/jstsp_mex.cpp can never meet the real header here.  This file declares
 * exactly the part of the documented MEX C API (interleaved-complex, -R2018a) that the gateway uses, with the
 * documented names and semantics, implemented in stub.cpp on a plain struct.  It exists so that the gateway is compiled
 * by a real compiler and its argument handling, output creation and error routing are executed by
 * tests/test_mex_gateway.py.  It pins nothing about MATLAB itself. */
#ifndef JSTSP_STUB_MATRIX_H
#define JSTSP_STUB_MATRIX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct mxArray_tag mxArray;
typedef struct { double real, imag; } mxComplexDouble;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS = 1, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity flag);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double value);
mxArray *mxCreateCellMatrix(mwSize m, mwSize n);
mxArray *mxCreateString(const char *str);
mxArray *mxDuplicateArray(const mxArray *in);
void mxDestroyArray(mxArray *a);
void mxSetCell(mxArray *cell, mwIndex index, mxArray *value);
mxArray *mxGetCell(const mxArray *cell, mwIndex index);

int mxIsChar(const mxArray *a);
int mxIsDouble(const mxArray *a);
int mxIsComplex(const mxArray *a);
int mxIsEmpty(const mxArray *a);
mxClassID mxGetClassID(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);                       /* product of dimensions 2..end, as in MATLAB */
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
double mxGetScalar(const mxArray *a);
double *mxGetDoubles(const mxArray *a);
mxComplexDouble *mxGetComplexDoubles(const mxArray *a);
void *mxGetData(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);   /* 0 on success, 1 if it does not fit */

#ifdef __cplusplus
}
#endif
#endif
