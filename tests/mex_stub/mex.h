/* tests/mex_stub/mex.h - FIRST-PARTY stand-in for MATLAB's mex.h (TEST INFRASTRUCTURE, see matrix.h). */
#ifndef JSTSP_STUB_MEX_H
#define JSTSP_STUB_MEX_H
#include "matrix.h"
#ifdef __cplusplus
extern "C" {
#endif

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);      /* defined by the gateway */
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);                      /* does not return (long jump) */
int mexAtExit(void (*fn)(void));

/* ---- the test driver's side (not part of MATLAB's API) -------------------------------------------------------------
 * stub_call runs mexFunction under the jump buffer mexErrMsgIdAndTxt returns to: 0 = returned normally, 1 = raised an
 * error (identifier and message via stub_error_id / stub_error_message).  stub_run_at_exit calls what mexAtExit registered. */
int stub_call(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
const char *stub_error_id(void);
const char *stub_error_message(void);
void stub_run_at_exit(void);

#ifdef __cplusplus
}
#endif
#endif
