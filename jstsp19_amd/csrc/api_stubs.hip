// Entry points declared in include/jstsp.h whose kernels are not written yet.
#include "solver_common.h"
using namespace jstsp;
extern "C" {
int jstsp_omp_c32(jstsp_ctx *, int, int, int, const jstsp_c32 *, long long, const jstsp_c32 *, int,
                  jstsp_c32 *, int32_t *, jstsp_c32 *, int)
{
    set_error("jstsp_omp_c32: not implemented yet");
    return JSTSP_E_UNSUPPORTED;
}
int jstsp_omp_kron_c32(jstsp_ctx *, int, int, int, int, int, const jstsp_c32 *, long long, const jstsp_c32 *,
                       long long, const jstsp_c32 *, int, jstsp_c32 *, int32_t *, int)
{
    set_error("jstsp_omp_kron_c32: not implemented yet");
    return JSTSP_E_UNSUPPORTED;
}
int jstsp_sparse_admm_c32(jstsp_ctx *, int, int, int, int, int, const jstsp_c32 *, const jstsp_c32 *,
                          const jstsp_c32 *, const jstsp_c32 *, int, jstsp_c32 *, double *, int)
{
    set_error("jstsp_sparse_admm_c32: not implemented yet");
    return JSTSP_E_UNSUPPORTED;
}
}
