"""jstsp19_amd — MI355X (gfx950) implementation of the sparse wideband-mmWave channel
estimation solver path of vlaxose/jstsp19, behind the reference's own function signatures.

The compute lives in ``csrc/libjstsp_mi355x.so`` (hand-written HIP, C ABI in
``include/jstsp.h``); this package is the host-side mirror of the reference interface.
There is no CPU fallback: without the built library / without a GPU the calls raise.
"""
from ._lib import Context, JstspError, default_context, load, LIB_PATH, HOST, DEVICE  # noqa: F401
from .solvers import (OMP, sparse_sca_estim, cawgn_estim_out, gradient_head, ls_estimate, colmajor, correlate, empty_colmajor, mc_admm, mc_svt, nmse_spectral, lambda_max_sequence,  # noqa: F401
                      omp_kron, pinv, mmv_omp, tssr, rate, proposed_algorithm, proposed_algorithm_begin, proposed_algorithm_angles, sparse_admm, svt, synthesize, vamp,
                      vamp_kron)

__version__ = "0.1.0"
