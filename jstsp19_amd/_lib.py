"""ctypes binding of libjstsp_mi355x.so (the C ABI declared in include/jstsp.h).

There is NO CPU fallback: importing this module without the built library, or creating a
context without an MI355X, raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JSTSP_EXPERIMENTS_LIB=1 (tools/ only): the -DJSTSP_EXPERIMENTS build (JSTSP_EXPERIMENTS=1 python jstsp19_amd/build.py), in which the
# switches of dropped experiments are still read from the environment (csrc/common.h)
LIB_PATH = os.path.join(_HERE, "csrc", "libjstsp_mi355x_xp.so" if os.environ.get("JSTSP_EXPERIMENTS_LIB") == "1"
                        else "libjstsp_mi355x.so")

HOST, DEVICE = 0, 1
TYPE_APPROXIMATE, TYPE_STD = 0, 1
BF_ZC, BF_DFT = 0, 1
RHO_MIN6, RHO_MAX = 0, 1
PILOTS_QAM4, PILOTS_GAUSS = 0, 1

c_void_p, c_int, c_ll, c_dp, c_ip = C.c_void_p, C.c_int, C.c_longlong, C.POINTER(C.c_double), C.POINTER(C.c_int)



class Model(C.Structure):
    """struct jstsp_model (include/jstsp.h)."""
    _fields_ = [(n, c_int) for n in ("Nt", "Nr", "L", "T_prop", "Mr", "Mr_e", "Gr", "Gt", "clusters", "rays",
                                     "T_hbf", "shared_pilots")] + [("noise_var", C.c_double), ("beamformer", c_int),
                                                                   ("rho_rule", c_int), ("rho_scale", C.c_double),
                                                                   ("pilots", c_int)]


class Trials(C.Structure):
    """struct jstsp_trials (include/jstsp.h): output pointers, NULL = not wanted."""
    _fields_ = [(n, c_void_p) for n in ("subY", "Omega", "A", "B", "Zbar", "H", "indx_S")] + \
               [(n, c_dp) for n in ("tau_Y", "tau_Z", "rho")] + \
               [(n, c_void_p) for n in ("Y_hbf", "A_hbf", "B_hbf", "gains", "u_r", "u_t", "noise", "qam_idx", "pilot_sym")]


# name -> (restype, argtypes); mirrors include/jstsp.h one to one
SIGNATURES = {
    "jstsp_create": (c_int, [c_int, C.POINTER(c_void_p)]),
    "jstsp_destroy": (c_int, [c_void_p]),
    "jstsp_set_stream": (c_int, [c_void_p, c_void_p]),
    "jstsp_use_own_stream": (c_int, [c_void_p]),
    "jstsp_synchronize": (c_int, [c_void_p]),
    "jstsp_last_error": (C.c_char_p, []),
    "jstsp_version": (C.c_char_p, []),
    "jstsp_workspace_bytes": (C.c_size_t, [c_void_p]),
    "jstsp_correlate_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_ll,
                                    c_void_p, c_ll, c_void_p, c_int]),
    "jstsp_synthesize_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_ll,
                                     c_void_p, c_ll, c_void_p, c_int]),
    "jstsp_gradient_head_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_ll,
                                        c_void_p, c_void_p, c_void_p, c_int]),
    "jstsp_proposed_algorithm_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                             c_void_p, c_ll, c_void_p, c_ll, c_int, c_dp, c_dp, c_dp, c_int,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "jstsp_proposed_algorithm_begin_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                   c_void_p, c_ll, c_void_p, c_ll, c_int, c_dp, c_dp, c_dp, c_int,
                                                   c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(c_void_p)]),
    "jstsp_proposed_algorithm_end": (c_int, [c_void_p, c_void_p, c_ip]),
    "jstsp_ls_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_ll,
                             c_void_p, c_int]),
    "jstsp_pinv_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int]),
    "jstsp_last_conditioning": (c_int, [c_void_p, c_dp, c_dp]),
    "jstsp_last_fused_fallbacks": (c_int, [c_void_p, c_ip]),
    "jstsp_last_lanczos_mismatches": (c_int, [c_void_p, c_ip]),
    "jstsp_last_dictionary_block": (c_int, [c_void_p, c_ip]),
    "jstsp_svt_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_dp, c_void_p, c_int]),
    "jstsp_omp_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_ll, c_void_p, c_int, c_void_p,
                              c_void_p, c_void_p, c_int]),
    "jstsp_omp_kron_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_ll, c_void_p, c_ll,
                                   c_void_p, c_int, c_void_p, c_void_p, c_int]),
    "jstsp_mmv_omp_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_ll, c_void_p, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_int]),
    "jstsp_rate_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, C.c_double, c_void_p, c_int]),
    "jstsp_sparse_admm_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]),
    "jstsp_mc_svt_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_dp, c_dp,
                                 c_void_p, c_int]),
    "jstsp_mc_admm_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_dp,
                                  c_dp, c_void_p, c_void_p, c_int]),
    "jstsp_vamp_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_ll, C.c_double, C.c_double, c_int,
                               c_void_p, c_int]),
    "jstsp_vamp_kron_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_ll,
                                    C.c_double, C.c_double, c_int, c_void_p, c_int]),
    "jstsp_sparse_sca_estim_f64": (c_int, [c_void_p, c_ll, c_void_p, C.c_double, C.c_double, C.c_double, c_void_p, c_void_p, c_int]),
    "jstsp_cawgn_estim_out_f64": (c_int, [c_void_p, c_ll, c_void_p, c_void_p, C.c_double, C.c_double, c_void_p, c_dp, c_int]),
    "jstsp_lambda_max_sequence_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int]),
    "jstsp_nmse_spectral_c32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "jstsp_build_trials_c32": (c_int, [c_void_p, C.POINTER(Model), C.c_uint64, c_int, c_ll, c_int, C.POINTER(Trials),
                                       c_int]),
    "jstsp_set_profiling": (c_int, [c_void_p, c_int]),
    "jstsp_get_profile": (c_int, [c_void_p, C.c_char_p, c_ip, c_dp]),
}


for _n in ("correlate", "synthesize", "proposed_algorithm", "svt", "omp", "sparse_admm", "mc_svt", "mc_admm", "vamp", "ls", "pinv",
           "mmv_omp", "vamp_kron", "nmse_spectral", "rate"):
    # the double-complex forms take the same argument lists (pointers are void* here)
    SIGNATURES["jstsp_%s_c64" % _n] = SIGNATURES["jstsp_%s_c32" % _n]


class JstspError(RuntimeError):
    """A failed C-ABI call; ``code`` is its status (< 0: JSTSP_E_*, > 0: hipError_t), None when raised on the Python side."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


E_UNSUPPORTED, E_ILLCOND = -3, -6          # include/jstsp.h


_lib = None


def load():
    """Load the shared library (once) and declare every prototype of include/jstsp.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JstspError(
            "%s is missing: build it with `python -m jstsp19_amd.build` (hipcc, gfx950). "
            "jstsp19_amd has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    # libhsa-runtime64; if ours (from /opt/rocm) were loaded first, torch would later find
    # "No HIP GPUs".  Importing torch first makes the loader resolve this library's
    # libamdhip64.so.7 dependency to the copy that is already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().jstsp_last_error().decode("utf-8", "replace")
        raise JstspError("%s failed with code %d: %s" % (what or "jstsp call", rc, msg), rc)


class Context:
    """One jstsp_ctx: a HIP stream + grow-only device workspace on one GPU."""

    def __init__(self, device=0):
        lib = load()
        h = c_void_p()
        check(lib.jstsp_create(int(device), C.byref(h)), "jstsp_create")
        self._h = h
        self.device = int(device)
        self._lib = lib

    @property
    def handle(self):
        if self._h is None:
            raise JstspError("context already destroyed")
        return self._h

    def set_stream(self, stream_ptr):
        check(self._lib.jstsp_set_stream(self.handle, c_void_p(stream_ptr)), "jstsp_set_stream")

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        check(self._lib.jstsp_synchronize(self.handle), "jstsp_synchronize")

    def workspace_bytes(self):
        return int(self._lib.jstsp_workspace_bytes(self.handle))

    def set_profiling(self, on):
        check(self._lib.jstsp_set_profiling(self.handle, int(bool(on))), "jstsp_set_profiling")

    def get_profile(self, kernel):
        n, ms = c_int(0), C.c_double(0.0)
        check(self._lib.jstsp_get_profile(self.handle, kernel.encode(), C.byref(n), C.byref(ms)),
              "jstsp_get_profile")
        return n.value, ms.value

    def last_fused_fallbacks(self):
        """Trials of the last proposed_algorithm call re-solved after a k-scale overflow in the fused pass."""
        n = C.c_int(0)
        check(self._lib.jstsp_last_fused_fallbacks(self.handle, C.byref(n)), "jstsp_last_fused_fallbacks")
        return int(n.value)

    def last_lanczos_mismatches(self):
        """Periodic cold checks of the last solve's warm-started lambda_max values that disagreed (2e-5 relative)."""
        n = C.c_int(0)
        check(self._lib.jstsp_last_lanczos_mismatches(self.handle, C.byref(n)), "jstsp_last_lanczos_mismatches")
        return int(n.value)

    def last_dictionary_block(self):
        """Block height of the block-Toeplitz structure the last proposed_algorithm call found in its dictionary (0: none)."""
        n = C.c_int(0)
        check(self._lib.jstsp_last_dictionary_block(self.handle, C.byref(n)), "jstsp_last_dictionary_block")
        return int(n.value)

    def last_conditioning(self):
        """(rcond_min, ns_residual_max) of the last call that (pseudo-)inverted a dictionary factor."""
        rc, res = C.c_double(1.0), C.c_double(0.0)
        check(self._lib.jstsp_last_conditioning(self.handle, C.byref(rc), C.byref(res)), "jstsp_last_conditioning")
        return rc.value, res.value

    def close(self):
        if self._h is not None:
            self._lib.jstsp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx
