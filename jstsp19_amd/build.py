"""Build libjstsp_mi355x.so (hipcc, gfx950 only) in-tree under jstsp19_amd/csrc/.

Cross-compiles without a GPU.  Objects are rebuilt only when a source or header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# JSTSP_EXPERIMENTS=1: the experiments build (csrc/common.h: the switches of dropped experiments are read from the environment);
# its objects and library carry the suffix _xp and never replace the shipped ones
XP = os.environ.get("JSTSP_EXPERIMENTS") == "1"
LIB = os.path.join(CSRC, "libjstsp_mi355x_xp.so" if XP else "libjstsp_mi355x.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -packed-fp32-ops: no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in the device code.  Measured on MI355X (round 2):
# v_pk_fma_f32 with op_sel operand selection - what hipcc emits for complex multiplies - returns different bits from
# run to run while waves of an MFMA-heavy kernel share the SIMD (tools/probe/pk_fp32_probe.hip: 7 % of the results
# beside an MFMA loop, 0 alone; wait states do not help).  In this library it hit the Lanczos lambda_max kernel
# (tools/probe/lanczos_race.cpp: 5 of 3184 runs beside the three-Gram pass, 3041 beside an MFMA loop with barriers,
# 0 alone; 0 everywhere when built with this flag).  There is no switch for the op_sel forms alone; the flag costs
# nothing measurable.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] + (["-DJSTSP_EXPERIMENTS"] if XP else [])


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "jstsp.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# Translation units compiled WITH packed fp32 (round 6, tools/probe/pk_jacobi_race.py): none by default; JSTSP_PK_FP32_FILES=a.hip,b.hip
# for the experiment
PK_FP32_FILES = set(f for f in os.environ.get("JSTSP_PK_FP32_FILES", "").split(",") if f)


def _compile(src):
    obj = os.path.join(CSRC, src[:-4] + ("_xp.o" if XP else ".o"))
    path = os.path.join(CSRC, src)
    flags = [f for f in FLAGS if f not in ("-Xclang", "-target-feature", "-packed-fp32-ops")] if src in PK_FP32_FILES else FLAGS
    if _stale(obj, [path] + _headers()) or src in PK_FP32_FILES:
        cmd = [HIPCC] + flags + ["-x", "hip", "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        # (the host half of the compilation does not know the device feature and says so: not worth showing)
        err = "".join(l for l in r.stderr.splitlines(True) if "is not a recognized feature for this target" not in l)
        if err.strip():
            sys.stderr.write(err)
    return obj


def build(verbose=False, jobs=4):
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(verbose=True)
