"""The C-ABI library loads and exports every symbol include/jstsp.h declares (no compute:
there is no GPU in the CPU test tier), the ctypes table mirrors the header, and the product
fails loudly — never silently falls back — without a GPU."""
import ctypes
import os
import re

import pytest

import jstsp19_amd
from jstsp19_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "jstsp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(jstsp_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert "jstsp_proposed_algorithm_c32" in names and "jstsp_correlate_c32" in names and len(names) >= 18
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -m jstsp19_amd.build"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "jstsp19_amd/_lib.py SIGNATURES out of sync with include/jstsp.h"


def test_version_and_error_string_without_gpu():
    lib = jstsp19_amd.load()
    assert b"gfx950" in lib.jstsp_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(jstsp19_amd.JstspError):
        jstsp19_amd.Context(0)
    import numpy as np
    with pytest.raises(jstsp19_amd.JstspError):
        jstsp19_amd.svt(np.eye(3, dtype=complex), 0.1)


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under jstsp19_amd/ may import it."""
    pkg = os.path.join(ROOT, "jstsp19_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_device_code_has_no_packed_fp32_instructions(tmp_path):
    """The library is built with -target-feature -packed-fp32-ops (jstsp19_amd/build.py, DESIGN.md section 5
    'Reproducibility': dependent v_pk_fma_f32 chains returned run-to-run different results beside MFMA-heavy waves on
    MI355X).  Checked on the code objects inside the shared library."""
    import shutil
    import subprocess
    from jstsp19_amd import build as B
    assert "-packed-fp32-ops" in B.FLAGS
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    lib = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], cwd=str(tmp_path), check=True, capture_output=True)
    devs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert devs, "no gfx950 code object found in the library"
    mfma = 0
    for f in devs:
        asm = subprocess.run([objdump, "-d", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
        mfma += "v_mfma_f32" in asm
        for op in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
            assert op not in asm, "%s found in %s" % (op, f)
    assert mfma >= 5        # the disassembly really is the kernels'


def test_every_environment_switch_the_library_reads_is_documented_in_the_header():
    """include/jstsp.h lists the JSTSP_* switches of the SHIPPED library: at most 15, every name that a getenv / env_int call of
    jstsp19_amd/csrc reads outside `#ifdef JSTSP_EXPERIMENTS` must be in that list, and every one must be toggled by a test.
    The switches of the experiments build (xp_getenv, or env_int inside `#ifdef JSTSP_EXPERIMENTS`) must be named in the
    header's "Experiments build" paragraph - and nowhere in tests/ (the tests run the shipped library)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shipped, xp = set(), set()
    pat = r'(?<![a-z_])(?:getenv|env_int|env_flt)\("(JSTSP_[A-Z0-9_]+)"'
    for f in glob.glob(os.path.join(root, "jstsp19_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "jstsp19_amd", "csrc", "*.h")):
        src = open(f).read()
        xp |= set(re.findall(r'xp_getenv\("(JSTSP_[A-Z0-9_]+)"', src))
        for blk in re.findall(r'#ifdef JSTSP_(?:EXPERIMENTS|FUSED_DBG_BUILD)\b(.*?)#endif', src, flags=re.S):
            xp |= set(re.findall(pat, blk))
        src = re.sub(r'#ifdef JSTSP_(?:EXPERIMENTS|FUSED_DBG_BUILD)\b.*?#endif', '', src, flags=re.S)
        shipped |= set(re.findall(pat, src))
    header = open(os.path.join(root, "include", "jstsp.h")).read()
    env = header[header.index("---- Environment"):header.index("---- kernel-level entry points")]
    main, exper = env.split("Experiments build.")
    assert 10 <= len(shipped) <= 15, sorted(shipped)
    assert not [n for n in shipped if ("  " + n) not in main], sorted(shipped)
    assert not [n for n in xp if n not in exper], sorted(xp)
    assert not (shipped & xp)
    tests = "".join(open(f).read() for f in glob.glob(os.path.join(root, "tests", "test_*.py")) if not f.endswith("test_capi_symbols.py"))
    assert not [n for n in shipped if n not in tests], [n for n in shipped if n not in tests]
    assert not [n for n in xp if n in tests], [n for n in xp if n in tests]

