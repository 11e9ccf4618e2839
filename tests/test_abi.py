"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mpassit_amd.h
declares, fails loudly without a GPU (no CPU fallback), and the product never touches the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mpassit_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpg_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from mpassit_amd import _lib, build
    build.build()
    lib = _lib.load()
    decl = header_symbols()
    assert decl == sorted(_lib.SYMBOLS)
    for name in decl:
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (mpg_[a-z0-9_]+)", out))
    assert set(decl) <= exported


def test_code_object_is_gfx950_only():
    from mpassit_amd import _lib
    blob = open(_lib.SO_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_no_cpu_fallback():
    from mpassit_amd import _lib
    from mpassit_amd._lib import MpgError
    with pytest.raises(MpgError) as e:
        _lib.init(0)
    assert "no CPU fallback" in str(e.value)
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.mpg_mesh_create(ctypes.c_int64(1), ctypes.c_int64(1), ctypes.c_int(3), None, None, None, None, None, ctypes.byref(h))
    assert rc == 1  # MPG_ERR_NOT_INITIALIZED
    assert b"no CPU fallback" in lib.mpg_last_error()


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "mpassit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".F90", ".f90")):
                txt = open(os.path.join(dirpath, f)).read()
                bad = re.search(r"import oracle|from oracle|libmpassit_oracle|#include\s*[\"<][^\"\n>]*oracle|orc_[a-z_]+\s*\(", txt)
                assert bad is None, (os.path.join(dirpath, f), bad.group(0))
    for f in sorted(os.listdir(os.path.join(ROOT, "tools"))):          # the tools are no tests either: none of them may call the oracle
        if f.endswith((".py", ".sh", ".c", ".hip")) and f != "sanitize_cpu.sh":   # (sanitize_cpu.sh BUILDS the oracle with sanitizers for the tests it then runs)
            txt = open(os.path.join(ROOT, "tools", f)).read()
            bad = re.search(r"import oracle|from oracle|libmpassit_oracle|orc_[a-z_]+\s*\(", txt)
            assert bad is None, (f, bad.group(0))
    code = ("import sys; sys.path.insert(0, %r); import mpassit_amd, mpassit_amd.regrid, mpassit_amd.interp, mpassit_amd.dist, "
            "mpassit_amd.fields, mpassit_amd.target_grid, mpassit_amd.workloads; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_ncio_library_exports_every_declared_symbol():
    """include/mpassit_ncio.h (the nf90_* stand-ins either side of the hot path) vs hostio/libmpassit_ncio.so."""
    from mpassit_amd import build
    so = build.build_ncio()
    txt = open(os.path.join(ROOT, "include", "mpassit_ncio.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decl = sorted(set(re.findall(r"\b(ncio_[a-z0-9_]+)\s*\(", txt)))
    assert len(decl) >= 20 and "ncio_var_extent" in decl
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ncio_[a-z0-9_]+)", out))
    assert set(decl) <= exported, sorted(set(decl) - exported)
    lib = ctypes.CDLL(so)
    for name in decl:
        assert hasattr(lib, name), name
