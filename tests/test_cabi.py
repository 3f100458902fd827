"""CPU tests of the drop-in boundary: the C-ABI library loads and exports exactly what include/gingr_hip.h declares,
the ctypes prototypes cover every symbol, and the Python host layer fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gingr_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gingr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("gingr_ctx_create", "gingr_cpd_stats", "gingr_nn", "gingr_gauss_block", "gingr_model_upload",
                 "gingr_model_posterior_mean", "gingr_fitter_update_cpd_async", "gingr_fitter_update_icp_async",
                 "gingr_fitter_cpd_phase_async", "gingr_fitter_exchange", "gingr_last_error"):
        assert must in names
    assert len(names) >= 35


def test_library_exports_every_declared_symbol():
    from gingr_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in gingr_hip.h but not exported: {missing}"


def test_ctypes_prototypes_cover_the_header():
    from gingr_amd import _native
    assert sorted(_native.SIGNATURES) == declared_functions()
    _native.load()


def test_no_torch_or_cxx_types_in_the_boundary():
    raw = open(HEADER).read()
    assert 'extern "C"' in raw
    code = re.sub(r"/\*.*?\*/", "", raw, flags=re.S)     # declarations only, comments stripped
    for forbidden in ("torch", "std::", "hipStream_t", "at::", "jobject", "JNIEnv", "template", "class "):
        assert forbidden not in code, forbidden


def test_fails_loudly_without_a_gpu():
    import gingr_amd as ga
    from gingr_amd import _native
    lib = _native.load()
    if lib.gingr_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ga.GingrNativeError) as e:
        ga.Context(0)
    assert e.value.code == _native.ERR_NO_DEVICE
    h = ctypes.c_void_p()
    assert lib.gingr_ctx_create(0, ctypes.byref(h)) == _native.ERR_NO_DEVICE and not h.value


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gingr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include.*oracle|cpd_oracle|libcpd_oracle", src, flags=re.M), f


def test_header_is_self_contained_c():
    """The boundary header must compile on its own as plain C (what a cgo / JNI / FFI binding generator feeds on): catches
    C++-isms and declarations that use a type before its definition."""
    import shutil
    import subprocess
    import tempfile
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    hdr = os.path.join(ROOT, "include", "gingr_hip.h")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        with open(src, "w") as f:
            f.write('#include "gingr_hip.h"\nint main(void) { return (int)sizeof(gingr_state_scalars) == 0; }\n')
        subprocess.check_call([cc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.dirname(hdr), src])
