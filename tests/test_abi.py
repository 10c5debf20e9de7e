"""CPU-only: the C-ABI library loads and exports exactly what include/ppbo_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ppbo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(ppbo_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_header_lists_entry_points():
    syms = header_symbols()
    assert "ppbo_gram" in syms and "ppbo_predict" in syms and "ppbo_fit_fmap" in syms
    assert len(syms) >= 19


def test_library_exports_every_declared_symbol():
    from ppbo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from ppbo_amd.build import build
        build(verbose=False)
    lib = _lib.load()
    for s in header_symbols():
        assert hasattr(lib, s), f"libppbo_hip.so does not export {s}"
        assert s in _lib.SIGNATURES, f"ctypes binding lacks {s}"
    assert set(_lib.SIGNATURES) == set(header_symbols())
    assert lib.ppbo_abi_version() == _lib.ABI_VERSION == 6


def test_library_exports_nothing_but_the_c_abi():
    """-fvisibility=hidden + the version script: the dynamic symbol table is the header's entry points and the HIP
    toolchain's per-TU registration ids (__hip_cuid_*) -- no mangled internals, no device stubs, no libstdc++ weak symbols."""
    import shutil
    import subprocess
    from ppbo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from ppbo_amd.build import build
        build(verbose=False)
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    exported = sorted(n for n in names if not n.startswith("__hip_"))
    assert exported == header_symbols(), sorted(set(exported) ^ set(header_symbols()))
    assert all(n.startswith("__hip_cuid_") for n in names if n.startswith("__hip_"))


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ppbo_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU path"):
        Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ppbo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# the oracle", ""), \
                    f"{f} mentions the oracle: the product path must not depend on it"


def test_ctx_create_without_gpu_returns_an_error_code():
    """No GPU in the build container: the C entry point must report it as a status code, not crash."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ppbo_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.ppbo_ctx_create(0, C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert lib.ppbo_ctx_destroy(None) == 0
    buf = C.create_string_buffer(16)
    assert lib.ppbo_last_error(None, buf, 16) != 0
