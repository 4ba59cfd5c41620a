"""The C-ABI library loads and exports every symbol include/pcs_hip.h declares; argument
validation works without a GPU (no compute calls here)."""
import ctypes
import re
from pathlib import Path

import pytest

from pycamset_amd import _capi

REPO = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (REPO / "include" / "pcs_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcs_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_symbols()
    assert len(names) >= 20
    assert set(names) == set(_capi.SYMBOLS), set(names) ^ set(_capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _capi.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), f"libpcs_hip.so does not export {name}"
    assert lib.pcs_version() >= 100
    assert lib.pcs_device_count() >= 0


def test_argument_errors_do_not_need_a_gpu():
    lib = _capi.lib()
    h = ctypes.c_void_p()
    assert lib.pcs_create(ctypes.byref(h), 7, 0, 3, 4, 8, 0) == _capi.PCS_ERR_ARG
    assert b"chain" in lib.pcs_last_error()
    assert lib.pcs_create(ctypes.byref(h), 0, 5, 3, 4, 8, 0) == _capi.PCS_ERR_ARG
    assert lib.pcs_create(ctypes.byref(h), 0, 0, 0, 4, 8, 0) == _capi.PCS_ERR_ARG
    assert lib.pcs_create(None, 0, 0, 3, 4, 8, 0) == _capi.PCS_ERR_ARG
    assert lib.pcs_eval(None, None, None, None) == _capi.PCS_ERR_ARG
    assert lib.pcs_destroy(None) == _capi.PCS_OK
    assert lib.pcs_n_params(None) == -1


def test_no_cpu_fallback_when_no_device():
    lib = _capi.lib()
    if lib.pcs_device_count() > 0:
        pytest.skip("a GPU is visible")
    from pycamset_amd.engine import Engine
    with pytest.raises(_capi.PcsError) as e:
        Engine("template", 3, 4, 8)
    assert e.value.code == _capi.PCS_ERR_NODEVICE


def test_product_package_never_imports_the_oracle():
    for p in (REPO / "pycamset_amd").rglob("*.py"):
        src = p.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("no CPU fallback", ""), p
    for p in (REPO / "pycamset_amd" / "csrc").iterdir():
        assert "ba_oracle" not in p.read_text(), p


def test_header_is_valid_c99_and_the_c_demo_links():
    """The boundary is a C ABI: the header must compile as plain C, and a C program must link against
    libpcs_hip.so without any C++ / torch symbol."""
    import subprocess
    inc = REPO / "include"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", str(inc / "pcs_hip.h")], check=True)
    out = REPO / "examples" / "_c_api_demo"
    try:
        subprocess.run(["gcc", "-std=c99", "-Wall", f"-I{inc}", str(REPO / "examples" / "c_api_demo.c"), "-o", str(out),
                        f"-L{REPO / 'pycamset_amd'}", "-lpcs_hip", f"-Wl,-rpath,{REPO / 'pycamset_amd'}", "-Wl,-rpath,/opt/rocm/lib",
                        "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
        lib = _capi.lib()
        if lib.pcs_device_count() == 0:   # CPU box: the program must fail cleanly with the no-device error
            res = subprocess.run([str(out)], capture_output=True, text=True)
            assert res.returncode == 1 and "no HIP device" in res.stderr
    finally:
        if out.exists():
            out.unlink()
