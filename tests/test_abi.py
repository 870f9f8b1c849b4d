"""CPU tests of the C-ABI boundary: libuniir_hip.so loads here (no GPU needed to dlopen) and exports every symbol
that include/uniir_hip.h declares; the ctypes table covers the header; error strings work."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "uniir_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(uniir_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from uniir_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/uniir_hip.h but not exported"


def test_ctypes_table_matches_header():
    from uniir_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()


def test_error_strings_and_version():
    from uniir_amd import _lib
    lib = _lib.load()
    assert lib.uniir_abi_version() >= 1
    assert lib.uniir_strerror(0) == b"ok"
    assert b"aligned" in lib.uniir_strerror(-3)


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from uniir_amd import ops
    with pytest.raises(RuntimeError):
        ops.call("uniir_cast_f32_to_bf16", torch.zeros(8), torch.zeros(8, dtype=torch.bfloat16), 8)
