"""CPU: the C-ABI library loads and exports every symbol include/e4t_hip.h declares (no compute calls)."""
import os
import re

from e4t import _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "e4t_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(e4t_[A-Za-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    lib = _C.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libe4t_hip.so does not export {s}"


def test_binding_table_covers_header():
    assert header_symbols() == set(_C.SIGNATURES), set(_C.SIGNATURES) ^ header_symbols()


def test_version_and_error_plumbing():
    lib = _C.load()
    assert lib.e4t_version() >= 100
    # argument validation happens on the host before any launch: safe to call without a GPU
    d = _C.GemmDesc()
    rc = lib.e4t_gemm_nt(d, None)
    assert rc == -22 and b"null operand" in lib.e4t_last_error()
    assert lib.e4t_wo_partial_floats(320, 320) == 10 * 320 * 2 + 10 * 320 + 10 * 320
