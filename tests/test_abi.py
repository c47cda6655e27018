"""CPU: the C-ABI library loads and exports every symbol include/e4t_hip.h declares (no compute calls)."""
import os
import re

import pytest
import torch

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


def test_library_exports_nothing_but_the_declared_abi():
    """every dynamic `e4t_*` symbol of the library is declared in the header (helpers shared by the translation units are hidden)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _C.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("e4t_")}
    assert exported == header_symbols(), exported ^ header_symbols()


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


def test_descriptor_structs_match_the_header(tmp_path):
    """The ctypes mirrors of e4t_gemm_desc / e4t_conv_desc / e4t_wo_desc must have the C compiler's layout: compile the
    header with gcc and compare sizeof and every field offset."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("gcc not available")
    structs = {"e4t_gemm_desc": _C.GemmDesc, "e4t_conv_desc": _C.ConvDesc, "e4t_wo_desc": _C.WODesc, "e4t_gemm_plan_t": _C.GemmPlan}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "e4t_hip.h")}"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.strip()}
    for cname, cls in structs.items():
        assert got[(cname, "sizeof")] == C.sizeof(cls), (cname, got[(cname, "sizeof")], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU / PyTorch fallback: without libe4t_hip.so the binding raises, and so does the first use of the op backend."""
    from e4t import _C, ops
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", str(tmp_path / "libe4t_hip.so"))
    with pytest.raises(_C.E4TError, match="no CPU / PyTorch fallback"):
        _C.load()
    monkeypatch.setattr(ops, "_backend", None)
    with pytest.raises(_C.E4TError):
        ops.backend()
    from e4t.vae import VAEEncoder
    with pytest.raises(_C.E4TError):
        VAEEncoder(block_out_channels=(64, 64)).moments(torch.zeros(1, 3, 16, 16))


def test_integration_md_gemm_desc_example_has_the_c_layout(tmp_path):
    """INTEGRATION.md shows maintainers a ctypes mirror of e4t_gemm_desc.  A struct that is short of the trailing `colstats`
    pointer makes the library read 8 bytes past the caller's buffer (round-1 review): extract the documented class and hold it
    against the C compiler's sizeof / offsetof, field by field."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class GemmDesc\(C\.Structure\):.*?\n(    _fields_ = \[.*?\])\s*(?:#[^\n]*)?\nlib\.", md, flags=re.S)
    assert m, "INTEGRATION.md no longer contains the GemmDesc example"
    ns = {"C": C}
    exec("class GemmDesc(C.Structure):\n" + m.group(1), ns)
    doc = ns["GemmDesc"]
    assert [f for f, _ in doc._fields_] == [f for f, _ in _C.GemmDesc._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "e4t_hip.h")}"', "int main(void) {",
             '  printf("sizeof %zu\\n", sizeof(e4t_gemm_desc));']
    lines += [f'  printf("{f} %zu\\n", offsetof(e4t_gemm_desc, {f}));' for f, _ in doc._fields_]
    lines += ["  return 0;", "}"]
    (tmp_path / "l.c").write_text("\n".join(lines))
    subprocess.run(["gcc", "-std=c99", str(tmp_path / "l.c"), "-o", str(tmp_path / "l")], check=True)
    got = dict(l.split() for l in subprocess.run([str(tmp_path / "l")], check=True, capture_output=True, text=True).stdout.strip().split("\n"))
    assert int(got["sizeof"]) == C.sizeof(doc), (got["sizeof"], C.sizeof(doc))
    for f, _ in doc._fields_:
        assert int(got[f]) == getattr(doc, f).offset, f


def test_comm_entry_points_fail_cleanly_without_a_gpu():
    """e4t_comm_* (csrc/comm.hip): RCCL is resolved at first use, not linked; with no device the calls return an error code and a message"""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    lib = _C.load()
    buf = C.create_string_buffer(128)
    rc = lib.e4t_comm_unique_id(buf)          # torch's RCCL answers without a device, ROCm's own does not: either is fine
    assert rc == 0 or lib.e4t_last_error()
    if rc == 0:
        h0 = _C.vp()
        assert lib.e4t_comm_init(C.byref(h0), buf, 0, 1) < 0 and lib.e4t_last_error() and not h0.value
    assert lib.e4t_comm_unique_id(None) == -22
    h = _C.vp()
    assert lib.e4t_comm_init(C.byref(h), buf, 2, 2) == -22 and b"rank 2 of world 2" in lib.e4t_last_error()
    assert lib.e4t_comm_allreduce(None, None, 0, 0, 0, None) == -22
    assert lib.e4t_comm_wait(None, None) == -22
    assert lib.e4t_comm_destroy(None) == 0
    import subprocess, sys
    out = subprocess.run(["ldd", _C.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in out, out          # not a link dependency
