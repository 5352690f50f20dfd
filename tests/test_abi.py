"""CPU: the C-ABI library builds/loads and exports exactly what include/ldt_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ldt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ldt_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from ldt_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(built):
    h = ctypes.CDLL(built.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(h, n), "missing export %s" % n


def test_binding_covers_header(built):
    assert sorted(built.SIGNATURES) == _header_functions()
    assert built.lib().ldt_abi_version() == built.ABI_VERSION


def test_plan_struct_layout_matches_header(built):
    # 8 int32 + (2 + 8*64 + 2) pointers + 3*64 pointers + 2 int32 + ptr + 2 int64 + 6 pointers + (fold ptr, int64, stats ptr) + 2 int32 + fold_monitor ptr
    assert ctypes.sizeof(built.ScorePlan) == 8 * 4 + (2 + 8 * built.MAX_BLOCKS + 2 + 3 * built.MAX_BLOCKS + 1 + 1 + 2 + 6 + 3 + 1 + 1) * 8


def test_argument_errors_are_reported_not_crashed(built):
    lib = built.lib()
    rc = lib.ldt_gemm_bf16(0, None, 0, None, 0, None, None, 0, None, 0, None, 0, None, 0, 0, None, 0, 1, 1, 64, None)
    assert rc == -1 and b"null" in lib.ldt_last_error()
    with pytest.raises(built.LdtHipError):
        built.check(rc, "ldt_gemm_bf16")
