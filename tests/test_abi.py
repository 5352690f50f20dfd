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


def test_no_export_outside_the_header(built):
    """Every `ldt_*` symbol the shared library exports is declared in include/ldt_hip.h (a leftover experiment object linked into the .so
    would show up here: VERDICT r4 item 9)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("ldt_")})
    assert exported == _header_functions(), sorted(set(exported) ^ set(_header_functions()))


def test_library_links_exactly_the_present_sources(built):
    """build.sh links one object per csrc/*.hip and nothing else (orphan objects are deleted, never linked)."""
    import glob
    csrc = os.path.join(ROOT, "ldt_amd", "csrc")
    srcs = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(csrc, "*.hip")))
    objs = sorted(os.path.basename(f)[:-2] for f in glob.glob(os.path.join(csrc, "build", "*.o")))
    assert objs == srcs


def _lint():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "ldt_amd", "csrc", "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_isa_lint_passes_on_the_built_library(built):
    """The hand-counted waits / asm loads / main-loop -> epilogue fence of the hot kernels are what was audited (csrc/isa_lint.py;
    build.sh runs the same check and fails the build on a violation)."""
    L = _lint()
    fps, errs, nk = L.analyse(built.LIB_PATH)
    assert not errs, errs[:5]
    gold = __import__("json").load(open(L.SIG_FILE))["kernels"]
    assert fps == gold and len(fps) > 50 and nk > 100
    assert L.analyse.tally.get("covered", 0) >= 100 and not L.analyse.tally.get("open") and not L.analyse.tally.get("violation")


def test_isa_lint_catches_what_it_is_for():
    """Negative controls on synthetic instruction streams: a copy of an asm-loaded register ahead of the wait, a wait that does not
    cover, a changed vmcnt immediate in the fingerprint, an LDS read between the last MFMA and the epilogue fence, a missing fence."""
    L = _lint()
    mk = lambda lines: [L.Ins(t.split()[0], t, 0x100 + 4 * i, False) for i, t in enumerate(lines)]
    ok = mk(["global_load_dwordx2 v[4:5], v[2:3], off", "global_load_lds_dwordx4 v[6:7], off", "s_waitcnt vmcnt(1)", "v_add_f32 v8, v4, v5"])
    t = {}
    assert L.check_asm_load_safety("k", ok, t, "global_load_dwordx2") == [] and t == {"covered": 1}
    moved = mk(["global_load_dwordx2 v[4:5], v[2:3], off", "v_mov_b32 v9, v4", "s_waitcnt vmcnt(0)"])
    assert len(L.check_asm_load_safety("k", moved, {}, "global_load_dwordx2")) == 1
    short = mk(["global_load_dwordx2 v[4:5], v[2:3], off", "global_load_lds_dwordx4 v[6:7], off", "s_waitcnt vmcnt(2)", "s_endpgm"])
    assert any("without a covering" in e for e in L.check_asm_load_safety("k", short, {}, "global_load_dwordx2"))
    # control flow: a load whose covering wait sits behind a loop's back edge is followed around the loop (covered: the second load of the
    # next trip is the one request behind it), and a destination named on ONE arm of a branch before that wait is found
    def loop(body):
        ins = mk(["s_mov_b32 s0, 4"] + body + ["s_sub_u32 s0, s0, 1", "s_cbranch_scc1 back", "s_waitcnt vmcnt(0)", "s_endpgm"])
        ins[-3].target = ins[1].addr                                     # back edge to the first instruction of the body
        return ins
    carried = loop(["s_waitcnt vmcnt(1)", "v_add_f32 v8, v4, v5", "global_load_dwordx2 v[4:5], v[2:3], off", "global_load_lds_dwordx4 v[6:7], off"])
    t = {}
    assert L.check_asm_load_safety("k", carried, t, "global_load_dwordx2") == [] and t == {"covered": 1}
    early = loop(["v_add_f32 v8, v4, v5", "s_waitcnt vmcnt(1)", "global_load_dwordx2 v[4:5], v[2:3], off", "global_load_lds_dwordx4 v[6:7], off"])
    assert len(L.check_asm_load_safety("k", early, {}, "global_load_dwordx2")) == 1
    arm = mk(["global_load_dwordx2 v[4:5], v[2:3], off", "s_cbranch_vccz skip", "v_mov_b32 v9, v5", "s_waitcnt vmcnt(0)", "s_endpgm"])
    arm[1].target = arm[3].addr
    assert len(L.check_asm_load_safety("k", arm, {}, "global_load_dwordx2")) == 1
    assert L.fingerprint(ok) == "L D W1" and L.fingerprint(short) == "L D W2"
    fence = ["v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"] + ["s_nop 15"] * 4 + ["ds_read_b128 v[0:3], v20"]
    assert L.check_mfma_c_hazard("k", mk(fence)) == []
    slipped = ["v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]", "ds_read_b128 v[0:3], v20"] + ["s_nop 15"] * 4
    assert len(L.check_mfma_c_hazard("k", mk(slipped))) == 1
    assert len(L.check_mfma_c_hazard("k", mk(fence[:1] + fence[5:]))) == 1
