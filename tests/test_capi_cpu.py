"""CPU-side checks of the C ABI: the library builds, loads, and exports every symbol the header declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from frostdb_amd import build
    path = build.build()
    assert os.path.exists(path)
    return ctypes.CDLL(path)


def declared_functions():
    text = open(os.path.join(ROOT, "include", "frostdb_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(fdb_[a-z_0-9]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ["fdb_plan_create", "fdb_plan_push", "fdb_plan_push_batch", "fdb_plan_finish", "fdb_plan_merge",
                 "fdb_plan_filter", "fdb_plan_select", "fdb_plan_draw", "fdb_plan_close", "fdb_batch_import",
                 "fdb_plan_partial_keys", "fdb_plan_partial_state"]:
        assert must in names


def test_library_exports_every_declared_symbol(built_lib):
    missing = [n for n in declared_functions() if not hasattr(built_lib, n)]
    assert not missing, missing


def test_version_and_error_strings(built_lib):
    built_lib.fdb_version.restype = ctypes.c_char_p
    assert b"gfx950" in built_lib.fdb_version()
    built_lib.fdb_last_error.restype = ctypes.c_char_p
    assert isinstance(built_lib.fdb_last_error(), bytes)


def test_descriptor_struct_layout_matches_header():
    # sizes the C compiler gives the descriptor structs vs the ctypes mirrors used by the binding
    import subprocess
    import tempfile
    from frostdb_amd import logicalplan as lp
    src = r'''
    #include <stdio.h>
    #include "frostdb_amd.h"
    int main(void) { printf("%zu %zu %zu %zu %zu\n", sizeof(fdb_literal), sizeof(fdb_expr), sizeof(fdb_aggregation),
                            sizeof(fdb_group_expr), sizeof(fdb_plan_desc)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(lp.CLiteral), ctypes.sizeof(lp.CExpr), ctypes.sizeof(lp.CAggregation),
                     ctypes.sizeof(lp.CGroupExpr), ctypes.sizeof(lp.CPlanDesc)]


def test_invalid_descriptors_are_rejected_without_a_gpu(built_lib):
    # Descriptor validation happens before any HIP call, so these error paths are observable on CPU.
    from frostdb_amd.logicalplan import CExpr, CPlanDesc
    from frostdb_amd.physicalplan import lib
    L = lib()
    bad = CExpr(op=11, left=-1, right=-1, column=b"x")  # OpAdd: unsupported boolean expression (filter.go:162-164)
    arr = (CExpr * 1)(bad)
    d = CPlanDesc()
    d.filter = ctypes.cast(arr, ctypes.POINTER(CExpr))
    d.n_filter = 1
    d.filter_root = 0
    out = ctypes.c_void_p()
    rc = L.fdb_plan_create(ctypes.byref(d), 0, ctypes.byref(out))
    assert rc == 2 and b"unsupported boolean expression" in L.fdb_last_error()
