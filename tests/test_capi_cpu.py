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


def test_library_exports_nothing_but_the_header(built_lib):
    """`nm -D --defined-only` ⊆ include/frostdb_amd.h: no C++ internals (fdb::Plan…, kernel launch stubs, libstdc++ template
    instantiations) in the dynamic symbol table — a cgo binary that links other C++ must not be able to collide with them."""
    import subprocess
    from frostdb_amd import build
    out = subprocess.run(["nm", "-D", "--defined-only", build.LIB], check=True, capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()})
    declared = declared_functions()
    extra = [n for n in exported if n not in declared]
    assert not extra, extra[:20]
    assert exported == declared, sorted(set(declared) - set(exported))
    # every prototype of the header carries the visibility macro (a new entry point without it would silently not be exported)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "frostdb_amd.h")).read(), flags=re.S)
    protos = re.findall(r"^(?:FDB_API )?(?:int|const char\*|void|int32_t|int64_t) (fdb_[a-z_0-9]+)\(", text, flags=re.M)
    marked = re.findall(r"^FDB_API (?:int|const char\*|void|int32_t|int64_t) (fdb_[a-z_0-9]+)\(", text, flags=re.M)
    assert sorted(protos) == sorted(marked) == declared


def test_header_is_plain_c99_and_links(built_lib, tmp_path):
    """The boundary is a C ABI: a translation unit that includes ONLY include/frostdb_amd.h compiles as strict C99 (what cgo's
    preamble is) and links against the library; it touches the handle-less entry points (no GPU needed)."""
    import subprocess
    from frostdb_amd import build
    src = tmp_path / "tu.c"
    src.write_text(r'''
#include "frostdb_amd.h"
#include <stdio.h>
#include <string.h>
int main(void) {
  fdb_plan_desc d;
  fdb_expr e;
  fdb_aggregation a;
  fdb_group_expr g;
  char buf[256];
  int64_t need = 0;
  int rc;
  memset(&d, 0, sizeof d); memset(&e, 0, sizeof e); memset(&a, 0, sizeof a); memset(&g, 0, sizeof g);
  e.op = FDB_OP_EQ; e.left = -1; e.right = -1; e.column = "labels.code";
  e.literal.type = FDB_LIT_STRING; e.literal.data = "200"; e.literal.len = 3;
  a.func = FDB_AGG_SUM; a.column = "value";
  g.name = "labels.path";
  d.filter = &e; d.n_filter = 1; d.filter_root = 0; d.aggs = &a; d.n_aggs = 1; d.groups = &g; d.n_groups = 1;
  rc = fdb_plan_explain(&d, buf, (int64_t)sizeof buf, &need);
  if (rc != FDB_OK) { fprintf(stderr, "explain: %d %s\n", rc, fdb_last_error()); return 1; }
  printf("%s|%s\n", fdb_version(), buf);
  return 0;
}
''')
    exe = tmp_path / "tu"
    inc = os.path.join(ROOT, "include")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, str(src), build.LIB,
                        "-Wl,-rpath," + os.path.dirname(build.LIB), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "gfx950" in out.stdout and "HashAggregate" in out.stdout, out.stdout


def test_version_and_error_strings(built_lib):
    built_lib.fdb_version.restype = ctypes.c_char_p
    assert b"gfx950" in built_lib.fdb_version()
    built_lib.fdb_last_error.restype = ctypes.c_char_p
    assert isinstance(built_lib.fdb_last_error(), bytes)


def test_descriptor_struct_layout_matches_header():
    # sizes and field offsets the C compiler gives the structs of the header vs the ctypes mirrors used by the binding
    import subprocess
    import tempfile
    from frostdb_amd import logicalplan as lp
    from frostdb_amd import physicalplan as pp
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "frostdb_amd.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(fdb_literal), sizeof(fdb_expr), sizeof(fdb_aggregation), sizeof(fdb_group_expr),
             sizeof(fdb_plan_desc), sizeof(fdb_proj_node), sizeof(fdb_projection), sizeof(fdb_parquet_chunk));
      printf("%zu %zu %zu %zu %zu %zu %zu\n", offsetof(fdb_parquet_chunk, name), offsetof(fdb_parquet_chunk, physical_type),
             offsetof(fdb_parquet_chunk, optional), offsetof(fdb_parquet_chunk, utf8), offsetof(fdb_parquet_chunk, codec),
             offsetof(fdb_parquet_chunk, data), offsetof(fdb_parquet_chunk, n_bytes));
      printf("%zu %zu %zu %zu %zu %zu\n", offsetof(fdb_proj_node, kind), offsetof(fdb_proj_node, op), offsetof(fdb_proj_node, left),
             offsetof(fdb_proj_node, right), offsetof(fdb_proj_node, column), offsetof(fdb_proj_node, literal));
      return 0;
    }
    '''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        lines = [[int(x) for x in ln.split()] for ln in subprocess.check_output([exe]).decode().splitlines()]
    assert lines[0] == [ctypes.sizeof(lp.CLiteral), ctypes.sizeof(lp.CExpr), ctypes.sizeof(lp.CAggregation), ctypes.sizeof(lp.CGroupExpr),
                        ctypes.sizeof(lp.CPlanDesc), ctypes.sizeof(lp.CProjNode), ctypes.sizeof(lp.CProjection), ctypes.sizeof(pp.ParquetChunk)]
    assert lines[1] == [getattr(pp.ParquetChunk, f).offset for f in ("name", "physical_type", "optional", "utf8", "codec", "data", "n_bytes")]
    assert lines[2] == [getattr(lp.CProjNode, f).offset for f in ("kind", "op", "left", "right", "column", "literal")]


def test_widening_of_narrow_result_indices(built_lib):
    """The host half of a big Finish: uint8 / uint16 indices → uint32, vector path and scalar head / tail, at every destination
    alignment and at sizes around the vector width."""
    import numpy as np
    rng = np.random.default_rng(12)
    for width, dt in ((1, np.uint8), (2, np.uint16), (4, np.uint32)):
        for n in (0, 1, 7, 31, 32, 33, 63, 64, 65, 1000, 4099):
            for shift in (0, 1, 3, 7):  # destination 4·shift bytes past a 64-byte boundary: exercises the alignment head
                src = rng.integers(0, np.iinfo(dt).max, n + 3, dtype=dt)[3:]  # (unaligned source too)
                raw = np.zeros(n + 32, dtype=np.uint32)
                base = (-raw.ctypes.data // 4) % 16
                dst = raw[base + shift: base + shift + n]
                guard_after = raw[base + shift + n: base + shift + n + 4].copy()
                rc = built_lib.fdb_selftest_widen(ctypes.c_void_p(src.ctypes.data), ctypes.c_int32(width), ctypes.c_void_p(dst.ctypes.data), ctypes.c_int64(n))
                assert rc == 0
                assert np.array_equal(dst, src.astype(np.uint32)), (width, n, shift)
                assert np.array_equal(raw[base + shift + n: base + shift + n + 4], guard_after)  # nothing written past the end
    # sub-byte transport: 2 / 4 bits per index, rows packed low bits first
    for bits in (2, 4):
        for n in (0, 1, 3, 4, 5, 15, 16, 17, 64, 255, 256, 1000, 4099):
            for shift in (0, 1, 2, 4, 12):  # (4, 12: 16-byte aligned, not 64: the vector path's scalar head)
                vals = rng.integers(0, 1 << bits, n, dtype=np.uint32)
                packed = np.zeros((n * bits + 7) // 8 + 1, dtype=np.uint8)
                for i, v in enumerate(vals):
                    packed[(i * bits) >> 3] |= np.uint8(int(v) << ((i * bits) & 7))
                raw = np.zeros(n + 32, dtype=np.uint32)
                base = (-raw.ctypes.data // 4) % 16
                dst = raw[base + shift: base + shift + n]
                rc = built_lib.fdb_selftest_widen(ctypes.c_void_p(packed.ctypes.data), ctypes.c_int32(-bits), ctypes.c_void_p(dst.ctypes.data), ctypes.c_int64(n))
                assert rc == 0
                assert np.array_equal(dst, vals), (bits, n, shift)
                assert not raw[base + shift + n: base + shift + n + 4].any()
                # … and through a rank → index table (width −12 / −14: the selftest's table is t[r] = 3 r + 5)
                raw[:] = 0
                rc = built_lib.fdb_selftest_widen(ctypes.c_void_p(packed.ctypes.data), ctypes.c_int32(-bits - 10), ctypes.c_void_p(dst.ctypes.data), ctypes.c_int64(n))
                assert rc == 0
                assert np.array_equal(dst, 3 * vals + 5), (bits, n, shift)
                assert not raw[base + shift + n: base + shift + n + 4].any()
    assert built_lib.fdb_selftest_widen(None, ctypes.c_int32(3), None, ctypes.c_int64(0)) == 1


def test_arrow_records_with_missing_tables_are_refused(built_lib):
    """Found by tools/arrow_fuzz.py: a string column that announced three buffers without a buffer table was dereferenced. Every
    defect the C data interface lets a consumer see — NULL tables behind non-zero counts, negative lengths, a dictionary schema
    without a dictionary array — now comes back as FDB_ERR_INVALID before anything is read."""
    import pyarrow as pa
    from frostdb_amd.arrow_c import ArrowArray, ArrowSchema, ExportedBatch
    rb = pa.RecordBatch.from_arrays([pa.array(["a", None, "ccc"]), pa.array([1, 2, 3], type=pa.int64()),
                                     pa.DictionaryArray.from_arrays(pa.array([0, 1, 0], type=pa.uint32()), pa.array([b"x", b"y"], type=pa.binary()))],
                                    names=["s", "v", "d"])

    def roundtrip(mutate):
        with ExportedBatch(rb) as ex:
            undo = mutate(ex)
            out, outs = ArrowArray(), ArrowSchema()
            rc = built_lib.fdb_arrow_roundtrip(ctypes.byref(ex.array), ctypes.byref(ex.schema), ctypes.byref(out), ctypes.byref(outs))
            undo()
            if rc == 0:
                pa.RecordBatch._import_from_c(ctypes.addressof(out), ctypes.addressof(outs))
            return rc

    assert roundtrip(lambda ex: (lambda: None)) == 0

    def field(path, name, value):
        def mutate(ex):
            st = path(ex)
            raw = ctypes.string_at(ctypes.addressof(st), ctypes.sizeof(st))  # (a pointer field read through ctypes is a VIEW of the struct)
            setattr(st, name, value)
            return lambda: ctypes.memmove(ctypes.addressof(st), raw, len(raw))
        return mutate

    child = lambda k: (lambda ex: ex.array.children[k].contents)
    for mut in (field(child(0), "buffers", None), field(child(1), "buffers", None), field(child(0), "length", -1), field(child(1), "offset", -3),
                field(child(2), "dictionary", None), field(child(0), "n_buffers", -1), field(lambda ex: ex.array, "children", None),
                field(lambda ex: ex.schema, "children", None), field(lambda ex: ex.array, "n_children", 9),
                field(lambda ex: ex.schema.children[2].contents, "dictionary", None)):
        assert roundtrip(mut) == 1


def test_descriptor_counts_and_arrays_must_agree(built_lib):
    """Found by tools/desc_fuzz.py: a projection without a node array under a dynamic aggregation was dereferenced by the copy the
    dynamic-aggregation family makes of the descriptor, before the plan's own validation saw it. Counts and arrays are now checked
    first, for every reader."""
    from frostdb_amd.logicalplan import CAggregation, CExpr, CGroupExpr, CPlanDesc, CProjNode, CProjection

    def explain(d):
        buf = ctypes.create_string_buffer(1024)
        need = ctypes.c_int64(0)
        return built_lib.fdb_plan_explain(ctypes.byref(d), buf, ctypes.c_int64(len(buf)), ctypes.byref(need))

    def base(dynamic):
        d = CPlanDesc()
        ag = (CAggregation * 1)()
        ag[0].func, ag[0].dynamic, ag[0].column = 1, dynamic, b"value"
        d.aggs, d.n_aggs = ctypes.cast(ag, ctypes.POINTER(CAggregation)), 1
        return d, [ag]

    for dynamic in (0, 1):
        d, keep = base(dynamic)
        assert explain(d) == 0
        pj = (CProjection * 1)()
        pj[0].name, pj[0].nodes, pj[0].n_nodes, pj[0].root = b"p", None, 1, 0  # nodes missing
        d.projections, d.n_projections = ctypes.cast(pj, ctypes.POINTER(CProjection)), 1
        assert explain(d) == 1
        nodes = (CProjNode * 1)()
        nodes[0].kind, nodes[0].column = 0, b"value"
        pj[0].nodes, pj[0].name = ctypes.cast(nodes, ctypes.POINTER(CProjNode)), None  # name missing
        assert explain(d) == 1
        pj[0].name, pj[0].n_nodes = b"p", -1
        assert explain(d) == 1
        for field, count in (("filter", "n_filter"), ("groups", "n_groups"), ("projections", "n_projections"), ("aggs", "n_aggs")):
            d, keep = base(dynamic)
            setattr(d, count, 3)        # a count without its array
            if field == "aggs":
                d.aggs = None
            assert explain(d) == 1, (field, dynamic)
            d, keep = base(dynamic)
            setattr(d, count, -1)       # a negative count
            assert explain(d) == 1, (field, dynamic)


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


def test_arrow_import_export_roundtrip_without_a_gpu(built_lib):
    """fdb_arrow_roundtrip: the host code behind push (column views at an offset, dictionaries of any index width, plain string
    columns encoded to distinct values + ids) and behind finish / filter (dictionary, plain string, bool, fixed-width output) —
    values, NULLs and types survive; dictionary indices come back as uint32."""
    import numpy as np
    import pyarrow as pa
    from frostdb_amd import physicalplan as pp
    rng = np.random.default_rng(5)
    n = 1003
    words = ["", "a", "é", "abc\x00d", "zeta"] + ["w%d" % k for k in range(50)]

    def strs(typ, null_frac):
        return pa.array([words[k] for k in rng.integers(0, len(words), n)], type=pa.string(), mask=rng.random(n) < null_frac).cast(typ)

    def dic(index_type, value_type, null_frac):
        idx = pa.array(rng.integers(0, 7, n), type=index_type, mask=rng.random(n) < null_frac)
        return pa.DictionaryArray.from_arrays(idx, pa.array(["v%d" % k for k in range(7)], type=pa.string()).cast(value_type))

    cols = {
        "i": pa.array(rng.integers(-9, 9, n), type=pa.int64(), mask=rng.random(n) < 0.1),
        "u": pa.array(rng.integers(0, 9, n).astype(np.uint64) << np.uint64(60), type=pa.uint64()),
        "f": pa.array(rng.normal(size=n), mask=rng.random(n) < 0.5),
        "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2),
        "s": strs(pa.string(), 0.1), "z": strs(pa.binary(), 0.0), "S": strs(pa.large_string(), 0.3), "Z": strs(pa.large_binary(), 1.0),
        "d8": dic(pa.int8(), pa.binary(), 0.1), "d16": dic(pa.uint16(), pa.string(), 0.0),
        "d32": dic(pa.uint32(), pa.large_string(), 0.2), "d64": dic(pa.int64(), pa.large_binary(), 0.05),
    }
    full = pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))
    for rec in (full, full.slice(0, 0), full.slice(13, 700), full.slice(1001, 2)):  # column offsets, byte-unaligned bitmaps, empty
        out = pp.arrow_roundtrip(rec)
        assert out.num_rows == rec.num_rows and out.schema.names == rec.schema.names
        for name in rec.schema.names:
            a, b = rec.column(name), out.column(name)
            if pa.types.is_dictionary(a.type):
                assert b.type.index_type == pa.uint32()
                assert pa.types.is_string(b.type.value_type) == (pa.types.is_string(a.type.value_type) or pa.types.is_large_string(a.type.value_type))
                a, b = a.dictionary_decode(), b.dictionary_decode()
                assert [None if v is None else (v.encode() if isinstance(v, str) else v) for v in a.to_pylist()] == \
                       [None if v is None else (v.encode() if isinstance(v, str) else v) for v in b.to_pylist()], name
            else:
                assert b.type == a.type, (name, a.type, b.type)
                assert a.to_pylist() == b.to_pylist(), name
            assert b.null_count == a.null_count, name
    with pytest.raises(pp.FdbError):
        pp.arrow_roundtrip(pa.RecordBatch.from_arrays([pa.array([[1], [2]])], names=["list"]))


def test_a_threads_recent_dictionaries_are_recognised_by_content_not_by_shape(built_lib):
    """read_dictionary asks the calling thread's recent dictionaries first, by two memcmps (raw offsets, bytes) instead of a content
    hash: records that follow each other with (a) the same dictionary in other buffers, (b) the same OFFSETS over different bytes,
    (c) the same BYTES cut differently, (d) a slice of a bigger dictionary (offsets that do not start at 0) must each decode to their
    own values, in any order and repeatedly."""
    import pyarrow as pa
    from frostdb_amd import physicalplan as pp

    def rec(values, typ=pa.binary()):
        idx = pa.array([i % len(values) for i in range(97)], type=pa.uint32())
        return pa.RecordBatch.from_arrays([pa.DictionaryArray.from_arrays(idx, pa.array(values, type=typ))], names=["labels.k"])

    a1 = rec([b"ab", b"c", b"def"])
    a2 = rec([bytes(b"ab"), bytes(b"c"), bytes(b"def")])          # same content, other buffers
    b = rec([b"xy", b"z", b"uvw"])                                 # same offsets, other bytes
    c = rec([b"a", b"bc", b"def"])                                 # same bytes, other offsets
    big = pa.array([b"pad", b"ab", b"c", b"def", b"tail"], type=pa.binary())
    d = pa.RecordBatch.from_arrays([pa.DictionaryArray.from_arrays(pa.array([i % 3 for i in range(97)], type=pa.uint32()), big.slice(1, 3))], names=["labels.k"])
    e = rec(["ab", "c", "def"], pa.string())                       # same bytes and offsets, utf8 instead of binary
    for r in (a1, a2, b, a1, c, d, b, e, c, a2, d, e, a1):
        out = pp.arrow_roundtrip(r)
        want = r.column(0).dictionary_decode().to_pylist()
        got = out.column(0).dictionary_decode().to_pylist()
        assert [v.encode() if isinstance(v, str) else v for v in got] == [v.encode() if isinstance(v, str) else v for v in want]
        assert pa.types.is_string(out.column(0).type.value_type) == pa.types.is_string(r.column(0).type.value_type)


def _explain_cases():
    from tests.golden import logictest_cases as G
    return G.EXPLAIN_CASES


@pytest.mark.parametrize("case", _explain_cases(), ids=[c["id"] for c in _explain_cases()])
def test_explain_strings_match_the_reference_plan_vectors(built_lib, case):
    """PhysicalPlan.Draw of the fused operators vs the reference's `explain` vectors (logictest/testdata/plan/…): the operator
    strings a FrostDB user sees for the part of the plan this library replaces. Host-only (fdb_plan_explain)."""
    from frostdb_amd import physicalplan as pp
    got = pp.explain(case["filter"], case["aggs"], case["groups"])
    assert got == case["expected"] + " [gfx950]", case["cite"]  # (the suffix names the executor, like the reference's "[concurrent]")


def test_explain_validates_like_create(built_lib):
    from frostdb_amd import physicalplan as pp
    from frostdb_amd.logicalplan import Col, Sum
    with pytest.raises(pp.FdbError):  # a regex std::regex cannot compile is rejected at plan build, without a device too
        pp.explain(Col("labels.x").RegexMatch("(unclosed"), [Sum(Col("value"))], [])


def test_dynamic_aggregations_explain_and_validation(built_lib):
    """`max(DynCol("foo"))` (Test_Aggregation_DynCol, aggregate_test.go:436-519): the concrete aggregations only come into being
    as records arrive (query/physicalplan/aggregate.go:306-336), so Draw lists the static ones (HashAggregate.Draw, :226-243);
    functions other than sum / min / max / count over a dynamic column set are refused."""
    from frostdb_amd import physicalplan as pp
    from frostdb_amd.logicalplan import Col, DynCol, Max, Sum, Unique
    assert pp.explain(None, [Max(DynCol("foo"))], []) == "HashAggregate ( by ) [gfx950]"
    assert pp.explain(Col("ts") > 1, [Max(DynCol("foo"))], [DynCol("labels")]) == "PredicateFilter (ts > 1) - HashAggregate ( by labels) [gfx950]"
    assert pp.explain(None, [Sum(Col("value")), Max(DynCol("foo"))], [Col("a")]) == "HashAggregate (sum(value) by a) [gfx950]"
    with pytest.raises(pp.UnsupportedError):
        pp.explain(None, [Unique(DynCol("foo"))], [])


def test_every_entry_point_has_a_typed_binding(built_lib):
    """The ctypes binding declares argument types for every function of the header that takes arguments, and a return type
    for every one that does not return int: an untyped call passes 64-bit addresses as C ints (it crashed once)."""
    from frostdb_amd.physicalplan import lib
    L = lib()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "frostdb_amd.h")).read(), flags=re.S)
    protos = re.findall(r"\b([A-Za-z_0-9 ]+?\**)\s*\b(fdb_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", text)
    assert len(protos) >= 40
    for ret, name, args in protos:
        fn = getattr(L, name)
        if args.strip() not in ("void", ""):
            assert fn.argtypes is not None, name
        ret = ret.strip()
        if ret.endswith("*") or "int64_t" in ret:
            assert fn.restype is not ctypes.c_int, name


def test_arrow_roundtrip_random_shapes(built_lib):
    """Randomised: column types, lengths 0 … 300, NULL densities 0 … 1, arbitrary slices (bit-unaligned validity and bool
    offsets, dictionary and string offsets), dictionaries with unused and duplicate entries."""
    import numpy as np
    import pyarrow as pa
    from frostdb_amd import physicalplan as pp
    rng = np.random.default_rng(20260924)

    def norm(a):
        if pa.types.is_dictionary(a.type):
            a = a.dictionary_decode()
        return [v.encode() if isinstance(v, str) else v for v in a.to_pylist()]

    for trial in range(150):
        n = int(rng.integers(0, 300))
        nf = float(rng.choice([0.0, 0.0, 0.1, 0.5, 1.0]))
        mask = rng.random(n) < nf
        kind = int(rng.integers(0, 8))
        if kind == 0:
            arr = pa.array(rng.integers(-2**62, 2**62, n), type=pa.int64(), mask=mask)
        elif kind == 1:
            arr = pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2), type=pa.uint64(), mask=mask)
        elif kind == 2:
            arr = pa.array(rng.normal(size=n), mask=mask)
        elif kind == 3:
            arr = pa.array(rng.random(n) < 0.5, mask=mask)
        elif kind in (4, 5):
            typ = [pa.string(), pa.binary(), pa.large_string(), pa.large_binary()][int(rng.integers(0, 4))]
            arr = pa.array(["x" * int(k) for k in rng.integers(0, 6, n)], type=pa.string(), mask=mask).cast(typ)
        else:
            it = [pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.int64(), pa.uint64()][int(rng.integers(0, 8))]
            vt = [pa.string(), pa.binary(), pa.large_string(), pa.large_binary()][int(rng.integers(0, 4))]
            values = pa.array(["d%d" % (k % 5) for k in range(9)], type=pa.string()).cast(vt)  # duplicates and unused entries
            arr = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 9, n), type=it, mask=mask), values)
        rec = pa.RecordBatch.from_arrays([arr, pa.array(np.arange(n), type=pa.int64())], names=["c", "row"])
        lo = int(rng.integers(0, n + 1))
        hi = int(rng.integers(lo, n + 1))
        for r in (rec, rec.slice(lo, hi - lo)):
            out = pp.arrow_roundtrip(r)
            assert norm(out.column("c")) == norm(r.column("c")), (trial, kind, lo, hi)
            assert out.column("row").to_pylist() == r.column("row").to_pylist()
            assert out.column("c").null_count == r.column("c").null_count


def test_local_communicator_endpoints_and_allocation_counters_without_a_gpu():
    """fdb_comm_init_local for ranks that share a device needs no HIP call: endpoints report their rank / size; the live-allocation
    counters (the leak check of the GPU suite) read zero in a process that allocated nothing."""
    from frostdb_amd import comm as fcomm
    from frostdb_amd import physicalplan as pp
    comms = fcomm.Comm.init_local([0, 0, 0])
    assert [c.rank for c in comms] == [0, 1, 2] and all(c.size == 3 for c in comms)
    for c in comms:
        c.close()
    assert pp.live_allocations() == {"device_blocks": 0, "device_bytes": 0, "pinned_blocks": 0}
    with pytest.raises(ValueError):
        fcomm.Comm(b"short", 2, 0, 0)


@pytest.mark.timeout(120)
def test_parquet_page_header_with_an_absurd_collection_is_refused_at_once():
    """Found by fuzzing the host parser: an unknown header field that is a list of 2^63 booleans (zero bytes each, had the reader
    believed it) kept the parsing thread busy for ever. Collection counts are now checked against what is left of the header."""
    from frostdb_amd import physicalplan as pp
    for elem in (0xF1, 0xF2, 0xFC):  # list<bool true>, list<bool false>, list<struct>
        bad = bytes([0xA9, elem]) + b"\xff" * 8 + b"\x7f" + b"\x00" * 32
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet([("x", pp.PARQUET_INT64, 0, False, bad, "UNCOMPRESSED")], 10)
        assert e.value.code == pp.FDB_ERR_INVALID
    bad = bytes([0xAB]) + b"\xff" * 8 + b"\x7f" + bytes([0x11]) + b"\x00" * 32  # map<bool, bool> with 2^63 pairs
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet([("x", pp.PARQUET_INT64, 0, False, bad, "UNCOMPRESSED")], 10)
    assert e.value.code == pp.FDB_ERR_INVALID


def test_parquet_chunks_are_parsed_and_refused_on_the_host():
    """fdb_batch_from_parquet reads page headers / run headers on the host BEFORE it touches a device: chunks it does not decode
    (INT32 / FLOAT columns, lists) or that are damaged (truncated bytes) come back as FDB_ERR_UNSUPPORTED / FDB_ERR_INVALID here, without
    a GPU; a well-formed chunk gets as far as the device call (FDB_ERR_DEVICE on this box)."""
    import numpy as np
    import pyarrow as pa
    from frostdb_amd import physicalplan as pp
    from tests.parquet_util import row_group_chunks, write_parquet
    n = 5000
    rng = np.random.default_rng(0)
    t = pa.table({"labels.a": pa.array([None if i % 9 == 0 else b"v%d" % (i % 13) for i in range(n)], type=pa.binary()),
                  "ts": pa.array(np.arange(n, dtype=np.int64)), "value": pa.array(rng.random(n), mask=rng.random(n) < 0.1)})
    good, rows = row_group_chunks(write_parquet(t), 0)
    assert rows == n and [c[1] for c in good] == [6, 2, 5]
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(good, rows)
    assert e.value.code == pp.FDB_ERR_DEVICE  # parsed fine, then no GPU
    # several row groups in one call (fdb_batches_from_parquet): the same order — every chunk of every row group is parsed first
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet_many([(good, rows), (good, rows)])
    assert e.value.code == pp.FDB_ERR_DEVICE
    cut = [good[0], (good[1][0], good[1][1], good[1][2], good[1][3], bytes(good[1][4][:100]), *good[1][5:]), good[2]]
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet_many([(good, rows), (cut, rows), (good, rows)])
    assert e.value.code == pp.FDB_ERR_INVALID, str(e.value)
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet_many([(good, rows), (good, rows + 1)])
    assert e.value.code == pp.FDB_ERR_INVALID, str(e.value)
    # every codec and string encoding a FrostDB schema can name (schema.proto:54-86) parses; what convert.go does not map either
    # is refused: INT32 / FLOAT physical types, repeated (list) columns
    for kw in (dict(compression="BROTLI"), dict(use_dictionary=False, column_encoding={"ts": "PLAIN", "labels.a": "DELTA_BYTE_ARRAY", "value": "PLAIN"}),
               dict(use_dictionary=False, column_encoding={"ts": "DELTA_BINARY_PACKED", "labels.a": "DELTA_LENGTH_BYTE_ARRAY", "value": "PLAIN"}, compression="ZSTD")):
        ok, rows = row_group_chunks(write_parquet(t, **kw), 0)
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet(ok, rows)
        assert e.value.code == pp.FDB_ERR_DEVICE, (kw, str(e.value))
    odd = pa.table({"i32": pa.array(np.arange(n, dtype=np.int32)), "f32": pa.array(rng.random(n).astype(np.float32)),
                    "lst": pa.array([[1, 2] if i % 2 else [] for i in range(n)], type=pa.list_(pa.int64()))})
    for name in ("i32", "f32", "lst"):
        bad, rows = row_group_chunks(write_parquet(odd.select([name])), 0)
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet(bad, rows)
        assert e.value.code == pp.FDB_ERR_UNSUPPORTED, (name, str(e.value))
    cut = [(nm, ty, opt, u8, data[: len(data) // 2]) for nm, ty, opt, u8, data, _ in good]
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(cut, rows)
    assert e.value.code == pp.FDB_ERR_INVALID
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(good, rows + 1)
    assert e.value.code == pp.FDB_ERR_INVALID
    # big compressed row groups are parsed on several host threads (one column chunk at a time each): the first failure in
    # column order comes back, whichever thread hit it
    nb = 400_000
    big = pa.table({"labels.a": pa.array([b"v%d" % (i % 97) for i in range(nb)], type=pa.binary()), "ts": pa.array(np.arange(nb, dtype=np.int64) * 1000),
                    "value": pa.array(rng.random(nb)), "other": pa.array(rng.random(nb))})
    chunks, rows = row_group_chunks(write_parquet(big, compression="SNAPPY"), 0)
    assert sum(len(c[4]) for c in chunks) > (1 << 20)
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(chunks, rows)
    assert e.value.code == pp.FDB_ERR_DEVICE, str(e.value)
    for victim in ("labels.a", "ts"):  # (the compressible chunks: damage lands on Snappy tags, not inside literals)
        hurt = []
        for nm, ty, opt, u8, data, cd in chunks:
            b = bytearray(data)
            if nm == victim:  # (0xFF = "copy with a 4-byte offset": offsets far outside what has been produced)
                for k in range(len(b) // 2, min(len(b) // 2 + 4096, len(b))):
                    b[k] = 0xFF
            hurt.append((nm, ty, opt, u8, bytes(b), cd))
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet(hurt, rows)
        assert e.value.code == pp.FDB_ERR_INVALID, victim
    # PLAIN BYTE_ARRAY pages (a writer without dictionary, or its dictionary fallback) are dictionary-encoded on the host
    plain, rows = row_group_chunks(write_parquet(t, use_dictionary=False), 0)
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(plain, rows)
    assert e.value.code == pp.FDB_ERR_DEVICE, str(e.value)
    short = [(nm, ty, opt, u8, data[: len(data) - 9] if nm == "labels.a" else data, cd) for nm, ty, opt, u8, data, cd in plain]
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(short, rows)
    assert e.value.code == pp.FDB_ERR_INVALID
    # DELTA_BINARY_PACKED INT64 pages: block / miniblock headers are walked on the host (the deltas are unpacked on the device)
    delta, rows = row_group_chunks(write_parquet(t, use_dictionary=["labels.a"], column_encoding={"ts": "DELTA_BINARY_PACKED"}), 0)
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(delta, rows)
    assert e.value.code == pp.FDB_ERR_DEVICE, str(e.value)
    short = [(nm, ty, opt, u8, data[: len(data) - 40] if nm == "ts" else data, cd) for nm, ty, opt, u8, data, cd in delta]
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet(short, rows)
    assert e.value.code == pp.FDB_ERR_INVALID
    # compressed pages are inflated on the host while the headers are walked: a sound chunk reaches the device call, a chunk
    # whose compressed bytes were damaged (or that names the wrong codec) is refused before it
    for codec in ("SNAPPY", "GZIP", "ZSTD", "LZ4"):
        for version in ("1.0", "2.0"):
            packed, rows = row_group_chunks(write_parquet(t, compression=codec, data_page_version=version), 0)
            assert {c[5] for c in packed} == {codec}
            with pytest.raises(pp.FdbError) as e:
                pp.ResidentBatch.from_parquet(packed, rows)
            assert e.value.code == pp.FDB_ERR_DEVICE, (codec, version, str(e.value))
        hurt = []
        for nm, ty, opt, u8, data, cd in packed:
            b = bytearray(data)
            for k in range(len(b) // 2, min(len(b) // 2 + 64, len(b))):
                b[k] ^= 0xA5
            hurt.append((nm, ty, opt, u8, bytes(b), cd))
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet(hurt, rows)
        assert e.value.code == pp.FDB_ERR_INVALID, codec
        wrong = [(nm, ty, opt, u8, data, "GZIP" if cd != "GZIP" else "ZSTD") for nm, ty, opt, u8, data, cd in packed]
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet(wrong, rows)
        assert e.value.code == pp.FDB_ERR_INVALID, codec


def _thrift_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _thrift_i32(field_delta, v):
    return bytes([(field_delta << 4) | 5]) + _thrift_varint((v << 1) ^ (v >> 31))


def _parquet_page(page_type, body, num_values, encoding):
    """A thrift-compact PageHeader (parquet.thrift) + body: type, sizes, then DataPageHeader (field 5) or DictionaryPageHeader (7)."""
    hdr = _thrift_i32(1, page_type) + _thrift_i32(1, len(body)) + _thrift_i32(1, len(body))
    if page_type == 0:  # DataPageHeader{num_values, encoding, definition_level_encoding = RLE, repetition_level_encoding = RLE}
        hdr += bytes([(2 << 4) | 12]) + _thrift_i32(1, num_values) + _thrift_i32(1, encoding) + _thrift_i32(1, 3) + _thrift_i32(1, 3) + b"\x00"
    else:               # DictionaryPageHeader{num_values, encoding}
        hdr += bytes([(4 << 4) | 12]) + _thrift_i32(1, num_values) + _thrift_i32(1, encoding) + b"\x00"
    return hdr + b"\x00" + body


def test_parquet_run_headers_that_overflow_and_empty_dictionaries_are_refused_on_the_host():
    """Two crafted BYTE_ARRAY RLE_DICTIONARY chunks (round-2 review): a bit-packed run whose varint header announces 2^59 groups of
    32-bit indices (groups × width wraps to 0 bytes, so the old length test passed and the device would have read n_rows × 32 bits
    that are not in the page; 2^60 groups of 16 bits turned the value count negative), and a chunk whose dictionary page is EMPTY
    while its data page has bit width 0 (every row = index 0 of nothing). Both must come back FDB_ERR_INVALID before any device call;
    the same pages with honest headers parse (FDB_ERR_DEVICE on this GPU-less box)."""
    import struct
    from frostdb_amd import physicalplan as pp
    n = 1000
    dict_page = _parquet_page(2, b"".join(struct.pack("<I", 2) + b"v%d" % i for i in range(4)), 4, 0)

    def chunk(bw, groups, payload, dpage=dict_page):
        body = bytes([bw]) + _thrift_varint((groups << 1) | 1) + payload
        return dpage + _parquet_page(0, body, n, 8)

    def code(data):
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet([("labels.a", pp.PARQUET_BYTE_ARRAY, 0, False, data, "UNCOMPRESSED")], n)
        return e.value.code

    assert code(chunk(32, 125, b"\x00" * (125 * 32))) == pp.FDB_ERR_DEVICE   # honest: 125 groups × 8 values
    assert code(chunk(32, 125, b"")) == pp.FDB_ERR_INVALID                   # honest header, payload missing
    assert code(chunk(32, 1 << 59, b"")) == pp.FDB_ERR_INVALID               # groups × 32 wraps to 0 bytes
    assert code(chunk(16, 1 << 60, b"")) == pp.FDB_ERR_INVALID               # groups × 8 turns negative
    assert code(chunk(32, 126, b"\x00" * (126 * 32))) == pp.FDB_ERR_INVALID  # more groups than the page has values
    empty_dict = _parquet_page(2, b"", 0, 0)
    assert code(empty_dict + _parquet_page(0, bytes([0]), n, 8)) == pp.FDB_ERR_INVALID  # bit width 0 into an empty dictionary
    assert code(dict_page + _parquet_page(0, bytes([0]), n, 8)) == pp.FDB_ERR_DEVICE     # bit width 0, index 0 exists


def test_gpu_local_cpus_helpers_say_so_when_there_is_no_gpu_to_ask():
    """physicalplan.local_cpus / pin_thread_near (where chain threads belong on a two-socket host): without a device whose PCI address can be read they
    return None / False and leave the calling thread's affinity alone."""
    import os
    from frostdb_amd import physicalplan as pp
    before = os.sched_getaffinity(0)
    cpus = pp.local_cpus(0)
    assert cpus is None or (isinstance(cpus, set) and cpus and all(isinstance(c, int) for c in cpus))
    if cpus is None:
        assert pp.pin_thread_near(0) is False
        assert os.sched_getaffinity(0) == before

