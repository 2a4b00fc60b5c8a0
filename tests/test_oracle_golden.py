"""Pins the CPU oracle (oracle/oracle.cpp) against the reference's own golden vectors.

Runs on CPU (-m "not gpu"). The same vectors drive the HIP path in tests/test_gpu_parity.py.
"""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import oracle
from oracle import OraclePlan
from frostdb_amd.logicalplan import Col, Count, Sum
from tests.golden import logictest_cases as G
from tests.util import batch_rows, fmt, parse_rows, record_from_rows, sort_key, table_records


def test_metrohash64_published_vectors():
    # MetroHash64 reference test vectors (metrohash testvector.h): key = 63 bytes "0123456789…012";
    # go-metro's Hash64 is a port of this function (dynparquet/hashed.go:207 calls it with seed 0).
    key = b"012345678901234567890123456789012345678901234567890123456789012"
    assert len(key) == 63
    assert oracle.metro_hash64(key, 0).to_bytes(8, "little") == bytes([0x6B, 0x75, 0x3D, 0xAE, 0x06, 0x70, 0x4B, 0xAD])
    assert oracle.metro_hash64(key, 1).to_bytes(8, "little") == bytes([0x3B, 0x0D, 0x48, 0x1C, 0xF4, 0xB9, 0xB8, 0xDF])


def test_hash_combine_matches_boost_formula():
    L = oracle.lib()
    lhs, rhs = 0x0123456789ABCDEF, 0xFEDCBA9876543210
    want = lhs ^ ((rhs + 0x9E3779B9 + ((lhs << 6) & (2**64 - 1)) + (lhs >> 2)) & (2**64 - 1))
    assert L.oracle_hash_combine(lhs, rhs) == want  # aggregate.go:245-247


def run_aggregate(case, nchains):
    plan = OraclePlan(case.get("filter"), case["aggs"], case["groups"], nchains=nchains)
    for rec in table_records(case["table"]):
        plan.push(rec)
    res = plan.finish().to_pydict()
    plan.close()
    return res


def check_aggregate(case, res):
    d = dict(res)
    if "avg_of" in case:  # the reference's AVG Projection (logicalplan/builder.go:205-238)
        s, c = case["avg_of"]
        d["avg"] = [(a // b if isinstance(a, int) else a / float(b)) for a, b in zip(d[s], d[c])]
    got = sorted([tuple(fmt(v) for v in row) for row in batch_rows(d, case["out"])], key=sort_key)
    want = sorted(case["expected"], key=sort_key)
    assert got == want, f"{case['cite']}: got {got} want {want}"


@pytest.mark.parametrize("nchains", [1, 2, 3])
@pytest.mark.parametrize("case", G.AGG_CASES, ids=[c["id"] for c in G.AGG_CASES])
def test_oracle_aggregate_golden(case, nchains):
    check_aggregate(case, run_aggregate(case, nchains))


def window_records(bucket):
    rec = table_records(G.WINDOW_TABLE)[0]
    ts = rec.column(rec.schema.get_field_index("timestamp"))
    b = pc.multiply(pc.divide(ts, pa.scalar(bucket, pa.int64())), pa.scalar(bucket, pa.int64()))
    return [rec.append_column("timestamp_bucket", b)]


@pytest.mark.parametrize("case", G.WINDOW_CASES, ids=[c["id"] for c in G.WINDOW_CASES])
def test_oracle_window_golden(case):
    plan = OraclePlan(None, case["aggs"], case["groups"], nchains=1)
    for rec in window_records(case["bucket"]):
        plan.push(rec)
    res = plan.finish().to_pydict()
    plan.close()
    got = sorted(batch_rows(res, case["out"]), key=sort_key)
    assert got == sorted(case["expected"], key=sort_key), case["cite"]


@pytest.mark.parametrize("case", G.FILTER_CASES, ids=[c["id"] for c in G.FILTER_CASES])
def test_oracle_filter_golden(case):
    rec = table_records(G.FILTER_TABLE)[0]
    plan = OraclePlan(case["filter"])
    out, idx = plan.filter(rec)
    assert list(idx) == case["rows"], case["cite"]
    if case["rows"]:
        d = out.to_pydict()
        assert d["timestamp"] == [r + 1 for r in case["rows"]]
        assert d["labels.label1"] == [b"value%d" % (r + 1) for r in case["rows"]]
    else:
        assert out is None  # filter.go:264-266: empty ⇒ nothing is pushed downstream
    plan.close()


@pytest.mark.parametrize("case", G.CONTAINS_CASES, ids=[c["id"] for c in G.CONTAINS_CASES])
def test_oracle_filter_contains_golden(case):
    """exec/filter/filter_contains: LIKE / NOT LIKE on a plain (non-dictionary) binary column, `=` on a UINT64 column."""
    from tests.util import bytes_schema_record
    rec = bytes_schema_record(G.CONTAINS_TABLE)
    plan = OraclePlan(case["filter"])
    out, idx = plan.filter(rec)
    assert list(idx) == case["rows"], case["cite"]
    d = out.to_pydict()
    for ci, name in enumerate(G.CONTAINS_TABLE["cols"]):
        assert d[name] == [G.CONTAINS_TABLE["rows"][r][ci] for r in case["rows"]], (case["cite"], name)
    plan.close()


BINARY_SCALAR_OPS = [("eq", lambda c: c == 4, 100_000), ("neq", lambda c: c != 4, 900_000), ("lt", lambda c: c < 4, 400_000),
                     ("le", lambda c: c <= 4, 500_000), ("gt", lambda c: c > 4, 500_000), ("ge", lambda c: c >= 4, 600_000)]


def binary_scalar_record():
    """The input of BenchmarkBinaryScalarOperation (binaryscalarexpr_test.go:15-41): 1 000 000 int64 values i % 10, compared with the
    scalar 4 under every comparison operator. The benchmark holds no expected output; the counts follow from the input."""
    import numpy as np
    import pyarrow as pa
    return pa.RecordBatch.from_arrays([pa.array(np.arange(1_000_000) % 10, type=pa.int64())], names=["v"])


@pytest.mark.parametrize("name,make,want", BINARY_SCALAR_OPS, ids=[o[0] for o in BINARY_SCALAR_OPS])
def test_oracle_binary_scalar_operation_shapes(name, make, want):
    from frostdb_amd.logicalplan import Col
    rec = binary_scalar_record()
    plan = OraclePlan(make(Col("v")))
    out, idx = plan.filter(rec)
    assert len(idx) == want and out.num_rows == want
    v = out.to_pydict()["v"]
    assert int(sum(v)) == {"eq": 4 * want, "neq": 41 * 100_000, "lt": 6 * 100_000, "le": 10 * 100_000, "gt": 35 * 100_000, "ge": 39 * 100_000}[name]
    plan.close()


def test_a_logical_operator_between_a_column_and_a_scalar_is_refused():
    """TestBinaryScalarOperationNotImplemented (binaryscalarexpr_test.go:97-106): OpAnd is not an operation between an array and a
    scalar — ErrUnsupportedBinaryOperation there, a refused expression at every layer here."""
    import frostdb_amd.logicalplan as lp
    from frostdb_amd.logicalplan import BinaryExpr, Col, Literal
    with pytest.raises(TypeError):
        OraclePlan(BinaryExpr(Col("v"), lp.OP_AND, Literal(4)))


def test_oracle_float_specials_follow_the_reference_loops():
    """NaN and ±Inf in a float64 column, as the reference's own loops treat them: comparisons are IEEE (Arrow's compare kernels,
    binaryscalarexpr.go:119-152: NaN satisfies only `!=`); SUM propagates NaN and −Inf + Inf is NaN; MIN / MAX start from the
    group's FIRST value and replace it on `<` / `>` (aggregate.go:847-860, :924-937) — so a NaN that comes first stays, a NaN
    later in the group is skipped. (The device orders NaN by bit pattern instead: DESIGN §5, deliberate differences.)"""
    import math
    import pyarrow as pa
    from frostdb_amd.logicalplan import Col, Count, Max, Min, Sum
    from tests.util import dict_array
    nan, inf = float("nan"), float("inf")
    rec = pa.RecordBatch.from_arrays([dict_array([b"a", b"a", b"a", b"b", b"b", b"b", b"c", b"c"]), pa.array([nan, 1.0, 2.0, 1.0, nan, 3.0, -inf, inf])], names=["k", "v"])
    o = OraclePlan(None, [Min(Col("v")), Max(Col("v")), Sum(Col("v")), Count(Col("v"))], [Col("k")])
    o.push(rec)
    d = o.finish().to_pydict()
    o.close()
    rows = {k: (mn, mx, sm, c) for k, mn, mx, sm, c in zip(d["k"], d["min(v)"], d["max(v)"], d["sum(v)"], d["count(v)"])}
    assert math.isnan(rows[b"a"][0]) and math.isnan(rows[b"a"][1]) and math.isnan(rows[b"a"][2]) and rows[b"a"][3] == 3
    assert rows[b"b"][0] == 1.0 and rows[b"b"][1] == 3.0 and math.isnan(rows[b"b"][2])
    assert rows[b"c"][0] == -inf and rows[b"c"][1] == inf and math.isnan(rows[b"c"][2])
    for filt, want in [(Col("v") > 1.5, [2, 5, 7]), (Col("v") != 1.0, [0, 2, 4, 5, 6, 7]), (Col("v") == 1.0, [1, 3]), (Col("v") <= inf, [1, 2, 3, 5, 6, 7]),
                       (Col("v") < 1.5, [1, 3, 6]), (Col("v") >= -inf, [1, 2, 3, 5, 6, 7])]:
        o = OraclePlan(filt)
        _, idx = o.filter(rec)
        assert list(idx) == want, str(filt)
        o.close()


@pytest.mark.parametrize("nchains", [1, 3])
def test_oracle_inconsistent_schema(nchains):
    from frostdb_amd.logicalplan import Col, Count, Max, Min, Sum
    spec = G.INCONSISTENT_SCHEMA
    recs = [record_from_rows(r["cols"], parse_rows(r["cols"], r["rows"])) for r in spec["records"]]
    fns = {"sum": [Sum], "min": [Min], "max": [Max], "count": [Count], "avg": [Sum, Count]}
    for name, want in spec["expected"].items():
        plan = OraclePlan(None, [f(Col("value")) for f in fns[name]], [Col("labels.label2")], nchains=nchains)
        for r in recs:
            plan.push(r)
        d = plan.finish().to_pydict()
        plan.close()
        if name == "avg":
            vals = [a // b for a, b in zip(d["sum(value)"], d["count(value)"])]
        else:
            vals = d[f"{name}(value)"]
        assert sorted(vals, reverse=True) == want, (spec["cite"], name)


def test_oracle_filter_contains_on_plain_binary():
    # logictest/testdata/exec/filter/filter_contains: schema "bytes" has a plain (non-dictionary) `value` column.
    from frostdb_amd.logicalplan import Col, UInt64
    from tests.util import dict_array
    rec = pa.RecordBatch.from_arrays(
        [dict_array([b"value1", b"value2", b"value3"]), pa.array([1, 2, 3], type=pa.uint64()),
         pa.array([b"foo", b"bar", b"baz"], type=pa.binary())],
        names=["labels.label1", "timestamp", "value"])
    p = OraclePlan(Col("value").Contains("a"))
    assert list(p.filter(rec)[1]) == [1, 2]  # filter_contains:15-19
    p = OraclePlan(Col("value").NotContains("a"))
    assert list(p.filter(rec)[1]) == [0]  # filter_contains:21-24
    p = OraclePlan(Col("timestamp") == UInt64(2))
    assert list(p.filter(rec)[1]) == [1]  # filter_contains:10-13


# ---- pre-aggregate Projection (project.go:73-399) pinned on the reference's math and window vectors --------------------

def run_math_case(make_plan, case):
    """`sum(<expr>) group by timestamp` (one group per row ⇒ per-row values) and, for NULL-ness, `count(value) group by <expr>`."""
    recs = table_records(G.MATH_TABLE)
    ts_order = [1, 3, 5, 11]
    expected = case["expected"]
    plan = make_plan(None, [Sum(case["expr"])], [Col("timestamp")])
    d = plan(recs)
    name = f"sum({case['expr'].name})"
    by_ts = dict(zip(d["timestamp"], d[name]))
    assert [by_ts[t] for t in ts_order] == [0 if e is None else e for e in expected], case["cite"]  # a NULL adds the builder's zero slot
    # NULL-ness of the computed value as a group KEY. Group identity in the reference is the hash alone, and both NULL and
    # the int64 value 0 hash to 0 (dynparquet/hashed.go:254-272 + the "skip zero hashes" fold, aggregate.go:398-409), so a
    # key of 0 and a NULL key are ONE group whose printed key is whichever row arrived first. Row by row (filter
    # timestamp == t ⇒ one row ⇒ "first" is unambiguous) the key must be exactly the golden value, NULL included.
    alias = case["expr"].Alias("q")
    for t, e in zip(ts_order, expected):
        d = make_plan(Col("timestamp") == t, [Count(Col("value"))], [alias])(recs)
        assert list(zip(d["q"], d["count(value)"])) == [(e, 1)], (case["cite"], t)
    d = make_plan(None, [Count(Col("value"))], [alias])(recs)
    got = sorted(((0 if k is None else k), n) for k, n in zip(d["q"], d["count(value)"]))
    want = {}
    for e in expected:
        want[e or 0] = want.get(e or 0, 0) + 1
    assert got == sorted(want.items()), case["cite"]


def _oracle_runner(filter_expr, aggs, groups):
    def run(recs):
        plan = OraclePlan(filter_expr, aggs, groups, nchains=1)
        for r in recs:
            plan.push(r)
        d = plan.finish().to_pydict()
        plan.close()
        return d
    return run


@pytest.mark.parametrize("case", G.MATH_CASES, ids=[c["id"] for c in G.MATH_CASES])
def test_oracle_math_projection_golden(case):
    run_math_case(_oracle_runner, case)


@pytest.mark.parametrize("case", G.WINDOW_CASES, ids=[c["id"] for c in G.WINDOW_CASES])
def test_oracle_window_golden_with_fused_projection(case):
    """The window vectors again, but `(timestamp/bucket)*bucket as timestamp_bucket` is evaluated by the operator
    (plan/aggregate/window: `Projection (value, timestamp / 1000 * 1000 as timestamp_bucket) - HashAggregate (… by timestamp_bucket)`)."""
    bucket = (Col("timestamp") / case["bucket"] * case["bucket"]).Alias("timestamp_bucket")
    groups = [bucket if g.name == "timestamp_bucket" else g for g in case["groups"]]
    res = _oracle_runner(None, case["aggs"], groups)(table_records(G.WINDOW_TABLE))
    got = sorted(batch_rows(res, case["out"]), key=sort_key)
    assert got == sorted(case["expected"], key=sort_key), case["cite"]


def test_oracle_uint64_arithmetic_known_answers():
    """AddUint64s / SubUint64s / MulUint64s / DivUint64s (project.go:335-395) restated: hand-computed rows — wrap-around modulo 2^64,
    an unsigned quotient beyond 2^63, NULL for a zero divisor, no NULL propagation (the loops read raw slots). No vector of the
    reference covers these four functions: this pins the restatement on their text alone (parity unpinned, as DESIGN §8 says)."""
    from frostdb_amd.logicalplan import UInt64
    M = 2**64
    a = [5, 2**63 + 4, M - 1, 0, 7]
    b = [2, 4, 3, 9, 0]
    rec = pa.RecordBatch.from_arrays([pa.array(a, type=pa.uint64()), pa.array(b, type=pa.uint64()), pa.array([1, 2, 3, 4, 5], type=pa.int64())], names=["a", "b", "value"])
    A, B = Col("a"), Col("b")
    for expr, want in (((A + B).Alias("k"), [(x + y) % M for x, y in zip(a, b)]),
                       ((A - B).Alias("k"), [(x - y) % M for x, y in zip(a, b)]),
                       ((A * B).Alias("k"), [(x * y) % M for x, y in zip(a, b)]),
                       ((A / B).Alias("k"), [x // y if y else None for x, y in zip(a, b)]),
                       ((B - UInt64(3)).Alias("k"), [(y - 3) % M for y in b])):
        res = _oracle_runner(None, [Sum(Col("value"))], [expr])([rec])
        got = dict(zip(res["k"], res["sum(value)"]))
        exp = {}
        for k, v in zip(want, [1, 2, 3, 4, 5]):  # (a key of 0 and a NULL key hash alike in the reference: one group)
            exp[0 if k is None else k] = exp.get(0 if k is None else k, 0) + v
        assert {(0 if k is None else k): v for k, v in got.items()} == exp, expr.name
    with pytest.raises(Exception):
        _oracle_runner(None, [Sum(A + B)], [Col("value")])([rec])        # ErrUnsupportedSumType (aggregate.go:736)
    with pytest.raises(Exception):
        _oracle_runner(None, [Sum(Col("value"))], [(A + 1).Alias("k")])([rec])  # an int64 scalar next to a uint64 array: the type assertion at project.go:139-147


def test_config1_simple_schema_known_answer():
    """BASELINE.json configs[0]: examples/simple schema, 10 k rows, `names.first_name == 'Frederic'` + SUM(value) — the
    reference's own CPU-runnable case (examples/simple/simple.go:66-75 filters exactly this). Known answer from numpy."""
    import numpy as np
    from tests.util import make_simple_batches
    batches = make_simple_batches(np.random.default_rng(1), 10_000, 3)
    want_sum, want_rows, by_surname = 0, 0, {}
    for b in batches:
        first = b.column(0).to_pylist()
        sur = b.column(1).to_pylist()
        val = b.column(b.schema.get_field_index("value")).to_pylist()
        for f, s, v in zip(first, sur, val):
            if f == "Frederic":
                want_sum += v; want_rows += 1
                by_surname[s] = by_surname.get(s, 0) + v
    run = _oracle_runner(Col("names.first_name") == "Frederic", [Sum(Col("value")), Count(Col("value"))], [])
    d = run(batches)
    assert d["sum(value)"] == [want_sum] and d["count(value)"] == [want_rows]
    d = _oracle_runner(Col("names.first_name") == "Frederic", [Sum(Col("value"))], [Col("names.surname")])(batches)
    got = {k.decode() if isinstance(k, bytes) else k: v for k, v in zip(d["names.surname"], d["sum(value)"])}
    assert got == by_surname


@pytest.mark.parametrize("nchains", [1, 2])
@pytest.mark.parametrize("case", G.DISTINCT_CASES, ids=[c["id"] for c in G.DISTINCT_CASES])
def test_oracle_distinct_golden(case, nchains):
    """Distinction (distinct.go:72-170) = a plan without aggregations; per chain, then Synchronizer + a final Distinction."""
    recs = table_records(G.DISTINCT_TABLE)
    plan = OraclePlan(case["filter"], [], case["groups"], nchains=nchains)
    for i, r in enumerate(recs):
        plan.push(r, chain=i % nchains)
    d = plan.finish().to_pydict()
    plan.close()
    got = sorted(batch_rows(d, case["out"]), key=sort_key)
    assert got == sorted(case["expected"], key=sort_key), case["cite"]


@pytest.mark.parametrize("indices,expected", [
    ([4, 6, 7, 8, 10], [(4, 5), (6, 9), (10, 11)]),                       # filter_test.go:10-20 TestBuildIndexRanges
    ([1, 3, 5, 7, 9], [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10)]),         # filter_test.go:27-30 "no consecutive"
    ([1, 2], [(1, 3)]),                                                   # :31-34 "only consecutive"
    ([1], [(1, 2)]),                                                      # :35-38 "only 1"
    ([1, 2, 7, 8, 9], [(1, 3), (7, 10)]),                                 # :39-42 "multiple"
])
def test_oracle_build_index_ranges_golden(indices, expected):
    assert oracle.build_index_ranges(indices) == expected


def test_oracle_and_short_circuits_like_the_reference():
    """filter_test.go:66-82 TestAndExprShortCircuits: when the left side of an AND selects nothing the right side is not evaluated —
    observable as "no error" for a right side that would raise (here: `<` on a dictionary column, binaryscalarexpr.go:106-108)."""
    from frostdb_amd.logicalplan import And
    rec = table_records(G.FILTER_TABLE)[0]
    bad = Col("labels.label1") < "x"
    plan = OraclePlan(bad)
    with pytest.raises(Exception):
        plan.filter(rec)
    plan.close()
    plan = OraclePlan(And(Col("labels.label1") == "no such value", bad))
    out, idx = plan.filter(rec)
    assert out is None and list(idx) == []
    plan.close()


# ---- UniqueAggregation / AndAggregation (aggregate.go:635-732), vectors of query/engine_test.go ------------------------------

def unique_and_cases():
    import pyarrow as pa
    from frostdb_amd.logicalplan import AndAgg, Unique
    uniq = pa.RecordBatch.from_arrays([pa.array([1, 2, 3], pa.int64()), pa.array([1, 1, 3], pa.int64())], names=["example", "timestamp"])
    andr = pa.RecordBatch.from_arrays([pa.array([True, False, True, True]), pa.array([1, 1, 3, 3], pa.int64())], names=["example", "timestamp"])
    return [
        # engine_test.go:19-74 TestUniqueAggregation: timestamps [1, 3]; unique(example) NULL for ts 1 (values 1 and 2), 3 for ts 3
        dict(id="unique", cite="query/engine_test.go:19-74", rec=uniq, agg=Unique(Col("example")), out="unique(example)", expected={1: None, 3: 3}),
        # engine_test.go:76-131 TestAndAggregation: and(example) false for ts 1 (true, false), true for ts 3
        dict(id="and", cite="query/engine_test.go:76-131", rec=andr, agg=AndAgg(Col("example")), out="and(example)", expected={1: False, 3: True}),
    ]


@pytest.mark.parametrize("nchains", [1, 2])
@pytest.mark.parametrize("case", unique_and_cases(), ids=lambda c: c["id"])
def test_oracle_unique_and_golden(case, nchains):
    plan = OraclePlan(None, [case["agg"]], [Col("timestamp")], nchains=nchains)
    rec = case["rec"]
    if nchains == 1:
        plan.push(rec)
    else:  # the rows of one group land on different chains: the final stage must still see them as one group
        plan.push(rec.slice(0, 1), chain=0)
        plan.push(rec.slice(1), chain=1)
    d = plan.finish().to_pydict()
    plan.close()
    assert dict(zip(d["timestamp"], d[case["out"]])) == case["expected"], case["cite"]


@pytest.mark.parametrize("case", G.DISTINCT_PROJ_CASES, ids=[c["id"] for c in G.DISTINCT_PROJ_CASES])
def test_oracle_distinct_bool_projection_golden(case):
    """distinct over a boolean projection (boolExprProjection, project.go:401-470): every row gets a valid bool key."""
    d = _oracle_runner(None, [], case["groups"])(table_records(case["table"]))
    # `timestamp 0` is an int64 key of 0: printed as 0 or NULL depending on which row came first (hash identity), fold for comparison
    rows = [tuple(0 if (v is None and c == "timestamp") else v for c, v in zip(case["out"], r)) for r in batch_rows(d, case["out"])]
    assert sorted(rows, key=sort_key) == sorted(case["expected"], key=sort_key), case["cite"]


def bool_table_record():
    import pyarrow as pa
    from tests.util import dict_array
    return pa.RecordBatch.from_arrays([dict_array([n for n, _ in G.BOOL_TABLE_ROWS]), pa.array([f for _, f in G.BOOL_TABLE_ROWS], type=pa.bool_())],
                                      names=["name", "found"])


def test_oracle_projection_vectors_on_this_path():
    """exec/projection/math_projection:17-21 (a grouped sum over a computed column) and exec/projection/bool:10-14 (a boolean
    column compared with a boolean literal)."""
    from oracle import OraclePlan
    c = G.PROJ_MATH_GROUPED
    d = _oracle_runner(None, c["aggs"], c["groups"])(table_records(G.PROJ_MATH_TABLE))
    assert sorted(batch_rows(d, c["out"]), key=sort_key) == sorted(c["expected"], key=sort_key), c["cite"]
    rec = bool_table_record()
    for case in G.BOOL_FILTER_CASES:
        o = OraclePlan(case["filter"])
        try:
            _, idx = o.filter(rec)
            assert list(idx) == case["rows"], case["cite"]
        finally:
            o.close()


def duration_record():
    import pyarrow as pa
    from tests.util import dict_array
    rows = G.DURATION_CASE["rows"]
    return pa.RecordBatch.from_arrays([pa.array([r[0] for r in rows], type=pa.int64()), dict_array([r[1] for r in rows]),
                                       pa.array([r[2] for r in rows], type=pa.int64())], names=["timestamp", "stacktrace", "value"])


def agg_projection_records():
    return [record_from_rows(r["cols"], parse_rows(r["cols"], r["rows"])) for r in G.AGG_PROJECTION_CASE["records"]]


def check_root_aggregate_cases(make_plan):
    c = G.DURATION_CASE
    d = make_plan(None, c["aggs"], c["groups"])([duration_record()])
    assert sorted(batch_rows(d, c["out"]), key=sort_key) == sorted(c["expected"], key=sort_key), c["cite"]
    c = G.AGG_PROJECTION_CASE
    d = make_plan(None, c["aggs"], c["groups"])(agg_projection_records())
    assert sorted(d.keys()) == sorted(c["fields"]), c["cite"]
    assert sorted(batch_rows(d, c["out"]), key=sort_key) == sorted(c["expected"], key=sort_key), c["cite"]


def test_oracle_root_aggregate_tests():
    """TestDurationAggregation and TestAggregationProjection (root aggregate_test.go) restated as known answers."""
    check_root_aggregate_cases(_oracle_runner)


def test_oracle_dynamic_column_aggregation():
    """Test_Aggregation_DynCol (root aggregate_test.go:436-519): one record per column of the dynamic set plus one with all three,
    `max(DynCol("foo"))`, no grouping → 3 columns, 1 row (the test's own assertion), with the values a max must have. And the
    reference's limits, restated as errors: a record lacking an aggregated column that creates a new group (nil dereference,
    aggregate.go:413-417); a record with no column of the set (:366-380)."""
    from frostdb_amd.logicalplan import DynCol, Max
    one = lambda name, v: pa.RecordBatch.from_arrays([pa.array([v], type=pa.int64())], names=[name])  # noqa: E731
    recs = [one("foo.bar", 7), one("foo.baz", 9), one("foo.bah", 3),
            pa.RecordBatch.from_arrays([pa.array([5]), pa.array([11]), pa.array([1])], names=["foo.bar", "foo.baz", "foo.bah"])]
    for nchains in (1, 2, 3):
        d = _oracle_runner_n(None, [Max(DynCol("foo"))], [], nchains)(recs)
        assert len(d) == 3 and all(len(v) == 1 for v in d.values())
        assert d == {"max(foo.bar)": [7], "max(foo.baz)": [11], "max(foo.bah)": [3]}
    with pytest.raises(Exception, match="not found"):
        _oracle_runner_n(None, [Max(DynCol("foo"))], [], 1)([one("other", 1)])
    grouped = [pa.RecordBatch.from_arrays([pa.array([1, 2]), pa.array([10, 20])], names=["k", "foo.a"]),
               pa.RecordBatch.from_arrays([pa.array([3]), pa.array([30])], names=["k", "foo.b"])]  # new group 3 without foo.a
    with pytest.raises(Exception, match="panic"):
        _oracle_runner_n(None, [Max(DynCol("foo"))], [Col("k")], 1)(grouped)


def _oracle_runner_n(filter_expr, aggs, groups, nchains):
    def run(recs):
        plan = OraclePlan(filter_expr, aggs, groups, nchains=nchains)
        try:
            for r in recs:
                plan.push(r)
            return plan.finish().to_pydict()
        finally:
            plan.close()
    return run


@pytest.mark.parametrize("seed", range(6))
def test_oracle_agrees_with_pyarrow_on_the_quirk_free_cases(seed):
    """Independent cross-check: random filter + group-by + SUM / MIN / MAX / COUNT through the oracle and through
    pyarrow's own compute kernels (`pc.*` comparisons, `Table.group_by().aggregate`). Only where the reference has no quirk:
    aggregated columns without NULLs (row 19: MIN / MAX read a NULL as 0), no int64 key of 0 (≡ NULL), COUNT = rows."""
    from frostdb_amd.logicalplan import And, Max, Min, Or
    from tests.util import dict_array
    rng = np.random.default_rng(4400 + seed)
    n = 20_000
    code = dict_array([None if rng.random() < 0.05 else b"c%d" % k for k in rng.integers(0, 6, n)])
    path = dict_array([None if rng.random() < 0.1 else b"p%02d" % k for k in rng.integers(0, 30, n)])
    bucket = pa.array(rng.integers(1, 9, n) * 100, type=pa.int64())
    value = pa.array(rng.integers(-1000, 1000, n), type=pa.int64())
    fval = pa.array(rng.uniform(-10, 10, n))
    rec = pa.RecordBatch.from_arrays([code, path, bucket, value, fval], names=["labels.code", "labels.path", "bucket", "value", "fval"])
    code_s, path_s = code.dictionary_decode(), path.dictionary_decode()
    filters = [
        (Col("labels.code") == "c1", pc.fill_null(pc.equal(code_s, pa.scalar(b"c1")), False)),
        (And(Col("value") > 0, Col("fval") <= 2.5), pc.and_(pc.greater(value, 0), pc.less_equal(fval, 2.5))),
        (Or(Col("labels.code") != "c2", Col("bucket") >= 500), pc.or_(pc.fill_null(pc.not_equal(code_s, pa.scalar(b"c2")), False), pc.greater_equal(bucket, 500))),
        (Col("labels.path") == None, pc.is_null(path_s)),  # noqa: E711
    ]
    fexpr, mask = filters[seed % len(filters)]
    groups, keys = [([Col("labels.path")], ["labels.path"]), ([Col("labels.path"), Col("bucket")], ["labels.path", "bucket"]), ([Col("bucket")], ["bucket"])][seed % 3]
    aggs = [Sum(Col("value")), Min(Col("value")), Max(Col("fval")), Count(Col("value")), Sum(Col("fval"))]
    got = _oracle_runner_n(fexpr, aggs, groups, 1 + seed % 3)([rec.slice(0, 7000), rec.slice(7000)])
    t = pa.table({"labels.path": path_s, "bucket": bucket, "value": value, "fval": fval}).filter(mask)
    want = t.group_by(keys, use_threads=False).aggregate([("value", "sum"), ("value", "min"), ("fval", "max"), ("value", "count"), ("fval", "sum")]).to_pydict()
    names = {"sum(value)": "value_sum", "min(value)": "value_min", "max(fval)": "fval_max", "count(value)": "value_count", "sum(fval)": "fval_sum"}
    rows_g = sorted(batch_rows(got, keys + list(names)), key=lambda r: sort_key(r[:len(keys)]))
    rows_w = sorted(batch_rows(want, keys + list(names.values())), key=lambda r: sort_key(r[:len(keys)]))
    assert len(rows_g) == len(rows_w) > 0
    for a, b in zip(rows_g, rows_w):
        for x, y in zip(a, b):
            if isinstance(y, float):
                assert x == pytest.approx(y, rel=1e-9, abs=1e-9)
            else:
                assert x == y, (a, b)


@pytest.mark.parametrize("typ", [pa.string(), pa.binary(), pa.large_string()], ids=["utf8", "binary", "large_utf8"])
def test_oracle_plain_string_leaves_agree_with_pyarrow(typ):
    """The reference sends = != < <= > >= on a plain string / binary column to Arrow's compare kernels
    (binaryscalarexpr.go:116-152) and contains to bytes.Contains (:234-270): the oracle's restatement against pyarrow's own
    kernels (bytewise order, NULL rows never match)."""
    rng = np.random.default_rng(77)
    words = ["", "a", "ab", "abc", "b", "ba", "zeta", "Zeta", "é", "value1", "value10", "value2"] + ["w%02d" % k for k in range(20)]
    n = 5000
    arr = pa.array([words[k] for k in rng.integers(0, len(words), n)], type=pa.string(), mask=rng.random(n) < 0.15).cast(typ)
    rec = pa.RecordBatch.from_arrays([arr], names=["name"])
    as_bin = arr.cast(pa.large_binary() if typ == pa.large_string() else pa.binary())
    N = Col("name")
    for lit in ("", "ab", "value1", "w10", "zz"):
        b = pa.scalar(lit.encode(), as_bin.type)
        cases = [(N == lit, pc.equal(as_bin, b)), (N != lit, pc.not_equal(as_bin, b)), (N < lit, pc.less(as_bin, b)), (N <= lit, pc.less_equal(as_bin, b)),
                 (N > lit, pc.greater(as_bin, b)), (N >= lit, pc.greater_equal(as_bin, b)),
                 (N.Contains(lit), pc.match_substring(as_bin, lit.encode())), (N.NotContains(lit), pc.invert(pc.match_substring(as_bin, lit.encode())))]
        for f, mask in cases:
            o = OraclePlan(f)
            try:
                _, idx = o.filter(rec)
            finally:
                o.close()
            want = np.flatnonzero(np.array(pc.fill_null(mask, False).to_pylist(), dtype=bool))
            assert np.array_equal(idx, want), (str(f), lit)


def test_oracle_prehashed_columns_replace_the_hash_not_the_keys():
    """FindHashedColumn (dynparquet/hashed.go:27-35, aggregate.go:386-392): a stored `hashed.<col>` int64 column is used as the
    group column's hash; with hashes that are a function of the value the groups and their printed keys are the same as without."""
    from tests.util import dict_array
    rng = np.random.default_rng(12)
    n = 5000
    vals = [None if rng.random() < 0.1 else b"p%d" % k for k in rng.integers(0, 20, n)]
    hashed = pa.array([0 if v is None else 1000 + int(v[1:]) for v in vals], type=pa.int64())  # NULL hashes to 0 (hashed.go:86-105)
    value = pa.array(rng.integers(0, 100, n), type=pa.int64())
    plain = pa.RecordBatch.from_arrays([dict_array(vals), value], names=["labels.path", "value"])
    pre = pa.RecordBatch.from_arrays([dict_array(vals), hashed, value], names=["labels.path", "hashed.labels.path", "value"])
    aggs, groups = [Sum(Col("value")), Count(Col("value"))], [Col("labels.path")]
    a = _oracle_runner_n(None, aggs, groups, 2)([plain.slice(0, 2000), plain.slice(2000)])
    b = _oracle_runner_n(None, aggs, groups, 2)([pre.slice(0, 2000), pre.slice(2000)])
    cols = ["labels.path", "sum(value)", "count(value)"]
    assert sorted(batch_rows(a, cols), key=sort_key) == sorted(batch_rows(b, cols), key=sort_key)
    assert sorted(b) == sorted(cols)  # the final stage drops the hashed.* helper columns (aggregate.go:561-563)


def test_convert_isnull_if_projections_follow_the_reference():
    """convertProjection / isNullProjection / ifExprProjection (physicalplan/project.go:493-702) as aggregate inputs and group keys.
    The convert case is logictest/testdata/exec/projection/convert's table and expression (`convert(value, float) * floatvalue`
    → 2.2, 4.4, 6.6 per row, :10-16) summed per stacktrace; isnull / if have no logictest vectors (parity unpinned beyond the
    code as read): a NULL row's RAW slot converts (no NULLs come out), isnull is a valid bool per row, if takes `then` only where
    the condition is valid and true."""
    import pyarrow as pa
    from frostdb_amd.logicalplan import BinaryExpr, Col, Convert, Count, If, IsNull, Literal, OP_GT, OP_MUL, Sum
    from oracle import OraclePlan
    from tests.util import dict_array
    rec = pa.RecordBatch.from_arrays([dict_array([b"value1"] * 3), dict_array([b"stack1", b"stack1", b"stack2"]), pa.array([1, 3, 5]), pa.array([2, 4, 6]),
                                      pa.array([1.1, 1.1, 1.1])], names=["labels.label1", "stacktrace", "timestamp", "value", "floatvalue"])
    e = BinaryExpr(Convert(Col("value")), OP_MUL, Col("floatvalue"))
    assert e.name == "convert(value, float64) * floatvalue"
    o = OraclePlan(None, [Sum(e)], [Col("stacktrace")])
    o.push(rec)
    d = o.finish().to_pydict()
    got = dict(zip(d["stacktrace"], d["sum(convert(value, float64) * floatvalue)"]))
    assert set(got) == {b"stack1", b"stack2"} and abs(got[b"stack1"] - 6.6) < 1e-9 and abs(got[b"stack2"] - 6.6) < 1e-9
    o.close()
    # isnull as a group key, if as an aggregate input, over a column with NULLs
    rec2 = pa.RecordBatch.from_arrays([pa.array([5, None, 7, None, 1]), pa.array([10, 20, 30, 40, 50])], names=["a", "b"])
    o = OraclePlan(None, [Sum(If(BinaryExpr(Col("a"), OP_GT, Literal(4)), Col("b"), Literal(1))), Count(Col("b"))], [IsNull(Col("a"))])
    o.push(rec2)
    d = o.finish().to_pydict()
    rows = sorted(zip(d["isnull(a)"], d["sum(if(a > 4) { b } else { 1})"], d["count(b)"]))
    assert rows == [(False, 10 + 30 + 1, 3), (True, 2, 2)]
    o.close()


def _overflow_records(n_records=3, rows=1000, key_bytes=1024):
    """Test_Aggregate_ArrayOverflow's input (query/physicalplan/aggregate_test.go:28-118), scaled: a fresh binary `stacktrace` per row — 8
    random bytes and zeros, like randByteSlice — an int64 `id` that counts up across the records, a random int64 `value`."""
    rng = np.random.default_rng(2024)
    recs = []
    for i in range(n_records):
        traces = [rng.integers(0, 256, 8, dtype=np.uint8).tobytes() + bytes(key_bytes - 8) for _ in range(rows)]
        recs.append(pa.RecordBatch.from_arrays(
            [pa.array(rng.integers(0, 1 << 40, rows), type=pa.int64()), pa.array(np.arange(rows, dtype=np.int64) + i * rows), pa.array(traces, type=pa.binary())],
            names=["value", "id", "stacktrace"]))
    return recs


def test_oracle_array_overflow_emits_several_records(monkeypatch):
    """Test_Aggregate_ArrayOverflow (aggregate_test.go:28-118): more key bytes than one binary builder takes (math.MaxInt32,
    optbuilders.go:221-224; lowered to 64 KiB here through $FDB_TEST_MAX_KEY_BYTES, the keys to 1 KiB) → the aggregate is split into several
    records (aggregate.go:426-468). The reference's own assertions: every record's columns have the record's length, the rows add up to
    n × rows; plus: no record holds more key bytes than the limit, there are several, and every group comes out once with its own sum."""
    limit = 64 * 1024
    monkeypatch.setenv("FDB_TEST_MAX_KEY_BYTES", str(limit))
    recs = _overflow_records()
    plan = OraclePlan(None, [Sum(Col("value"))], [Col("stacktrace"), Col("id")], nchains=1)
    for r in recs:
        plan.push(r)
    outs = [plan.finish()]
    while True:
        more = plan.finish_next()
        if more is None:
            break
        outs.append(more)
    got = [o.to_arrow() for o in outs]
    for o in outs:
        o.close()
    plan.close()
    assert len(got) >= 3000 * 1024 // limit  # 47 records of ≤ 64 rows
    total, seen = 0, {}
    for r in got:
        assert all(len(c) == r.num_rows for c in r.columns) and r.num_rows > 0
        st = r.column(r.schema.get_field_index("stacktrace"))
        assert sum(len(x) for x in st.to_pylist()) <= limit
        for k, i, v in zip(st.to_pylist(), r.column(r.schema.get_field_index("id")).to_pylist(), r.column(r.schema.get_field_index("sum(value)")).to_pylist()):
            assert (k, i) not in seen
            seen[(k, i)] = v
        total += r.num_rows
    assert total == 3000
    want = {}
    for r in recs:
        for v, i, k in zip(*(c.to_pylist() for c in r.columns)):
            want[(k, i)] = want.get((k, i), 0) + v
    assert seen == want
