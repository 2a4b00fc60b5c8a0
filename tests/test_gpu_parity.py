"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.

Every test here needs a real MI355X: run with ``pytest -m gpu``. Bars: bit-exact for COUNT/MIN/MAX/int64 SUM
and selection indices; float64 SUM within 1e-9 relative (BASELINE.json north_star).
"""
import math

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from frostdb_amd.logicalplan import OP_GT, OP_LT_EQ, OP_NOT_EQ, And, BinaryExpr, Col, Count, DynCol, Literal, Max, Min, Or, Sum, UInt64
from tests.golden import logictest_cases as G
from tests.util import (arrow_to_pydict, batch_rows, dict_array, fmt, make_prometheus_batch, parse_rows, record_from_rows,
                        sort_key, table_records)

pytestmark = pytest.mark.gpu

REL_TOL = 1e-9


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import physicalplan
    assert physicalplan.device_count() >= 1, "no HIP device visible"
    return physicalplan


@pytest.fixture(params=["specialised", "interpreted"])
def variant(request, monkeypatch):
    """Runs a test once with the run-time specialised plan kernel (the default) and once with the interpreting
    kernels only (FDB_NO_JIT is read at every launch)."""
    if request.param == "interpreted":
        monkeypatch.setenv("FDB_NO_JIT", "1")
    else:
        monkeypatch.delenv("FDB_NO_JIT", raising=False)
    return request.param


def rows_of(d, cols):
    return sorted(batch_rows(d, cols), key=sort_key)


def assert_same_result(got, want, cols, float_cols=()):
    """got/want: {col: [values]}; compares as multisets of rows; floats with REL_TOL, everything else exactly."""
    g, w = rows_of(got, cols), rows_of(want, cols)
    assert len(g) == len(w), (len(g), len(w))
    # sort by non-float columns only so that float noise cannot reorder rows
    key_cols = [i for i, c in enumerate(cols) if c not in float_cols]
    g.sort(key=lambda r: sort_key(tuple(r[i] for i in key_cols)))
    w.sort(key=lambda r: sort_key(tuple(r[i] for i in key_cols)))
    for rg, rw in zip(g, w):
        for c, a, b in zip(cols, rg, rw):
            if c in float_cols and a is not None and b is not None:
                assert math.isclose(a, b, rel_tol=REL_TOL, abs_tol=0.0) or a == b, (c, a, b)
            else:
                assert a == b, (c, rg, rw)


def run_gpu(pp, records, filter_expr, aggs, groups, resident=False):
    plan = pp.HashAggregatePlan(filter_expr, aggs, groups)
    keep = []
    try:
        for r in records:
            if resident:
                rb = pp.ResidentBatch(r)
                keep.append(rb)
                plan.Callback(rb)
            else:
                plan.Callback(r)
        return arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
        for k in keep:
            k.close()


def run_oracle(records, filter_expr, aggs, groups, nchains=1):
    from oracle import OraclePlan
    plan = OraclePlan(filter_expr, aggs, groups, nchains=nchains)
    for r in records:
        plan.push(r)
    d = plan.finish().to_pydict()
    plan.close()
    return d


# ---- golden vectors ---------------------------------------------------------------------------------------

def check_golden(case, d):
    d = dict(d)
    if "avg_of" in case:
        s, c = case["avg_of"]
        d["avg"] = [(a // b if isinstance(a, int) else a / float(b)) for a, b in zip(d[s], d[c])]
    got = sorted([tuple(fmt(v) for v in row) for row in batch_rows(d, case["out"])], key=sort_key)
    assert got == sorted(case["expected"], key=sort_key), case["cite"]


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("case", G.AGG_CASES, ids=[c["id"] for c in G.AGG_CASES])
def test_golden_aggregate(pp, variant, case, resident):
    d = run_gpu(pp, table_records(case["table"]), case.get("filter"), case["aggs"], case["groups"], resident=resident)
    check_golden(case, d)


@pytest.mark.parametrize("case", G.AGG_CASES, ids=[c["id"] for c in G.AGG_CASES])
def test_golden_aggregate_two_chains_merged(pp, case):
    """Two chains (one per insert) merged like Synchronizer + final stage (physicalplan.go:438-471)."""
    recs = table_records(case["table"])
    plans = [pp.HashAggregatePlan(case.get("filter"), case["aggs"], case["groups"]) for _ in range(2)]
    try:
        for i, r in enumerate(recs):
            plans[i % 2].Callback(r)
        plans[0].Merge(plans[1])
        check_golden(case, arrow_to_pydict(plans[0].Finish()))
    finally:
        for p in plans:
            p.Close()


def test_binary_scalar_operation_shapes(pp):
    """BenchmarkBinaryScalarOperation's input (binaryscalarexpr_test.go:15-41) on the device: 1 000 000 int64 values i % 10 against
    the scalar 4 under =, !=, <, <=, >, >= — selected rows and the filtered column are the oracle's (counts follow from the input)."""
    from tests.test_oracle_golden import BINARY_SCALAR_OPS, binary_scalar_record
    rec = binary_scalar_record()
    src = rec.column(0).to_numpy()
    for name, make, want in BINARY_SCALAR_OPS:
        plan = pp.HashAggregatePlan(make(Col("v")))
        try:
            idx = np.asarray(plan.Select(rec))
            assert len(idx) == want, name
            out = plan.Filter(rec)
            assert out.num_rows == want and np.array_equal(out.column(0).to_numpy(), src[idx]), name
            o = OraclePlanFilter(rec, make(Col("v")))
            assert np.array_equal(idx, o), name
        finally:
            plan.Close()


def OraclePlanFilter(rec, filt):
    from oracle import OraclePlan
    o = OraclePlan(filt)
    _, idx = o.filter(rec)
    o.close()
    return np.asarray(idx)


@pytest.mark.parametrize("case", G.CONTAINS_CASES, ids=[c["id"] for c in G.CONTAINS_CASES])
def test_golden_filter_contains(pp, case):
    """exec/filter/filter_contains on the device: the `bytes` schema's plain binary `value` column (LIKE / NOT LIKE) and its UINT64
    timestamp; the selected rows and every column of the filtered record are the vector's."""
    from tests.util import bytes_schema_record
    rec = bytes_schema_record(G.CONTAINS_TABLE)
    plan = pp.HashAggregatePlan(case["filter"])
    try:
        assert list(plan.Select(rec)) == case["rows"], case["cite"]
        out = plan.Filter(rec)
        d = arrow_to_pydict(out)
        assert out.schema.names == rec.schema.names
        for ci, name in enumerate(G.CONTAINS_TABLE["cols"]):
            assert d[name] == [G.CONTAINS_TABLE["rows"][r][ci] for r in case["rows"]], (case["cite"], name)
    finally:
        plan.Close()


@pytest.mark.parametrize("case", G.FILTER_CASES, ids=[c["id"] for c in G.FILTER_CASES])
def test_golden_filter(pp, case):
    rec = table_records(G.FILTER_TABLE)[0]
    plan = pp.HashAggregatePlan(case["filter"])
    try:
        idx = plan.Select(rec)
        assert list(idx) == case["rows"], case["cite"]
        out = plan.Filter(rec)
        if not case["rows"]:
            assert out is None
        else:
            d = arrow_to_pydict(out)
            assert d["timestamp"] == [r + 1 for r in case["rows"]]
            assert d["labels.label1"] == [b"value%d" % (r + 1) for r in case["rows"]]
            assert out.schema.names == rec.schema.names
            src = arrow_to_pydict(rec)
            for name in rec.schema.names:
                assert d[name] == [src[name][r] for r in case["rows"]], name
    finally:
        plan.Close()


def test_golden_inconsistent_schema(pp, variant):
    spec = G.INCONSISTENT_SCHEMA
    recs = [record_from_rows(r["cols"], parse_rows(r["cols"], r["rows"])) for r in spec["records"]]
    fns = {"sum": [Sum], "min": [Min], "max": [Max], "count": [Count], "avg": [Sum, Count]}
    for name, want in spec["expected"].items():
        d = run_gpu(pp, recs, None, [f(Col("value")) for f in fns[name]], [Col("labels.label2")])
        vals = [a // b for a, b in zip(d["sum(value)"], d["count(value)"])] if name == "avg" else d[f"{name}(value)"]
        assert sorted(vals, reverse=True) == want, (spec["cite"], name)
        assert sorted(d["labels.label2"], key=lambda x: (x is None, x)) == [b"value2", None]


# ---- randomized parity against the oracle ---------------------------------------------------------------------

CFG2 = dict(filter_expr=Col("labels.code") == "200", aggs=[Sum(Col("value"))], groups=[Col("labels.path")])
CFG3 = dict(
    filter_expr=And(Or(Col("labels.code") == "200", Col("labels.code") == "500"), Col("labels.method") == "GET",
               Col("labels.instance") != None),  # noqa: E711
    aggs=[Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp")), Sum(Col("value"))],
    groups=[Col("labels.path")])


@pytest.mark.parametrize("n", [0, 1, 7, 8, 63, 64, 65, 1023, 1024, 8191, 8192, 8193, 100_003])
def test_config2_sizes(pp, variant, n):
    rng = np.random.default_rng(1000 + n)
    b = make_prometheus_batch(rng, n)
    want = run_oracle([b], **CFG2) if n else {"labels.path": [], "sum(value)": []}
    got = run_gpu(pp, [b], **CFG2)
    if n == 0:
        assert all(len(v) == 0 for v in got.values())
        return
    assert_same_result(got, want, ["labels.path", "sum(value)"], float_cols={"sum(value)"})


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("rpt", [0, 4, 8])
def test_config3_multibatch(pp, variant, resident, rpt):
    rng = np.random.default_rng(7)
    batches = [make_prometheus_batch(rng, n, n_path=int(p)) for n, p in [(50_000, 64), (33_333, 200), (8192, 7), (1, 3)]]
    want = run_oracle(batches, **CFG3, nchains=2)
    plan = pp.HashAggregatePlan(CFG3["filter_expr"], CFG3["aggs"], CFG3["groups"])
    plan.set_tuning(rpt, 0)
    keep = []
    try:
        for b in batches:
            if resident:
                keep.append(pp.ResidentBatch(b))
                plan.Callback(keep[-1])
            else:
                plan.Callback(b)
        got = arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
    cols = ["labels.path", "count(value)", "min(timestamp)", "max(timestamp)", "sum(value)"]
    assert_same_result(got, want, cols, float_cols={"sum(value)"})


def test_multi_record_single_launch(pp, variant):
    """fdb_plan_push_batches: records with different dictionaries, sizes (incl. empty and sub-tile) and column sets
    scanned by ONE launch give the same result as the oracle fed record by record."""
    rng = np.random.default_rng(99)
    batches = [make_prometheus_batch(rng, n, n_path=int(p)) for n, p in
               [(10_000, 30), (0, 5), (3, 50), (2048, 7), (2049, 90), (70_001, 64), (1, 1)]]
    # one record lacks the filter column `labels.instance` entirely (missing-column rules) and has extra labels
    b = make_prometheus_batch(rng, 5000, n_path=11)
    b = b.drop_columns(["labels.instance"])
    batches.insert(3, b)
    for cfg in (CFG2, CFG3):
        want = run_oracle(batches, **cfg)
        plan = pp.HashAggregatePlan(cfg["filter_expr"], cfg["aggs"], cfg["groups"])
        keep = [pp.ResidentBatch(x) for x in batches]
        try:
            plan.CallbackResident(keep[:5])
            plan.CallbackResident(keep[5:])
            got = arrow_to_pydict(plan.Finish())
        finally:
            plan.Close()
        cols = ["labels.path"] + [a.Name() for a in cfg["aggs"]]
        assert_same_result(got, want, cols, float_cols={"sum(value)"})


def test_specialised_kernel_is_the_one_that_runs(pp, monkeypatch):
    """Default: the scan is the run-time compiled fdb_plan_kernel; FDB_NO_JIT / tuning mode 4: the interpreting kernel.
    Queries of the same shape but different literals share one compiled kernel (the literal is a run-time argument)."""
    rng = np.random.default_rng(5)
    b = make_prometheus_batch(rng, 20_000)
    monkeypatch.delenv("FDB_NO_JIT", raising=False)
    for cfg in (CFG2, CFG3):
        want = run_oracle([b], **cfg)
        cols = ["labels.path"] + [a.Name() for a in cfg["aggs"]]
        for mode, kernel in ((0, "fdb_plan_kernel"), (4, "scan_slots_kernel")):
            plan = pp.HashAggregatePlan(cfg["filter_expr"], cfg["aggs"], cfg["groups"])
            plan.set_tuning(0, mode << 25)
            try:
                plan.Callback(b)
                assert plan.last_kernel() == kernel
                got = arrow_to_pydict(plan.Finish())
            finally:
                plan.Close()
            assert_same_result(got, want, cols, float_cols={"sum(value)"})
    for code in ("404", "500"):
        cfg = dict(CFG2, filter_expr=Col("labels.code") == code)
        want = run_oracle([b], **cfg)
        plan = pp.HashAggregatePlan(cfg["filter_expr"], cfg["aggs"], cfg["groups"])
        try:
            plan.Callback(b)
            assert plan.last_kernel() == "fdb_plan_kernel"
            got = arrow_to_pydict(plan.Finish())
        finally:
            plan.Close()
        if want["labels.path"]:
            assert_same_result(got, want, ["labels.path", "sum(value)"], float_cols={"sum(value)"})
        else:
            assert all(len(v) == 0 for v in got.values())


def test_group_by_dynamic_labels_with_growing_dictionaries(pp, variant):
    """Group by the whole dynamic column set; later batches add dictionary entries AND a new label column,
    which forces the dense table to be re-laid-out (mixed-radix strides change)."""
    rng = np.random.default_rng(11)

    def batch(n, ncode, npath, with_zone):
        cols = {
            "labels.code": dict_array([None if rng.random() < 0.1 else b"c%d" % rng.integers(ncode) for _ in range(n)]),
            "labels.path": dict_array([None if rng.random() < 0.1 else b"p%d" % rng.integers(npath) for _ in range(n)]),
        }
        if with_zone:
            cols["labels.zone"] = dict_array([None if rng.random() < 0.5 else b"z%d" % rng.integers(3) for _ in range(n)])
        cols["value"] = pa.array(rng.integers(-1000, 1000, size=n), type=pa.int64())
        cols["floatvalue"] = pa.array(rng.normal(size=n))
        return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))

    batches = [batch(2000, 2, 3, False), batch(3000, 5, 3, False), batch(3000, 5, 9, True), batch(1000, 7, 12, True)]
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("value")), Max(Col("value")), Sum(Col("floatvalue")),
            Min(Col("floatvalue")), Max(Col("floatvalue"))]
    want = run_oracle(batches, None, aggs, [DynCol("labels")])
    got = run_gpu(pp, batches, None, aggs, [DynCol("labels")])
    cols = ["labels.code", "labels.path", "labels.zone"] + [a.Name() for a in aggs]
    assert_same_result(got, want, cols, float_cols={"sum(floatvalue)"})


def test_nullable_aggregated_columns_match_reference_quirks(pp, variant):
    """NULLs inside an aggregated column: COUNT counts them, SUM adds 0, MIN/MAX see the builder's zeroed slot
    (aggregate.go:784-950 + pqarrow/builder/optbuilders.go:337-340). Unpinned by reference tests —
    this pins the HIP path to the oracle's restatement of it."""
    rng = np.random.default_rng(5)
    n = 20_000
    g = dict_array([b"g%d" % rng.integers(5) for _ in range(n)])
    iv = pa.array(rng.integers(5, 100, size=n), type=pa.int64(), mask=rng.random(n) < 0.3)
    fv = pa.array(rng.uniform(1.0, 2.0, size=n), mask=rng.random(n) < 0.3)
    neg = pa.array(-rng.uniform(1.0, 2.0, size=n), mask=rng.random(n) < 0.3)
    b = pa.RecordBatch.from_arrays([g, iv, fv, neg], names=["labels.g", "value", "floatvalue", "neg"])
    aggs = [Count(Col("value")), Sum(Col("value")), Min(Col("value")), Max(Col("value")), Sum(Col("floatvalue")),
            Min(Col("floatvalue")), Max(Col("neg")), Count(Col("floatvalue"))]
    want = run_oracle([b], None, aggs, [Col("labels.g")])
    got = run_gpu(pp, [b], None, aggs, [Col("labels.g")])
    assert want["min(value)"] == [0] * 5 and want["min(floatvalue)"] == [0.0] * 5 and want["max(neg)"] == [0.0] * 5
    assert_same_result(got, want, ["labels.g"] + [a.Name() for a in aggs], float_cols={"sum(floatvalue)"})


def test_numeric_predicates_and_no_groups(pp, variant):
    rng = np.random.default_rng(21)
    n = 30_000
    b = pa.RecordBatch.from_arrays(
        [pa.array(rng.integers(0, 100, size=n), type=pa.int64(), mask=rng.random(n) < 0.05),
         pa.array(rng.uniform(0, 1, size=n), mask=rng.random(n) < 0.05),
         pa.array(rng.integers(0, 2**63, size=n, dtype=np.uint64), type=pa.uint64()),
         pa.array(rng.integers(0, 1000, size=n), type=pa.int64())],
        names=["timestamp", "floatvalue", "u", "value"])
    for f in [Col("timestamp") == 5, Col("timestamp") != 5, Col("timestamp") < 50, Col("timestamp") <= 50,
              Col("timestamp") > 50, Col("timestamp") >= 50, Col("floatvalue") < 0.25, Col("floatvalue") >= 0.5,
              Col("timestamp") < 33.5, Col("floatvalue") > 0, Col("u") > UInt64(2**62), Col("u") <= 2**40,
              And(Col("timestamp") > 10, Or(Col("floatvalue") < 0.1, Col("floatvalue") > 0.9)),
              Col("timestamp") == None, Col("missing") < 3, Col("missing") != 3]:  # noqa: E711
        aggs = [Count(Col("value")), Sum(Col("value")), Min(Col("value")), Max(Col("value"))]
        want = run_oracle([b], f, aggs, [])
        got = run_gpu(pp, [b], f, aggs, [])
        if not want:  # no row selected: the reference emits nothing, the C ABI a zero-row record
            assert all(len(v) == 0 for v in got.values()), str(f)
        else:
            assert got == want, str(f)


def test_sliced_columns_with_unaligned_offsets(pp, variant):
    """Arrow slices carry a non-zero offset: validity bitmaps must be re-based bit-exactly at import."""
    rng = np.random.default_rng(77)
    b = make_prometheus_batch(rng, 50_000, null_frac=0.2)
    for off, ln in [(3, 10_001), (8, 4096), (13, 8192), (49_990, 10)]:
        s = b.slice(off, ln)
        want = run_oracle([s], **CFG3)
        for resident in (False, True):
            got = run_gpu(pp, [s], **CFG3, resident=resident)
            cols = ["labels.path", "count(value)", "min(timestamp)", "max(timestamp)", "sum(value)"]
            if not want:
                assert all(len(v) == 0 for v in got.values())
            else:
                assert_same_result(got, want, cols, float_cols={"sum(value)"})


def test_selection_vector_matches_oracle_on_large_batch(pp):
    from oracle import OraclePlan
    rng = np.random.default_rng(3)
    b = make_prometheus_batch(rng, 250_007)
    f = CFG3["filter_expr"]
    plan = pp.HashAggregatePlan(f)
    o = OraclePlan(f)
    try:
        got = plan.Select(b)
        _, want = o.filter(b)
        assert np.array_equal(got, want)
        assert np.all(np.diff(got.astype(np.int64)) > 0)  # ascending, like bitmap.ToArray()
        out = plan.Filter(b)
        take = b.take(pa.array(want))
        assert arrow_to_pydict(out) == arrow_to_pydict(take)
    finally:
        plan.Close()
        o.close()


def test_error_behaviour_mirrors_reference(pp):
    rng = np.random.default_rng(4)
    b = make_prometheus_batch(rng, 100)
    # aggregate field not found (aggregate.go:367-380)
    plan = pp.HashAggregatePlan(None, [Sum(Col("nope"))], [Col("labels.path")])
    with pytest.raises(pp.FdbError) as e:
        plan.Callback(b)
    assert e.value.code == pp.FDB_ERR_NOT_FOUND and "aggregate field(s) not found" in e.value.msg
    plan.Close()
    # unsupported operator on a dictionary column (binaryscalarexpr.go:106-108)
    plan = pp.HashAggregatePlan(Col("labels.code") < "3", [Sum(Col("value"))], [])
    with pytest.raises(pp.UnsupportedError):
        plan.Callback(b)
    plan.Close()
    # SUM over a uint64 column: ErrUnsupportedSumType (aggregate.go:743-751)
    ub = pa.RecordBatch.from_arrays([pa.array([1, 2], type=pa.uint64())], names=["value"])
    plan = pp.HashAggregatePlan(None, [Sum(Col("value"))], [])
    with pytest.raises(pp.UnsupportedError) as e:
        plan.Callback(ub)
    assert "expected int64 or float64" in e.value.msg
    plan.Close()
    # regex on a dictionary<utf8> column is rejected like regexpfilter.go:55-61
    sb = pa.RecordBatch.from_arrays([dict_array(["a", "b"], pa.dictionary(pa.uint32(), pa.string())), pa.array([1, 2])],
                                    names=["labels.x", "value"])
    plan = pp.HashAggregatePlan(Col("labels.x").RegexMatch("a"), [Sum(Col("value"))], [])
    with pytest.raises(pp.UnsupportedError):
        plan.Callback(sb)
    plan.Close()
    # push after finish
    plan = pp.HashAggregatePlan(None, [Sum(Col("value"))], [Col("labels.path")])
    plan.Callback(b)
    plan.Finish()
    with pytest.raises(pp.FdbError):
        plan.Callback(b)
    plan.Close()


def test_draw(pp):
    plan = pp.HashAggregatePlan(Col("labels.code") == "200", [Sum(Col("value"))], [Col("labels.path")])
    assert plan.Draw().startswith("PredicateFilter (labels.code == 200) - HashAggregate (sum(value) by labels.path)")
    plan.Close()


def test_final_stage_plan_merges_partial_records(pp):
    """A plan created with final_stage=1 consumes the records partial plans emit (columns named by result name)
    and merges COUNT by SUM — the contract of HashAggregate(finalStage=true) (aggregate.go:340-348, :965-969)."""
    rng = np.random.default_rng(9)
    batches = [make_prometheus_batch(rng, 20_000, n_path=20) for _ in range(3)]
    aggs = [Count(Col("value")), Sum(Col("value")), Min(Col("timestamp")), Max(Col("timestamp"))]
    partials = [run_gpu_arrow(pp, [b], None, aggs, [Col("labels.path")]) for b in batches]
    final = pp.HashAggregatePlan(None, aggs, [Col("labels.path")], final_stage=True)
    for p in partials:
        final.Callback(p)
    got = arrow_to_pydict(final.Finish())
    final.Close()
    want = run_oracle(batches, None, aggs, [Col("labels.path")], nchains=3)
    assert_same_result(got, want, ["labels.path"] + [a.Name() for a in aggs], float_cols={"sum(value)"})


def run_gpu_arrow(pp, records, filter_expr, aggs, groups):
    plan = pp.HashAggregatePlan(filter_expr, aggs, groups)
    try:
        for r in records:
            plan.Callback(r)
        return plan.Finish()
    finally:
        plan.Close()


# ---- size-independent properties at a larger size ------------------------------------------------------------------

def test_properties_at_scale(pp):
    """4M rows: Σ count == selected rows, min ≤ max, group sums add up to the ungrouped sum, idempotent re-run,
    partition invariance (one batch vs four)."""
    rng = np.random.default_rng(2024)
    n = 4_000_000
    b = make_prometheus_batch(rng, n, n_path=1024, with_method=False)
    f = Col("labels.code") == "200"
    aggs = [Count(Col("value")), Sum(Col("value")), Min(Col("timestamp")), Max(Col("timestamp"))]
    rb = pp.ResidentBatch(b)
    plan = pp.HashAggregatePlan(f, aggs, [Col("labels.path")])
    plan.Callback(rb)
    grouped = arrow_to_pydict(plan.Finish())
    plan.Close()
    plan = pp.HashAggregatePlan(f, aggs, [])
    plan.Callback(rb)
    total = arrow_to_pydict(plan.Finish())
    plan.Close()
    code = b.column(0)
    sel = pc.fill_null(pc.equal(code.dictionary_decode(), pa.scalar(b"200", pa.binary())), False)
    n_sel = pc.sum(sel.cast(pa.int64())).as_py()
    assert sum(grouped["count(value)"]) == n_sel == total["count(value)"][0]
    assert all(lo <= hi for lo, hi in zip(grouped["min(timestamp)"], grouped["max(timestamp)"]))
    assert min(grouped["min(timestamp)"]) == total["min(timestamp)"][0]
    assert max(grouped["max(timestamp)"]) == total["max(timestamp)"][0]
    assert math.isclose(math.fsum(grouped["sum(value)"]), total["sum(value)"][0], rel_tol=REL_TOL)
    exact = math.fsum(pc.filter(b.column(b.schema.get_field_index("value")), sel).to_numpy())
    assert math.isclose(total["sum(value)"][0], exact, rel_tol=REL_TOL)
    # partition invariance
    parts = [pp.ResidentBatch(b.slice(i * (n // 4), n // 4)) for i in range(4)]
    plan = pp.HashAggregatePlan(f, aggs, [Col("labels.path")])
    for p in parts:
        plan.Callback(p)
    again = arrow_to_pydict(plan.Finish())
    plan.Close()
    cols = ["labels.path"] + [a.Name() for a in aggs]
    assert_same_result(again, grouped, cols, float_cols={"sum(value)"})
    rb.close()
    for p in parts:
        p.close()


def test_rccl_merge_single_rank(pp):
    """The cross-GPU merge (frostdb_amd/distributed.py) on a 1-rank RCCL group: both the aligned-layout fast path
    (raw table all-reduce) and the key-unification path must reproduce what Finish() gives on the same table.
    (World sizes 2 and 3 of the same code are covered on CPU/gloo in tests/test_distributed_cpu.py.)"""
    import os
    import torch
    import torch.distributed as dist
    from frostdb_amd import distributed as fd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(31)
        batches = [make_prometheus_batch(rng, 40_000, n_path=50), make_prometheus_batch(rng, 30_000, n_path=80)]
        cols = ["labels.path"] + [a.Name() for a in CFG3["aggs"]]

        def fresh():
            plan = pp.HashAggregatePlan(CFG3["filter_expr"], CFG3["aggs"], CFG3["groups"])
            for b in batches:
                plan.Callback(b)
            return plan

        want = run_oracle(batches, **CFG3)
        p1 = fresh()
        ok, rec = fd.merge_plan_aligned(p1)
        assert ok and rec is not None
        assert_same_result(arrow_to_pydict(rec), want, cols, float_cols={"sum(value)"})
        p1.Close()
        p2 = fresh()
        keys = p2.partial_keys()
        tensors = []
        for j in range(len(p2.aggs)):
            t = torch.empty((keys.num_rows,), dtype=torch.float64 if p2.agg_format(j) == "g" else torch.int64, device="cuda:0")
            p2.partial_state_into(j, t.data_ptr(), keys.num_rows * 8)
            tensors.append(t)
        rec2 = fd.merge_partials(keys, tensors, p2.aggs, key_types={f.name: f.type for f in keys.schema})
        assert_same_result(arrow_to_pydict(rec2), want, cols, float_cols={"sum(value)"})
        p2.Close()
    finally:
        if created:
            dist.destroy_process_group()


# ---- high-cardinality path: global hash table (many key columns, int64 keys, growth, migration, merge) -------------

def window_records_gpu(bucket):
    rec = table_records(G.WINDOW_TABLE)[0]
    ts = rec.column(rec.schema.get_field_index("timestamp"))
    b = pc.multiply(pc.divide(ts, pa.scalar(bucket, pa.int64())), pa.scalar(bucket, pa.int64()))
    return [rec.append_column("timestamp_bucket", b)]


@pytest.mark.parametrize("case", G.WINDOW_CASES, ids=[c["id"] for c in G.WINDOW_CASES])
def test_golden_window_int64_group_keys(pp, variant, case):
    """logictest/testdata/exec/aggregate/window: GROUP BY an int64 time bucket (the bucket column is materialised by
    the harness the way the reference's pre-aggregate Projection would) — exercises int64 keys in the hash table."""
    d = run_gpu(pp, window_records_gpu(case["bucket"]), None, case["aggs"], case["groups"])
    got = sorted(batch_rows(d, case["out"]), key=sort_key)
    assert got == sorted(case["expected"], key=sort_key), case["cite"]


def many_label_batch(rng, n, n_cols, card, n_groups=None, null_frac=0.03, int_key=False, sorted_rows=False):
    """n rows over `n_cols` dictionary label columns; if n_groups is given rows are drawn from that many distinct
    label tuples (mixed-radix digits of a group id, some digits NULL), like BASELINE.json's cfg 5."""
    if n_groups is None:
        digits = rng.integers(0, card, size=(n, n_cols))
        nulls = rng.random((n, n_cols)) < null_frac
    else:
        gid = rng.integers(0, n_groups, size=n)
        if sorted_rows:  # rows of one group next to each other: a scan of a table sorted by its label columns
            gid.sort()
        tab = rng.integers(0, card, size=(n_groups, n_cols))
        tnull = rng.random((n_groups, n_cols)) < null_frac
        digits, nulls = tab[gid], tnull[gid]
    arrays, names = [], []
    for c in range(n_cols):
        idx = pa.array(digits[:, c].astype(np.uint32), type=pa.uint32(), mask=nulls[:, c])
        arrays.append(pa.DictionaryArray.from_arrays(idx, pa.array([b"v%d_%d" % (c, k) for k in range(card)], type=pa.binary())))
        names.append("labels.l%02d" % c)
    if int_key:
        arrays.append(pa.array(rng.integers(1, 50, size=n) * 1000, type=pa.int64()))
        names.append("bucket")
    arrays += [pa.array(rng.integers(-100, 100, size=n), type=pa.int64()), pa.array(rng.uniform(0, 10, size=n))]
    names += ["value", "floatvalue"]
    return pa.RecordBatch.from_arrays(arrays, names=names)


def key_cols_of(batches, extra=()):
    names = []
    for b in batches:
        for n in b.schema.names:
            if (n.startswith("labels.") or n in extra) and n not in names:
                names.append(n)
    return names


def test_hash_path_many_columns_vs_oracle(pp, variant):
    rng = np.random.default_rng(404)
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("value")), Max(Col("value")), Sum(Col("floatvalue")), Min(Col("floatvalue"))]
    batches = [many_label_batch(rng, 30_000, 12, 3, n_groups=4000), many_label_batch(rng, 20_000, 12, 3, n_groups=3000)]
    want = run_oracle(batches, None, aggs, [DynCol("labels")])
    for resident in (False, True):
        got = run_gpu(pp, batches, None, aggs, [DynCol("labels")], resident=resident)
        cols = key_cols_of(batches) + [a.Name() for a in aggs]
        assert_same_result(got, want, cols, float_cols={"sum(floatvalue)"})


def test_resident_finish_keeps_the_result_in_hbm_and_equals_finish(pp):
    """fdb_plan_finish_batch (≙ Finish for a consumer on the device): a hash table with 12 label columns + an int64 key, COUNT /
    SUM / MIN / MAX over int64 and float64 (float MIN / MAX are decoded from their ordered keys on the device) is materialised as a
    resident batch — its export equals fdb_plan_finish of an identical plan row for row (as sets) and the oracle; the batch is then
    the INPUT of a final-stage plan (a device-side consumer: no host copy in between); small dense tables and plans with UNIQUE
    take the host route and come back resident too."""
    rng = np.random.default_rng(77)
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("value")), Max(Col("floatvalue")), Sum(Col("floatvalue")), Min(Col("floatvalue"))]
    groups = [DynCol("labels"), Col("bucket")]
    batches = [many_label_batch(rng, 40_000, 12, 3, n_groups=5000, int_key=True), many_label_batch(rng, 25_000, 12, 3, n_groups=3000, int_key=True)]
    cols = key_cols_of(batches, extra=("bucket",)) + [a.Name() for a in aggs]
    fcols = {"sum(floatvalue)"}
    want = run_oracle(batches, None, aggs, groups)

    def scanned():
        p = pp.HashAggregatePlan(None, aggs, groups)
        keep = [pp.ResidentBatch(b) for b in batches]
        p.CallbackResident(keep)
        return p, keep

    p1, k1 = scanned()
    p2, k2 = scanned()
    try:
        rb = p1.FinishResident()
        host = p2.Finish()
        got = rb.to_arrow()
        assert rb.num_rows == host.num_rows == len(want[cols[0]])
        assert got.schema.names == host.schema.names
        assert [str(f.type) for f in got.schema] == [str(f.type) for f in host.schema]
        assert_same_result(arrow_to_pydict(got), arrow_to_pydict(host), cols, float_cols=fcols)
        assert_same_result(arrow_to_pydict(got), want, cols, float_cols=fcols)
        # the resident partial record feeds a final-stage chain on the device
        fin = pp.HashAggregatePlan(None, aggs, groups, final_stage=True)
        fin.Callback(rb)
        assert_same_result(arrow_to_pydict(fin.Finish()), want, cols, float_cols=fcols)
        fin.Close()
        rb.close()
    finally:
        for p in (p1, p2):
            p.Close()
        for k in k1 + k2:
            k.close()
    # a small dense table, and a composite reducer: host route, resident result
    from frostdb_amd.logicalplan import Unique
    small = make_prometheus_batch(rng, 20_000, n_path=30)
    for a2 in ([Sum(Col("value")), Count(Col("value"))], [Unique(Col("timestamp")), Sum(Col("value"))]):
        p = pp.HashAggregatePlan(Col("labels.code") == "200", a2, [Col("labels.path")])
        q = pp.HashAggregatePlan(Col("labels.code") == "200", a2, [Col("labels.path")])
        p.Callback(small); q.Callback(small)
        rb = p.FinishResident()
        names = ["labels.path"] + [a.Name() for a in a2]
        assert_same_result(arrow_to_pydict(rb.to_arrow()), arrow_to_pydict(q.Finish()), names, float_cols={"sum(value)"})
        rb.close(); p.Close(); q.Close()


def test_hash_path_growth_and_filter(pp, variant):
    """> 100 k distinct groups: the table grows (device re-hash) several times while batches arrive; a filter runs in
    front of the hash scan; string + int64 key columns together."""
    rng = np.random.default_rng(405)
    aggs = [Sum(Col("value")), Count(Col("value")), Max(Col("floatvalue"))]
    batches = [many_label_batch(rng, 120_000, 10, 4, int_key=True), many_label_batch(rng, 90_000, 10, 4, int_key=True)]
    f = And(Col("labels.l00") != "v0_1", Col("value") > -50)
    groups = [DynCol("labels"), Col("bucket")]
    want = run_oracle(batches, f, aggs, groups)
    got = run_gpu(pp, batches, f, aggs, groups, resident=True)
    cols = key_cols_of(batches, extra=("bucket",)) + [a.Name() for a in aggs]
    assert len(want["value" if False else "sum(value)"]) > 100_000
    assert_same_result(got, want, cols)


def test_hash_path_sorted_input_folds_runs_in_lanes_and_waves(pp, variant):
    """Rows ordered by group (FrostDB's sorting columns): runs of equal keys are folded inside a lane's 4 rows and across the lanes
    of a wave before the table is touched. Long runs (≈28 rows: they span lanes), short ones (≈3 rows), a filter that punches holes
    into the runs, every reducer; against the oracle."""
    rng = np.random.default_rng(4243)
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("floatvalue")), Max(Col("floatvalue")), Sum(Col("floatvalue")), Min(Col("value"))]
    batches = [many_label_batch(rng, 100_000, 12, 3, n_groups=3_500, sorted_rows=True),
               many_label_batch(rng, 100_000, 12, 3, n_groups=30_000, sorted_rows=True),
               many_label_batch(rng, 30_001, 12, 3, n_groups=40, sorted_rows=True)]
    cols = key_cols_of(batches) + [a.Name() for a in aggs]
    for f in (None, Col("value") > -40, And(Col("labels.l00") != "v0_1", Col("floatvalue") < 9.0)):
        want = run_oracle(batches, f, aggs, [DynCol("labels")])
        got = run_gpu(pp, batches, f, aggs, [DynCol("labels")], resident=True)
        assert_same_result(got, want, cols, float_cols={"sum(floatvalue)"})


def test_hash_finish_transport_widths_and_slices(pp, monkeypatch):
    """Finish of a big hash table ships dictionary indices at the narrowest width their dictionary allows and widens them on host
    threads: a 1 000-entry dictionary (uint16 transport), a 70 000-entry one (uint32, copied as is), 12 entries (4 bits), 200 (uint8), 3 and 4
    entries (2 bits), an int64 key (8 bytes), NULLs in every label column but the last; 150 k groups in slices of 2^16 rows ($FDB_FINISH_SLICE_SHIFT; 2^20 by default):
    three slices, the last one short."""
    monkeypatch.setenv("FDB_FINISH_SLICE_SHIFT", "16")
    rng = np.random.default_rng(4242)
    n = 150_000
    cards = [1000, 70_000, 12, 200] + [3] * 4 + [4]  # (transport: 2 bytes, 4 bytes, 4 bits, 1 byte, 2 bits …; the last column has no NULLs: its bitmap stays on the device)
    arrays, names = [], []
    for c, card in enumerate(cards):
        idx = pa.array(rng.integers(0, card, n).astype(np.uint32), type=pa.uint32(), mask=(rng.random(n) < 0.02) if c < len(cards) - 1 else None)
        arrays.append(pa.DictionaryArray.from_arrays(idx, pa.array([b"w%d_%05d" % (c, k) for k in range(card)], type=pa.binary())))
        names.append("labels.l%02d" % c)
    arrays += [pa.array(rng.integers(1, 4, n) * 1000, type=pa.int64()), pa.array(rng.integers(-100, 100, n), type=pa.int64()), pa.array(rng.uniform(0, 10, n))]
    names += ["bucket", "value", "floatvalue"]
    b = pa.RecordBatch.from_arrays(arrays, names=names)
    aggs = [Count(Col("value")), Sum(Col("value")), Max(Col("floatvalue"))]
    groups = [DynCol("labels"), Col("bucket")]
    want = run_oracle([b], None, aggs, groups)
    got = run_gpu(pp, [b], None, aggs, groups, resident=True)
    assert len(want["count(value)"]) > (1 << 17)
    assert_same_result(got, want, names[:10] + [a.Name() for a in aggs])


def test_hash_finish_ships_the_ids_present_not_the_dictionary(pp, monkeypatch):
    """Finish prices PCIe per row, so a column whose dictionary is big but whose RESULT uses few entries ships the rank of each id among
    the ids present (present_ids_kernel → rank_ids_kernel) and the host widens through the rank → index table: a 1 000-entry dictionary
    of which 3 entries occur (2 bits per row instead of 16), 60 000 entries / 200 present (1 byte), 300 / 12 (4 bits), 5 000 / 400 (stays
    2 bytes), a column of NULLs only, NULLs everywhere else; three slices. Equal to the oracle, and to the same Finish with
    $FDB_NO_PRESENT_IDS."""
    monkeypatch.setenv("FDB_FINISH_SLICE_SHIFT", "16")
    monkeypatch.setenv("FDB_PRESENT_IDS_MIN_BYTES", "0")
    rng = np.random.default_rng(4343)
    n = 160_000
    shapes = [(1000, 3), (60_000, 200), (300, 12), (5000, 400), (700, 0)]
    arrays, names = [], []
    for c, (card, used) in enumerate(shapes):
        pick = rng.choice(card, size=max(used, 1), replace=False)
        idx = pick[rng.integers(0, len(pick), n)].astype(np.uint32)
        mask = np.ones(n, dtype=bool) if used == 0 else rng.random(n) < 0.03
        arrays.append(pa.DictionaryArray.from_arrays(pa.array(idx, type=pa.uint32(), mask=mask), pa.array([b"p%d_%05d" % (c, k) for k in range(card)], type=pa.binary())))
        names.append("labels.l%02d" % c)
    arrays += [pa.array(np.arange(n, dtype=np.int64) % 1500), pa.array(rng.integers(-100, 100, n), type=pa.int64())]
    names += ["bucket", "value"]
    b = pa.RecordBatch.from_arrays(arrays, names=names)
    aggs = [Sum(Col("value")), Count(Col("value"))]
    groups = [DynCol("labels"), Col("bucket")]
    want = run_oracle([b], None, aggs, groups)
    assert len(want["count(value)"]) > (1 << 17)
    got = run_gpu(pp, [b], None, aggs, groups, resident=True)
    assert_same_result(got, want, names[:6] + [a.Name() for a in aggs])
    monkeypatch.setenv("FDB_NO_PRESENT_IDS", "1")
    plain = run_gpu(pp, [b], None, aggs, groups, resident=True)
    assert_same_result(plain, want, names[:6] + [a.Name() for a in aggs])


def test_finish_emits_several_records_at_the_key_builders_size_limit(pp, monkeypatch):
    """Test_Aggregate_ArrayOverflow (query/physicalplan/aggregate_test.go:28-118) through the device path: 3 000 groups keyed by a fresh 1 KiB
    binary `stacktrace` + an int64 `id`, the key builder's limit lowered from math.MaxInt32 to 64 KiB ($FDB_TEST_MAX_KEY_BYTES) — Finish
    emits several records (fdb_plan_finish, then fdb_plan_finish_next until none is left; aggregate.go:426-468). The reference's assertions
    (every record's columns have its length, the rows add up), the per-record bound, and the union of the records against the oracle's."""
    from oracle import OraclePlan
    from tests.test_oracle_golden import _overflow_records
    limit = 64 * 1024
    monkeypatch.setenv("FDB_TEST_MAX_KEY_BYTES", str(limit))
    recs = _overflow_records()
    aggs, groups = [Sum(Col("value"))], [Col("stacktrace"), Col("id")]

    def rows_of(batches, device=True):
        out = {}
        for r in batches:
            assert all(len(c) == r.num_rows for c in r.columns) and r.num_rows > 0
            st = r.column(r.schema.get_field_index("stacktrace"))
            if device:
                assert pa.types.is_binary(st.type)  # (the key column keeps its input type: plain binary, 32-bit offsets)
            else:
                st = st.dictionary_decode() if pa.types.is_dictionary(st.type) else st  # (OracleBatch.to_arrow hands strings over as codes + table)
            assert sum(len(x) for x in st.to_pylist()) <= limit
            for k, i, v in zip(st.to_pylist(), r.column(r.schema.get_field_index("id")).to_pylist(), r.column(r.schema.get_field_index("sum(value)")).to_pylist()):
                assert (k, i) not in out
                out[(k, i)] = v
        return out

    oplan = OraclePlan(None, aggs, groups, nchains=1)
    for r in recs:
        oplan.push(r)
    outs = [oplan.finish()]
    while (more := oplan.finish_next()) is not None:
        outs.append(more)
    want = rows_of([o.to_arrow() for o in outs], device=False)
    for o in outs:
        o.close()
    oplan.close()
    for resident in (False, True):
        plan = pp.HashAggregatePlan(None, aggs, groups)
        keep = []
        try:
            for r in recs:
                if resident:
                    keep.append(pp.ResidentBatch(r))
                    plan.Callback(keep[-1])
                else:
                    plan.Callback(r)
            got = plan.FinishAll()
        finally:
            plan.Close()
            for k in keep:
                k.close()
        assert len(got) >= 3000 * 1024 // limit and sum(r.num_rows for r in got) == 3000
        assert rows_of(got) == want
    # the next operator of a plan sees every record (≙ next.Callback per aggregate, aggregate.go:617-626)
    seen, finished = [], []
    plan = pp.HashAggregatePlan(None, aggs, groups)
    try:
        plan.SetNext(seen.append, lambda: finished.append(True))
        for r in recs:
            plan.Callback(r)
        plan.Finish()
    finally:
        plan.Close()
    assert finished == [True] and rows_of(seen) == want
    # every kind of column through the cut: a dictionary key with NULLs, a bool key, NULL stacktraces, float and count aggregates
    rng = np.random.default_rng(7)
    wide = []
    for r in recs:
        n = r.num_rows
        st = pa.array([None if i % 17 == 0 else v for i, v in enumerate(r.column(2).to_pylist())], type=pa.binary())
        lab = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 5, n).astype(np.uint32), mask=rng.random(n) < 0.1), pa.array([b"a", b"b", b"c", b"d", b"e"], type=pa.binary()))
        wide.append(pa.RecordBatch.from_arrays([r.column(0), r.column(1), st, lab, pa.array(rng.random(n) < 0.5), pa.array(rng.uniform(0, 1, n))],
                                                names=["value", "id", "stacktrace", "labels.x", "flag", "f"]))
    aggs2, groups2 = [Sum(Col("value")), Count(Col("value")), Max(Col("f"))], [Col("stacktrace"), Col("labels.x"), Col("flag"), Col("id")]
    monkeypatch.delenv("FDB_TEST_MAX_KEY_BYTES")  # (the oracle's one-record answer: the cut changes which record a group is in, not the groups)
    want2 = run_oracle(wide, None, aggs2, groups2)
    monkeypatch.setenv("FDB_TEST_MAX_KEY_BYTES", str(limit))
    plan = pp.HashAggregatePlan(None, aggs2, groups2)
    try:
        for r in wide:
            plan.Callback(r)
        got_recs = plan.FinishAll()
    finally:
        plan.Close()
    assert len(got_recs) > 20 and sum(r.num_rows for r in got_recs) == len(want2["id"])
    for r in got_recs:
        st = r.column(r.schema.get_field_index("stacktrace"))
        assert sum(len(x) for x in st.to_pylist() if x is not None) <= limit
    got2 = _concat_results(got_recs)
    assert_same_result(got2, want2, ["stacktrace", "labels.x", "flag", "id"] + [a.Name() for a in aggs2])
    # without the hook: one record (3 MB of keys are far below math.MaxInt32)
    monkeypatch.delenv("FDB_TEST_MAX_KEY_BYTES")
    plan = pp.HashAggregatePlan(None, aggs, groups)
    try:
        for r in recs:
            plan.Callback(r)
        assert len(plan.FinishAll()) == 1
    finally:
        plan.Close()


def test_dense_to_hash_migration_and_merge(pp, variant):
    """First record: two label columns (dense table). Second record brings ten more label columns → the plan migrates
    its dense state into the hash table. Then a second chain in hash mode is merged in (Synchronizer + final stage)."""
    rng = np.random.default_rng(406)
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("floatvalue")), Max(Col("value"))]
    b1 = many_label_batch(rng, 20_000, 2, 5)
    b2 = many_label_batch(rng, 20_000, 12, 3, n_groups=2500)
    b3 = many_label_batch(rng, 15_000, 12, 3, n_groups=2500)
    want = run_oracle([b1, b2, b3], None, aggs, [DynCol("labels")])
    p1 = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
    p2 = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
    try:
        p1.Callback(b1)
        p1.Callback(b2)
        p2.Callback(b3)
        p1.Merge(p2)
        got = arrow_to_pydict(p1.Finish())
    finally:
        p1.Close(); p2.Close()
    cols = key_cols_of([b1, b2, b3]) + [a.Name() for a in aggs]
    assert_same_result(got, want, cols)
    # merging a hash-mode chain INTO a dense one
    p1 = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
    p2 = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
    try:
        p1.Callback(b1)
        p2.Callback(b2)
        p2.Callback(b3)
        p1.Merge(p2)
        got = arrow_to_pydict(p1.Finish())
    finally:
        p1.Close(); p2.Close()
    assert_same_result(got, want, cols)


def test_hash_path_properties_at_scale(pp, variant):
    """2 M rows, 16 label columns, 300 k groups: Σcount = rows, group count = distinct tuples, sums add up."""
    rng = np.random.default_rng(407)
    n, n_groups = 2_000_000, 300_000
    b = many_label_batch(rng, n, 16, 4, n_groups=n_groups)
    aggs = [Count(Col("value")), Sum(Col("value")), Sum(Col("floatvalue"))]
    rb = pp.ResidentBatch(b)
    plan = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
    plan.Callback(rb)
    out = plan.Finish()
    plan.Close(); rb.close()
    d = {k: v for k, v in zip(out.schema.names, out.columns)}
    assert pc.sum(d["count(value)"]).as_py() == n
    assert pc.sum(d["sum(value)"]).as_py() == pc.sum(b.column(b.schema.get_field_index("value"))).as_py()
    exact = math.fsum(b.column(b.schema.get_field_index("floatvalue")).to_numpy())
    assert math.isclose(math.fsum(d["sum(floatvalue)"].to_numpy()), exact, rel_tol=REL_TOL)
    keys = pa.table([c for n_, c in d.items() if n_.startswith("labels.")], names=[n_ for n_ in d if n_.startswith("labels.")])
    distinct_in = pa.table([b.column(i) for i, n_ in enumerate(b.schema.names) if n_.startswith("labels.")],
                           names=[n_ for n_ in b.schema.names if n_.startswith("labels.")]).group_by(keys.column_names).aggregate([]).num_rows
    assert out.num_rows == distinct_in


# ---- hash-partitioned exchange of high-cardinality tables ("G large") ----------------------------------------

def _concat_results(recs):
    out = {}
    for r in recs:
        d = arrow_to_pydict(r)
        for k, v in d.items():
            out.setdefault(k, []).extend(v)
    return out


def test_hash_exchange_two_virtual_ranks(pp):
    """Two 'ranks' (two plans on this GPU) with different data, different dictionaries and a column only one of them has:
    schema agreement → each exports its table hash-partitioned for 2 owners → each owner imports its partition from both.
    The union of the two shards equals the oracle over all the data, and no group appears in both shards."""
    from frostdb_amd import distributed as fd
    rng = np.random.default_rng(777)
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("value")), Max(Col("floatvalue")), Sum(Col("floatvalue"))]
    groups = [DynCol("labels"), Col("bucket")]
    ba = [many_label_batch(rng, 40_000, 10, 4, n_groups=9000, int_key=True), many_label_batch(rng, 10_000, 10, 4, n_groups=500, int_key=True)]
    bb = [many_label_batch(rng, 35_000, 11, 5, n_groups=8000, int_key=True)]  # one more label column, one more value per dictionary
    want = run_oracle(ba + bb, None, aggs, groups)
    plans = []
    for batches in (ba, bb):
        p = pp.HashAggregatePlan(None, aggs, groups)
        for b in batches:
            p.Callback(b)
        plans.append(p)
    schema = fd.unify_group_schemas([fd._schema_to_obj(p.group_schema()) for p in plans])
    shards = [plans[0].clone_empty(), plans[0].clone_empty()]
    try:
        for s in shards:
            s.seed_groups(schema)
        for p in plans:
            ptr, counts, row_bytes = p.hash_export(shards[0], 2)
            assert sum(counts) == p.num_groups() and min(counts) > 0.3 * sum(counts) / 2
            shards[0].hash_import(ptr, counts[0])
            shards[1].hash_import(ptr + counts[0] * row_bytes, counts[1])
        recs = [s.Finish() for s in shards]
    finally:
        for p in plans + shards:
            p.Close()
    cols = key_cols_of(ba + bb, extra=("bucket",)) + [a.Name() for a in aggs]
    got = _concat_results(recs)
    assert_same_result(got, want, cols, float_cols={"sum(floatvalue)"})
    keysets = [set(rows_of(arrow_to_pydict(r), key_cols_of(ba + bb, extra=("bucket",)))) if r.num_rows else set() for r in recs]
    assert not (keysets[0] & keysets[1])


def test_hash_exchange_rccl_single_rank_and_dense_source(pp):
    """merge_plan_alltoall on a 1-rank RCCL group: the shard is the whole result; the source table may be dense (cfg 3) —
    it is migrated to a hash table — or already a hash table."""
    import os
    import torch
    import torch.distributed as dist
    from frostdb_amd import distributed as fd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(32)
        dense_batches = [make_prometheus_batch(rng, 40_000, n_path=50), make_prometheus_batch(rng, 30_000, n_path=80)]
        aggs_h = [Sum(Col("value")), Count(Col("value")), Max(Col("floatvalue"))]
        hash_batches = [many_label_batch(rng, 60_000, 12, 4, n_groups=20_000)]
        for batches, cfg, cols in (
                (dense_batches, CFG3, ["labels.path"] + [a.Name() for a in CFG3["aggs"]]),
                (hash_batches, dict(filter_expr=None, aggs=aggs_h, groups=[DynCol("labels")]), key_cols_of(hash_batches) + [a.Name() for a in aggs_h])):
            want = run_oracle(batches, **cfg)
            plan = pp.HashAggregatePlan(cfg["filter_expr"], cfg["aggs"], cfg["groups"])
            for b in batches:
                plan.Callback(b)
            shard = fd.merge_plan_alltoall(plan, chunk_bytes=(128 << 20) if cfg is CFG3 else 100_000)  # second case: ≈30 slices
            try:
                got = arrow_to_pydict(shard.Finish())
            finally:
                shard.Close()
                plan.Close()
            assert_same_result(got, want, cols, float_cols={"sum(value)", "sum(floatvalue)"})
    finally:
        if created:
            dist.destroy_process_group()


# ---- plain string / binary columns (not dictionary-encoded): filter leaves and group keys ---------------------------------

def plain_batch(rng, n, typ=None, n_vals=40, null_frac=0.1, extra_null_col=False):
    """`name` : plain string / binary (typ), `tag` : the other one, `code` : dictionary, `flag` : bool, `value` / `floatvalue`."""
    typ = typ or pa.string()
    other = pa.binary() if typ in (pa.string(), pa.large_string()) else pa.string()
    words = ["", "a", "ab", "abc", "b", "ba", "zeta", "Zeta", "é", "abc\x00", "value1", "value10", "value2"] + ["w%03d" % k for k in range(n_vals)]
    pick = rng.integers(0, len(words), n)
    name = pa.array([words[k] for k in pick], type=pa.string(), mask=rng.random(n) < null_frac).cast(typ)
    tag = pa.array([("t%d" % (k % 7)) for k in rng.integers(0, 1000, n)], type=pa.string(), mask=rng.random(n) < null_frac).cast(other)
    code = dict_array([None if rng.random() < 0.05 else b"c%d" % k for k in rng.integers(0, 5, n)])
    flag = pa.array(rng.integers(0, 2, n).astype(bool), mask=rng.random(n) < null_frac)
    arrays = [name, tag, code, flag, pa.array(rng.integers(-50, 50, n), type=pa.int64()), pa.array(rng.uniform(0, 10, n))]
    names = ["name", "tag", "labels.code", "flag", "value", "floatvalue"]
    return pa.RecordBatch.from_arrays(arrays, names=names)


PLAIN_FILTERS = [
    Col("name") == "abc", Col("name") != "abc", Col("name") < "b", Col("name") <= "ab", Col("name") > "value1", Col("name") >= "w010",
    Col("name") == "", Col("name") != "", Col("name") == None, Col("name") != None,  # noqa: E711  (a NULL scalar: no row either way)
    Col("name").RegexMatch("^value1"), Col("name").RegexNotMatch("^w0[0-3]"), Col("name").Contains("a"), Col("name").NotContains("b"),
    Col("name").Contains(None), Col("name").NotContains(None),
    Or(Col("name") == "abc", Col("name") == "zeta"), And(Col("name") >= "a", Col("name") < "c"),
    And(Or(Col("name") < "b", Col("tag") == "t3"), Col("labels.code") != "c1", Col("value") > -20),
    Or(Col("name") == None, Col("name") > "w"),  # noqa: E711
]


@pytest.mark.parametrize("typ", [pa.string(), pa.binary(), pa.large_string(), pa.large_binary()], ids=["utf8", "binary", "large_utf8", "large_binary"])
def test_plain_string_filter_leaves_vs_oracle(pp, typ):
    """Every operator the reference accepts on *array.String / *array.Binary (Arrow compare kernels, contains, regex) — selection
    vector and filtered record (plain columns leave as plain columns, bools as bools) against the oracle."""
    from oracle import OraclePlan
    rng = np.random.default_rng(8101)
    b = plain_batch(rng, 20_011, typ)
    for f in PLAIN_FILTERS:
        plan = pp.HashAggregatePlan(f)
        o = OraclePlan(f)
        try:
            got = plan.Select(b)
            _, want = o.filter(b)
            assert np.array_equal(got, want), str(f)
            out = plan.Filter(b)
            if len(want) == 0:
                assert out is None
            else:
                take = b.take(pa.array(want))
                assert out.schema.field("name").type == typ and out.schema.field("flag").type == pa.bool_()
                assert arrow_to_pydict(out) == arrow_to_pydict(take), str(f)
        finally:
            plan.Close()
            o.close()


def test_golden_projection_vectors_on_this_path(pp):
    """exec/projection/math_projection:17-21 and exec/projection/bool:10-14 through the device path."""
    from tests.test_oracle_golden import bool_table_record
    c = G.PROJ_MATH_GROUPED
    d = run_gpu(pp, table_records(G.PROJ_MATH_TABLE), None, c["aggs"], c["groups"])
    assert sorted(batch_rows(d, c["out"]), key=sort_key) == sorted(c["expected"], key=sort_key), c["cite"]
    rec = bool_table_record()
    for case in G.BOOL_FILTER_CASES:
        plan = pp.HashAggregatePlan(case["filter"])
        try:
            assert list(plan.Select(rec)) == case["rows"], case["cite"]
            out = plan.Filter(rec)
            assert arrow_to_pydict(out)["name"] == [G.BOOL_TABLE_ROWS[r][0] for r in case["rows"]]
        finally:
            plan.Close()


def test_golden_root_aggregate_tests(pp):
    """TestDurationAggregation / TestAggregationProjection (root aggregate_test.go:150-343) through the device path."""
    from tests.test_oracle_golden import check_root_aggregate_cases
    check_root_aggregate_cases(_gpu_runner(pp))


def test_bool_column_filter_vs_oracle(pp):
    """Arrow's compare kernels on a boolean column with a boolean scalar (false < true); NULL rows never match."""
    from oracle import OraclePlan
    rng = np.random.default_rng(8106)
    b = plain_batch(rng, 30_003)
    for f in (Col("flag") == True, Col("flag") != True, Col("flag") == False, Col("flag") < True, Col("flag") >= False,  # noqa: E712
              And(Col("flag") == True, Col("value") > 0), Or(Col("flag") == False, Col("name") == "abc")):  # noqa: E712
        plan = pp.HashAggregatePlan(f, [Sum(Col("value")), Count(Col("value"))], [Col("labels.code")])
        o = OraclePlan(f)
        try:
            _, want = o.filter(b)
            assert np.array_equal(plan.Select(b), want), str(f)
            plan.Callback(b)
            got = arrow_to_pydict(plan.Finish())
        finally:
            plan.Close()
            o.close()
        assert_same_result(got, run_oracle([b], f, [Sum(Col("value")), Count(Col("value"))], [Col("labels.code")]), ["labels.code", "sum(value)", "count(value)"])


def test_bool_and_uint64_group_keys_vs_oracle(pp):
    """HashArray hashes a bool key as 1 / 2 and NULL as 0 (dynparquet/hashed.go:228-242): false, true and NULL are three groups;
    a uint64 key hashes by identity (0 ≡ NULL like int64). Key columns keep their type. Two chains merged, and AND() beside them."""
    from frostdb_amd.logicalplan import AndAgg
    rng = np.random.default_rng(8107)
    batches = []
    for n in (40_000, 25_000):
        b = plain_batch(rng, n)
        u = pa.array(rng.integers(1, 2000, n).astype(np.uint64) * np.uint64(1 << 53), type=pa.uint64(), mask=rng.random(n) < 0.05)
        batches.append(b.append_column("shard", u))
    aggs = [Sum(Col("value")), Count(Col("value")), AndAgg(Col("flag"))]
    for groups in ([Col("flag")], [Col("flag"), Col("labels.code")], [Col("shard")], [Col("shard"), Col("flag"), Col("name")]):
        want = run_oracle(batches, Col("value") > -40, aggs, groups, nchains=2)
        p1 = pp.HashAggregatePlan(Col("value") > -40, aggs, groups)
        p2 = pp.HashAggregatePlan(Col("value") > -40, aggs, groups)
        try:
            p1.Callback(batches[0]); p2.Callback(batches[1])
            p1.Merge(p2)
            rec = p1.Finish()
        finally:
            p1.Close(); p2.Close()
        for g in groups:
            assert rec.schema.field(g.name).type == batches[0].schema.field(g.name).type
        if groups[0].name == "flag":
            assert set(want["flag"]) == {True, False, None}
        assert_same_result(arrow_to_pydict(rec), want, [g.name for g in groups] + [a.Name() for a in aggs])


def test_host_regex_engine_decides_regex_leaves(pp):
    """fdb_plan_desc.regex_match: the host application's regex engine is asked once per distinct value (the Go shim passes
    regexp.Regexp.Match, so `=~` means exactly what it means in the reference). Python's `re` plays the host here — its syntax
    has (?i) and (?P<name>…), which std::regex does not: without the engine such a plan is rejected at create, with it the
    selection equals re.search per row, on dictionary and plain columns, and for the missing-column rule (matches iff the
    pattern matches the empty string, regexpfilter.go:23-33)."""
    import re
    from frostdb_amd.logicalplan import regex_matcher
    calls = []

    def search(pattern, value):
        calls.append(value)
        return re.search(pattern, value) is not None

    engine = regex_matcher(search)
    rng = np.random.default_rng(8108)
    b = plain_batch(rng, 50_000, pa.binary())
    names = b.column(0).to_pylist()
    codes = b.column(2).to_pylist()
    for pat in (b"(?i)^ZETA$", b"(?P<head>w0)[0-4]", b"^$", b"b+a?"):
        with_engine = pp.HashAggregatePlan(Col("name").RegexMatch(pat.decode()), regex=engine)
        try:
            calls.clear()
            got = with_engine.Select(b)
            want = np.array([i for i, v in enumerate(names) if v is not None and re.search(pat, v)], dtype=np.uint32)
            assert np.array_equal(got, want), pat
            assert 0 < len(calls) <= len(set(names)) + 1  # per distinct value (+ the compile probe), not per row
            calls.clear()
            assert np.array_equal(with_engine.Select(b), want) and not calls  # same values again: the truth table is remembered
        finally:
            with_engine.Close()
    p = pp.HashAggregatePlan(And(Col("labels.code").RegexNotMatch("(?i)C[12]"), Col("labels.absent").RegexMatch("(?i)x*")), regex=engine)
    try:
        want = np.array([i for i, v in enumerate(codes) if v is not None and not re.search(b"(?i)C[12]", v)], dtype=np.uint32)
        assert np.array_equal(p.Select(b), want)
    finally:
        p.Close()
    with pytest.raises(pp.FdbError):  # the host engine rejects the pattern: surfaces at create, like regexp.Compile at plan build
        pp.HashAggregatePlan(Col("name").RegexMatch("(unclosed"), regex=engine)
    # Without a host engine the library's own RE2-syntax engine (fdb_regex.h) decides: inline flags, named groups, POSIX classes,
    # \Q…\E, `$` = end of text — the same rows as the oracle, whose `=~` goes through Python's `re` on the translated pattern
    # (an independent engine), on plain and dictionary columns, filter and aggregate alike.
    for pat in ("(?i)^ZETA$", "(?P<head>w0)[0-4]$", "^[[:alpha:]]+[[:digit:]]?$", "\\Qw0\\E[5-9]|^$", "(?i:ze)ta|b{2,}", "^(?U)a+?b*$"):
        builtin = pp.HashAggregatePlan(Col("name").RegexMatch(pat))
        try:
            got = builtin.Select(b)
        finally:
            builtin.Close()
        _, idx = _oracle_filter(b, Col("name").RegexMatch(pat))
        assert np.array_equal(got, idx), pat
    f2 = And(Col("labels.code").RegexNotMatch("(?i)C[12]$"), Col("name").RegexMatch("(?s)^.{2,5}$"))
    got = run_gpu(pp, [b], f2, [Count(Col("value"))], [Col("labels.code")])
    want = run_oracle([b], f2, [Count(Col("value"))], [Col("labels.code")])
    assert_same_result(got, want, ["labels.code", "count(value)"])
    with pytest.raises(pp.FdbError):  # not RE2 syntax (look-ahead): refused at create
        pp.HashAggregatePlan(Col("name").RegexMatch("(?=z)eta"))


def test_plain_string_filter_errors(pp):
    rng = np.random.default_rng(8102)
    b = plain_batch(rng, 100)
    # a string array against a number: Arrow has no such compare kernel → ErrUnsupportedBinaryOperation (binaryscalarexpr.go:126-129)
    plan = pp.HashAggregatePlan(Col("name") == 3, [Count(Col("value"))], [])
    with pytest.raises(pp.UnsupportedError):
        plan.Callback(b)
    plan.Close()


@pytest.mark.parametrize("typ", [pa.string(), pa.binary(), pa.large_string()], ids=["utf8", "binary", "large_utf8"])
def test_plain_string_group_keys_vs_oracle(pp, typ, variant):
    """Group by plain string columns (the reference's own Test_Aggregate_ArrayOverflow groups by a *array.Binary column,
    query/physicalplan/aggregate_test.go:28-118): NULL keys, "" ≠ NULL, two chains merged, the key columns come back as plain columns of the input type."""
    rng = np.random.default_rng(8103)
    batches = [plain_batch(rng, 60_000, typ), plain_batch(rng, 45_000, typ, n_vals=70)]
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("floatvalue")), Max(Col("value"))]
    for groups, filt in (([Col("name")], None), ([Col("name"), Col("tag"), Col("labels.code")], Col("name") != "ab"), ([Col("tag")], Col("name") < "w")):
        want = run_oracle(batches, filt, aggs, groups, nchains=2)
        p1 = pp.HashAggregatePlan(filt, aggs, groups)
        p2 = pp.HashAggregatePlan(filt, aggs, groups)
        try:
            p1.Callback(batches[0]); p2.Callback(batches[1])
            p1.Merge(p2)
            rec = p1.Finish()
        finally:
            p1.Close(); p2.Close()
        for g in groups:
            assert rec.schema.field(g.name).type == batches[0].schema.field(g.name).type
        cols = [g.name for g in groups] + [a.Name() for a in aggs]
        assert_same_result(arrow_to_pydict(rec), want, cols, float_cols={"min(floatvalue)"})


def test_plain_string_keys_hash_table_and_exchange(pp):
    """A high-cardinality plain string key (hash table path) plus a dictionary key: resident and pushed records, then the
    hash-partitioned exchange between two plans whose value sets differ (the schema agreement carries the plain column's values)."""
    from frostdb_amd import distributed as fd
    rng = np.random.default_rng(8104)

    def big(n, lo, hi):
        ids = rng.integers(lo, hi, n)
        user = pa.array(["user-%07d" % k for k in ids], type=pa.string(), mask=rng.random(n) < 0.02)
        b = many_label_batch(rng, n, 9, 4, n_groups=3000)
        return b.append_column("user", user)

    batches = [big(40_000, 0, 30_000), big(30_000, 20_000, 60_000)]
    aggs = [Sum(Col("value")), Count(Col("value")), Max(Col("floatvalue"))]
    groups = [DynCol("labels"), Col("user")]
    want = run_oracle(batches, None, aggs, groups)
    cols = key_cols_of(batches, extra=("user",)) + [a.Name() for a in aggs]
    assert len(want["user"]) > 50_000
    for resident in (False, True):
        got_rec = None
        plan = pp.HashAggregatePlan(None, aggs, groups)
        keep = []
        try:
            for b in batches:
                if resident:
                    keep.append(pp.ResidentBatch(b))
                    plan.Callback(keep[-1])
                else:
                    plan.Callback(b)
            got_rec = plan.Finish()
        finally:
            plan.Close()
        assert got_rec.schema.field("user").type == pa.string()
        assert_same_result(arrow_to_pydict(got_rec), want, cols)
    plans = [pp.HashAggregatePlan(None, aggs, groups), pp.HashAggregatePlan(None, aggs, groups)]
    shards = [plans[0].clone_empty(), plans[0].clone_empty()]
    try:
        for p, b in zip(plans, batches):
            p.Callback(b)
        schema = fd.unify_group_schemas([fd._schema_to_obj(p.group_schema()) for p in plans])
        for sh in shards:
            sh.seed_groups(schema)
        for p in plans:
            ptr, counts, row_bytes = p.hash_export(shards[0], 2)
            shards[0].hash_import(ptr, counts[0])
            shards[1].hash_import(ptr + counts[0] * row_bytes, counts[1])
        recs = [sh.Finish() for sh in shards]
    finally:
        for p in plans + shards:
            p.Close()
    assert all(r.schema.field("user").type == pa.string() for r in recs)
    assert_same_result(_concat_results(recs), want, cols)


def test_plain_and_dictionary_key_types_do_not_mix(pp):
    """The reference's key builder is typed by the first batch it sees; a column that arrives as a dictionary in one record and
    as plain strings in the next is an error, not a silent re-encoding."""
    rng = np.random.default_rng(8105)
    b1 = plain_batch(rng, 500)
    b2 = b1.set_column(0, "name", b1.column(0).dictionary_encode().cast(pa.dictionary(pa.uint32(), pa.string())))
    plan = pp.HashAggregatePlan(None, [Count(Col("value"))], [Col("name")])
    try:
        plan.Callback(b1)
        with pytest.raises(pp.FdbError):
            plan.Callback(b2)
            plan.Finish()
    finally:
        plan.Close()


# ---- aggregations over a DynamicColumn: max(foo) over every foo.* column (aggregate.go:38-46, :306-336) ---------------------------

def _final_over_partials(pp, filt, aggs, groups, chains):
    """Partial plans (one per chain) → their records into a final-stage plan, like Synchronizer + HashAggregate(final)."""
    partials = []
    for recs in chains:
        p = pp.HashAggregatePlan(filt, aggs, groups)
        try:
            for r in recs:
                p.Callback(r)
            partials.append(p.Finish())
        finally:
            p.Close()
    f = pp.HashAggregatePlan(None, aggs, groups, final_stage=True)
    try:
        for r in partials:
            if r.num_rows:
                f.Callback(r)
        return f.Finish()
    finally:
        f.Close()


def test_golden_dynamic_column_aggregation(pp):
    """Test_Aggregation_DynCol (root aggregate_test.go:436-519): one record per column foo.bar / foo.baz / foo.bah plus one with
    all three, `max(DynCol("foo"))` without grouping → 1 row, 3 columns. The partial stage names them after the fields, the final
    stage max(<field>) (aggregate.go:313-334)."""
    from frostdb_amd.logicalplan import DynCol as D
    recs = [pa.RecordBatch.from_arrays([pa.array([7], type=pa.int64())], names=["foo.bar"]),
            pa.RecordBatch.from_arrays([pa.array([9], type=pa.int64())], names=["foo.baz"]),
            pa.RecordBatch.from_arrays([pa.array([3], type=pa.int64())], names=["foo.bah"]),
            pa.RecordBatch.from_arrays([pa.array([5]), pa.array([11]), pa.array([1])], names=["foo.bar", "foo.baz", "foo.bah"])]
    aggs = [Max(D("foo"))]
    assert arrow_to_pydict(_final_over_partials(pp, None, aggs, [], [recs])) == {"max(foo.bar)": [7], "max(foo.baz)": [11], "max(foo.bah)": [3]}
    assert arrow_to_pydict(_final_over_partials(pp, None, aggs, [], [recs[:2], recs[2:]])) == run_oracle(recs, None, aggs, [], nchains=1)
    p = pp.HashAggregatePlan(None, aggs, [])
    try:
        assert p.Draw().startswith("HashAggregate ( by )")
        for r in recs:
            p.Callback(r)
        assert p.num_groups() == 1
        assert arrow_to_pydict(p.Finish()) == {"foo.bar": [7], "foo.baz": [11], "foo.bah": [3]}
    finally:
        p.Close()
    # a record without any column of the dynamic set (aggregate.go:366-380)
    p = pp.HashAggregatePlan(None, aggs, [])
    try:
        with pytest.raises(pp.FdbError) as e:
            p.Callback(pa.RecordBatch.from_arrays([pa.array([1])], names=["other"]))
        assert e.value.code == pp.FDB_ERR_NOT_FOUND
    finally:
        p.Close()


def test_dynamic_column_aggregations_vs_oracle(pp):
    """Columns of the dynamic set come and go between records; sum / min / max / count over them next to a static aggregation and a
    filter; ungrouped (the only shape the reference survives when columns appear late) and grouped with every column in every
    record; pushed and resident; two chains → final stage. NULLs inside the aggregated columns."""
    from frostdb_amd.logicalplan import DynCol as D
    rng = np.random.default_rng(8201)

    def rec(n, cols, with_labels):
        arrays, names = [], []
        if with_labels:
            arrays.append(dict_array([None if rng.random() < 0.1 else b"g%d" % k for k in rng.integers(0, 40, n)]))
            names.append("labels.g")
        for c in cols:
            if c == "m.f":
                arrays.append(pa.array(rng.uniform(-5, 5, n), mask=rng.random(n) < 0.1))
            else:
                arrays.append(pa.array(rng.integers(-100, 100, n), type=pa.int64(), mask=rng.random(n) < 0.1))
            names.append(c)
        arrays.append(pa.array(rng.integers(0, 10, n), type=pa.int64()))
        names.append("value")
        return pa.RecordBatch.from_arrays(arrays, names=names)

    filt = Col("value") > 2
    # ungrouped: the column set changes from record to record. ONE dynamic aggregation per plan, like the reference's test: with
    # several over the same set the partial stage names all their results after the field and the final stage reads the last
    # column of that name for every one of them (aggregate.go:313-334, :338-361)
    recs = [rec(5000, ["m.a"], False), rec(3000, ["m.b", "m.f"], False), rec(20_000, ["m.a", "m.f"], False), rec(10, ["m.c"], False)]
    for fn in (Sum, Min, Max, Count):
        aggs = [fn(D("m")), Count(Col("value"))]
        want = run_oracle(recs, filt, aggs, [], nchains=1)
        assert len(want) == 5
        for chains in ([recs], [recs[:2], recs[2:]]):
            got = arrow_to_pydict(_final_over_partials(pp, filt, aggs, [], chains))
            assert sorted(got) == sorted(want)
            for k in want:
                if isinstance(want[k][0], float):
                    assert got[k] == pytest.approx(want[k], rel=1e-9), k
                else:
                    assert got[k] == want[k], k
    # a filter that selects nothing in the only record carrying m.c: that record never reaches the aggregate, no column for it
    none_of_c = And(filt, Or(Col("value") < 9, Col("value") > 9))  # (all rows of the other records keep some selected rows)
    tiny = pa.RecordBatch.from_arrays([pa.array([1, 2], type=pa.int64()), pa.array([0, 1], type=pa.int64())], names=["m.z", "value"])
    want = run_oracle(recs[:1] + [tiny], none_of_c, [Max(D("m"))], [], nchains=1)
    assert sorted(want) == ["max(m.a)"]
    assert arrow_to_pydict(_final_over_partials(pp, none_of_c, [Max(D("m"))], [], [recs[:1] + [tiny]])) == want
    # grouped: every record carries every column (a record lacking one that creates a new group panics in the reference, :413-417)
    grecs = [rec(30_000, ["m.a", "m.f"], True), rec(20_000, ["m.a", "m.f"], True)]
    for fn in (Sum, Max):
        gaggs = [fn(D("m")), Count(Col("value"))]
        name = fn(Col("x")).Name()[:3]
        want = run_oracle(grecs, filt, gaggs, [DynCol("labels")], nchains=1)
        got_rec = _final_over_partials(pp, filt, gaggs, [DynCol("labels")], [grecs[:1], grecs[1:]])
        cols = ["labels.g", f"{name}(m.a)", f"{name}(m.f)", "count(value)"]
        assert sorted(got_rec.schema.names) == sorted(cols)
        assert_same_result(arrow_to_pydict(got_rec), want, cols, float_cols={"sum(m.f)"})
    # resident records + same-stage merge of two chains: the partial stage's naming (the fields themselves)
    gaggs = [Sum(D("m")), Count(Col("value"))]
    want = run_oracle(grecs, filt, gaggs, [DynCol("labels")], nchains=1)
    p1, p2 = pp.HashAggregatePlan(filt, gaggs, [DynCol("labels")]), pp.HashAggregatePlan(filt, gaggs, [DynCol("labels")])
    keep = [pp.ResidentBatch(r) for r in grecs]
    try:
        p1.Callback(keep[0]); p2.CallbackResident(keep[1:])
        p1.Merge(p2)
        got = arrow_to_pydict(p1.Finish())
    finally:
        p1.Close(); p2.Close()
    renamed = {"labels.g": want["labels.g"], "m.a": want["sum(m.a)"], "m.f": want["sum(m.f)"], "count(value)": want["count(value)"]}
    assert sorted(got) == sorted(renamed)
    assert_same_result(got, renamed, list(renamed), float_cols={"m.f"})


def test_dynamic_column_aggregation_over_hash_tables(pp):
    """The family's join at Finish when main and children are hash tables (12 label columns + an int64 key whose 0 ≡ NULL + a plain
    string key; results come back in the shared pinned block): a dynamic MAX and a static SUM vs the oracle, two chains merged
    through a final-stage plan."""
    from frostdb_amd.logicalplan import DynCol as D
    rng = np.random.default_rng(8202)

    def rec(n):
        b = many_label_batch(rng, n, 12, 3, n_groups=4000)
        b = b.append_column("bucket", pa.array(rng.integers(0, 3, n) * 500, type=pa.int64(), mask=rng.random(n) < 0.05))
        b = b.append_column("host", pa.array(["h%d" % k for k in rng.integers(0, 5, n)], type=pa.string(), mask=rng.random(n) < 0.05))
        b = b.append_column("m.a", pa.array(rng.integers(-50, 50, n), type=pa.int64()))
        return b.append_column("m.b", pa.array(rng.uniform(-1, 1, n)))

    recs = [rec(40_000), rec(30_000)]
    aggs = [Max(D("m")), Sum(Col("value"))]
    groups = [DynCol("labels"), Col("bucket"), Col("host")]
    want = run_oracle(recs, None, aggs, groups, nchains=1)
    assert len(want["sum(value)"]) > 10_000
    got = arrow_to_pydict(_final_over_partials(pp, None, aggs, groups, [recs[:1], recs[1:]]))
    # int64 key 0 ≡ NULL: which of the two is printed depends on arrival order — fold for comparison
    for d in (got, want):
        d["bucket"] = [0 if v is None else v for v in d["bucket"]]
    cols = key_cols_of(recs, extra=("bucket", "host")) + ["max(m.a)", "max(m.b)", "sum(value)"]
    assert sorted(got) == sorted(cols)
    assert_same_result(got, want, cols)


# ---- pre-aggregate Projection fused into the scan (project.go:73-399) ----------------------------------------

def _gpu_runner(pp):
    def make(filter_expr, aggs, groups):
        def run(recs):
            return run_gpu(pp, recs, filter_expr, aggs, groups)
        return run
    return make


@pytest.mark.parametrize("case", G.MATH_CASES, ids=[c["id"] for c in G.MATH_CASES])
def test_golden_math_projection(pp, case):
    """The reference's arithmetic vectors (logictest/testdata/exec/aggregate/math) through the fused operator: computed
    aggregate inputs (`sum(<expr>) group by timestamp`) and computed group keys (`count(value) group by <expr>`)."""
    from tests.test_oracle_golden import run_math_case
    run_math_case(_gpu_runner(pp), case)


@pytest.mark.parametrize("case", G.WINDOW_CASES, ids=[c["id"] for c in G.WINDOW_CASES])
def test_golden_window_fused_projection(pp, case):
    """window:13-63 with `(timestamp/bucket)*bucket as timestamp_bucket` computed inside the scan kernel."""
    bucket = (Col("timestamp") / case["bucket"] * case["bucket"]).Alias("timestamp_bucket")
    groups = [bucket if g.name == "timestamp_bucket" else g for g in case["groups"]]
    d = run_gpu(pp, table_records(G.WINDOW_TABLE), None, case["aggs"], groups)
    assert sorted(batch_rows(d, case["out"]), key=sort_key) == sorted(case["expected"], key=sort_key), case["cite"]


def test_projection_dense_and_hash_vs_oracle(pp):
    """Computed aggregate inputs on the dense path (two label columns) and computed int64 keys on the hash path, random data,
    int64 and float64 arithmetic, division by zero (value == 0 occurs), a filter in front; checked against the oracle."""
    rng = np.random.default_rng(91)
    batches = [many_label_batch(rng, 60_000, 2, 5, int_key=True), many_label_batch(rng, 45_001, 2, 5, int_key=True)]
    V, F, B = Col("value"), Col("floatvalue"), Col("bucket")
    aggs = [Sum(V * B), Count(V), Min(V * V - B), Max(B / V), Sum(F * 2.5), Max(F / F), Min((F + 1.0) * F), Sum(V)]
    filt = V > -50
    for groups, float_cols in (([DynCol("labels")], {"sum(floatvalue * 2.5)"}),
                               ([Col("labels.l00"), (B / 7000 * 7000).Alias("b7")], {"sum(floatvalue * 2.5)"}),
                               ([(V / 10).Alias("tens"), (B - B).Alias("zero")], {"sum(floatvalue * 2.5)"})):
        want = run_oracle(batches, filt, aggs, groups)
        got = run_gpu(pp, batches, filt, aggs, groups, resident=True)
        keys = [g.name for g in groups if not g.dynamic] if not any(g.dynamic for g in groups) else ["labels.l00", "labels.l01"]
        cols = keys + [a.Name() for a in aggs]
        # an int64 key of 0 and a NULL key are one group in the reference (hash 0 for both); which of the two is printed depends on
        # arrival order, so compare with NULL folded into 0 for computed keys
        for d in (want, got):
            for k in keys:
                if not k.startswith("labels."):
                    d[k] = [0 if v is None else v for v in d[k]]
        assert_same_result(got, want, cols, float_cols=float_cols)


def test_uint64_arithmetic_projections_vs_oracle(pp):
    """binaryExprProjection over *array.Uint64 (project.go:138-150, :335-395): + − × wrap modulo 2^64, the quotient is unsigned (values
    beyond 2^63 stay positive), a zero divisor is a NULL; the computed column is a uint64 group key. A uint64 column compared with a
    uint64 literal inside a boolean projection goes the way a filter leaf does (unsigned). What the reference refuses is refused:
    aggregating a uint64 expression (aggregate.go:736), mixing uint64 with an int64 scalar (the type assertion at project.go:139-147),
    comparing computed uint64 values. The oracle restates the four Go loops; no reference vector exists for them (parity unpinned)."""
    rng = np.random.default_rng(93)
    n = 40_000
    a = rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64)
    small = rng.integers(0, 6, n).astype(np.uint64)
    recs = []
    for lo, hi in ((0, 25_000), (25_000, n)):
        recs.append(pa.RecordBatch.from_arrays([pa.array(a[lo:hi] >> np.uint64(50), mask=rng.random(hi - lo) < 0.1), pa.array(small[lo:hi]),
                                                pa.array(a[lo:hi]), pa.array(rng.integers(0, 100, hi - lo))], names=["a", "b", "big", "value"]))
    A, B, BIG, V = Col("a"), Col("b"), Col("big"), Col("value")
    aggs = [Sum(V), Count(V)]
    for groups in ([(A / B).Alias("q")],                                   # NULL where b == 0
                   [(B * B - B).Alias("w")],
                   [(BIG / UInt64(1 << 61)).Alias("top"), (B - UInt64(3)).Alias("wrapped")],   # quotients of values ≥ 2^63; 0 − 3 wraps to 2^64 − 3
                   # (one computed key per case from here on: the reference's group identity is the 64-bit combination of the columns'
                   # IDENTITY hashes — two small-integer keys collide there, e.g. (6286, 1) and (6578, 2), and the oracle restates that)
                   [(A + A).Alias("s")], [(B / UInt64(2)).Alias("h")],
                   [A > UInt64(5000)]):
        want = run_oracle(recs, V >= 0, aggs, groups)
        got = run_gpu(pp, recs, V >= 0, aggs, groups, resident=True)
        keys = [g.name for g in groups]
        for d in (want, got):  # (a key of 0 and a NULL key are one group, as for computed int64 keys)
            for k in keys:
                d[k] = [0 if v is None else v for v in d[k]]
        assert_same_result(got, want, keys + [x.Name() for x in aggs])
        if "wrapped" in keys:
            assert 2**64 - 3 in got["wrapped"] and max(got["wrapped"]) == 2**64 - 1 and max(got["top"]) >= 4
    for bad_aggs, bad_groups, code in (([Sum(A + A)], [B], pp.FDB_ERR_UNSUPPORTED), ([Sum(V)], [(A + 5).Alias("x")], pp.FDB_ERR_INVALID),
                                       ([Sum(V)], [BinaryExpr(A + A, OP_GT, Literal(UInt64(3))).Alias("c")], pp.FDB_ERR_UNSUPPORTED)):
        with pytest.raises(pp.FdbError) as e:
            run_gpu(pp, recs, None, bad_aggs, bad_groups)
        assert e.value.code == code, str(e.value)
        with pytest.raises(Exception):
            run_oracle(recs, None, bad_aggs, bad_groups)


def test_a_new_shape_is_interpreted_while_its_kernel_is_built(pp, monkeypatch, tmp_path):
    """$FDB_JIT_ASYNC=1: the first query of a shape does not wait for hiprtc (130–500 ms) — its scan runs on the interpreting slot
    kernel while a background thread builds the specialised one, and a later query of the shape finds it. Same results either way.
    (A private, empty cache directory: nothing can come from the disk.)"""
    import os
    import time
    os.chmod(tmp_path, 0o700)
    monkeypatch.setenv("FDB_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("FDB_JIT_ASYNC", "1")
    rng = np.random.default_rng(1234)
    recs = [make_prometheus_batch(rng, 30_011, n_path=37), make_prometheus_batch(rng, 20_003, n_path=37)]
    # (a shape no other test of the process uses: these four aggregates behind this three-leaf predicate)
    filt = And(Or(Col("labels.code") == "200", Col("labels.method") != "PUT"), Col("value") <= 777.25)
    aggs, groups = [Max(Col("timestamp")), Min(Col("value")), Count(Col("value")), Sum(Col("timestamp"))], [Col("labels.path")]
    want = run_oracle(recs, filt, aggs, groups)
    cols = ["labels.path"] + [a.Name() for a in aggs]
    before = pp.jit_stats()["compiled"]

    def query():
        plan = pp.HashAggregatePlan(filt, aggs, groups)
        rbs = [pp.ResidentBatch(r) for r in recs]
        try:
            plan.CallbackResident(rbs)
            out = arrow_to_pydict(plan.Finish())
            return out, plan.last_kernel()
        finally:
            plan.Close()
            for r in rbs:
                r.close()

    got, kernel = query()
    assert kernel == "scan_slots_kernel", kernel   # interpreted: the kernel of this shape did not exist
    assert_same_result(got, want, cols)
    deadline = time.time() + 60
    while pp.jit_stats()["compiled"] == before and time.time() < deadline:
        time.sleep(0.02)
    assert pp.jit_stats()["compiled"] > before
    for _ in range(200):  # (compiled, then loaded: the entry is published a moment later)
        got2, kernel2 = query()
        if kernel2 == "fdb_plan_kernel":
            break
        time.sleep(0.02)
    assert kernel2 == "fdb_plan_kernel", kernel2
    assert_same_result(got2, want, cols)
    assert any(f.endswith(".hsaco") for f in os.listdir(tmp_path))
    # without the switch the same first query waits for its kernel
    monkeypatch.delenv("FDB_JIT_ASYNC")
    filt2 = And(Or(Col("labels.code") == "500", Col("labels.method") != "GET"), Col("value") <= 12.5, Col("timestamp") > 5)
    plan = pp.HashAggregatePlan(filt2, aggs, groups)
    rb = pp.ResidentBatch(recs[0])
    try:
        plan.CallbackResident([rb])
        plan.Finish()
        assert plan.last_kernel() == "fdb_plan_kernel"
    finally:
        plan.Close()
        rb.close()


def test_projection_needs_specialised_kernel(pp, monkeypatch):
    monkeypatch.setenv("FDB_NO_JIT", "1")
    rng = np.random.default_rng(92)
    b = many_label_batch(rng, 1000, 2, 3, int_key=True)
    plan = pp.HashAggregatePlan(None, [Sum(Col("value") * Col("bucket"))], [DynCol("labels")])
    try:
        with pytest.raises(pp.UnsupportedError):
            plan.Callback(b)
    finally:
        plan.Close()
    plan = pp.HashAggregatePlan(None, [Sum(Col("value") * Col("floatvalue"))], [DynCol("labels")])  # int64 * float64: the reference panics
    monkeypatch.delenv("FDB_NO_JIT")
    try:
        with pytest.raises(pp.FdbError):
            plan.Callback(b)
    finally:
        plan.Close()


def test_config1_simple_schema(pp, variant):
    """BASELINE.json configs[0] (examples/simple schema: utf8 dictionaries, names.middle_name only in some records, int64 value):
    `names.first_name == 'Frederic'` + SUM(value), ungrouped, by surname, and by every names.* column — vs the oracle."""
    from tests.util import make_simple_batches
    batches = make_simple_batches(np.random.default_rng(1), 10_000, 3)
    f = Col("names.first_name") == "Frederic"
    for aggs, groups, cols in (([Sum(Col("value")), Count(Col("value"))], [], ["sum(value)", "count(value)"]),
                               ([Sum(Col("value"))], [Col("names.surname")], ["names.surname", "sum(value)"]),
                               ([Sum(Col("value")), Max(Col("value"))], [DynCol("names")],
                                ["names.first_name", "names.surname", "names.middle_name", "sum(value)", "max(value)"])):
        want = run_oracle(batches, f, aggs, groups)
        got = run_gpu(pp, batches, f, aggs, groups)
        assert_same_result(got, want, cols)


# ---- Distinct = a plan without aggregations (distinct.go:21-170) ---------------------------------------------

@pytest.mark.parametrize("case", G.DISTINCT_CASES, ids=[c["id"] for c in G.DISTINCT_CASES])
def test_golden_distinct(pp, case, variant):
    d = run_gpu(pp, table_records(G.DISTINCT_TABLE), case["filter"], [], case["groups"])
    assert sorted(batch_rows(d, case["out"]), key=sort_key) == sorted(case["expected"], key=sort_key), case["cite"]


@pytest.mark.parametrize("case", G.DISTINCT_PROJ_CASES, ids=[c["id"] for c in G.DISTINCT_PROJ_CASES])
def test_golden_distinct_bool_projection(pp, case):
    """distinct over a boolean projection (`timestamp > 0` as a key; boolExprProjection, project.go:401-470): every row gets a
    valid bool — computed per row in the hash kernel, emitted as an Arrow bool column."""
    d = run_gpu(pp, table_records(case["table"]), None, [], case["groups"])
    rows = [tuple(0 if (v is None and c == "timestamp") else v for c, v in zip(case["out"], r)) for r in batch_rows(d, case["out"])]
    assert sorted(rows, key=sort_key) == sorted(case["expected"], key=sort_key), case["cite"]


def test_bool_projection_over_string_and_dictionary_columns(pp):
    """boolExprProjection evaluates its expression like a filter does (project.go:409-447 → BinaryScalarExpr.Eval,
    binaryscalarexpr.go:41-152): `labels.l00 == 'v0_1'` as a distinct / group key is a per-dictionary-entry truth table (one more leaf of
    the record's argument block, outside the filter program) whose match bit is the row's bool — never NULL: a NULL label compares
    false. Dictionary and plain string columns, == and !=, a column the record lacks (the missing-column rules), AND / OR with a numeric
    comparison, `if (labels… == …)` as an aggregate's input, records with different dictionaries; against the oracle."""
    from frostdb_amd.logicalplan import IfExpr
    rng = np.random.default_rng(9911)
    batches = []
    for n, ncols in ((60_000, 3), (25_000, 4)):
        b = many_label_batch(rng, n, ncols, 4, n_groups=400, int_key=True)
        user = pa.array(["u%d" % k for k in rng.integers(0, 5, n)], type=pa.string(), mask=rng.random(n) < 0.05)
        batches.append(b.append_column("user", user))
    first_vals = sorted({v for b in batches for v in b.column(b.schema.get_field_index("labels.l00")).dictionary.to_pylist()})
    keys = (Col("labels.l00") == first_vals[1].decode(), Col("labels.l00") != first_vals[0].decode(), Col("user") == "u3", Col("user") > "u1",
            Col("labels.l03") == "nope", Col("labels.l03") != "nope", Col("labels.l00") == None,  # noqa: E711
            And(Col("labels.l01") == first_vals[0].decode().replace("v0_", "v1_"), Col("value") > 0),
            Or(Col("user") == "u0", BinaryExpr(Col("bucket"), OP_LT_EQ, Literal(1000))))
    aggs = [Sum(Col("value")), Count(Col("value"))]
    for key in keys:
        groups = [Col("labels.l02"), key]
        want = run_oracle(batches, None, aggs, groups, nchains=2)
        got = run_gpu(pp, batches, None, aggs, groups)
        assert_same_result(got, want, ["labels.l02", key.name] + [a.Name() for a in aggs])
        assert set(want[key.name]) <= {True, False} and None not in got[key.name]
    # distinct over the projection alone, resident records
    for key in keys[:4]:
        want = run_oracle(batches, None, [], [key])
        got = run_gpu(pp, batches, None, [], [key], resident=True)
        assert sorted(got[key.name]) == sorted(want[key.name])
    # the comparison as the condition of an if: sum(if (user == 'u2') value else 0)
    cond_sum = [Sum(IfExpr(Col("user") == "u2", Col("value"), Literal(0)))]
    for groups in ([Col("labels.l01")], [DynCol("labels"), Col("bucket")]):
        want = run_oracle(batches, None, cond_sum, groups)
        got = run_gpu(pp, batches, None, cond_sum, groups, resident=True)
        assert_same_result(got, want, (key_cols_of(batches, extra=("bucket",)) if len(groups) == 2 else ["labels.l01"]) + [cond_sum[0].Name()])


def test_bool_projection_key_at_scale(pp):
    """A comparison as a group key next to label columns: NULLs in the compared column, int64 vs float64 operands, an arithmetic
    operand, AND / OR, aggregations beside it, two chains merged — vs the oracle."""
    from frostdb_amd import distributed as fd
    rng = np.random.default_rng(9901)
    batches = []
    for n in (70_000, 30_000):
        b = many_label_batch(rng, n, 3, 4, n_groups=500, int_key=True)
        ts = pa.array(rng.integers(-5, 6, n), type=pa.int64(), mask=rng.random(n) < 0.1)
        batches.append(b.append_column("timestamp", ts))
    keys = (Col("timestamp") > 0, BinaryExpr(Col("timestamp") * 10000, OP_LT_EQ, Col("bucket")), Col("floatvalue") >= 5,
            BinaryExpr(Col("timestamp") + Col("value"), OP_NOT_EQ, Literal(3)),
            And(Col("timestamp") > -2, Or(Col("floatvalue") < 2.5, Col("value") == 7)))
    for key in keys:
        groups = [DynCol("labels"), key]
        aggs = [Sum(Col("value")), Count(Col("value"))]
        want = run_oracle(batches, None, aggs, groups, nchains=2)
        p1 = pp.HashAggregatePlan(None, aggs, groups)
        p2 = pp.HashAggregatePlan(None, aggs, groups)
        try:
            p1.Callback(batches[0]); p2.Callback(batches[1])
            p1.Merge(p2)
            got = arrow_to_pydict(p1.Finish())
        finally:
            p1.Close(); p2.Close()
        assert set(want[key.name]) == {True, False}
        cols = key_cols_of(batches) + [key.name] + [a.Name() for a in aggs]
        assert_same_result(got, want, cols)
        # the same through the hash-partitioned exchange: schema agreement carries the bool column, two owners import their partitions
        plans = [pp.HashAggregatePlan(None, aggs, groups), pp.HashAggregatePlan(None, aggs, groups)]
        shards = [plans[0].clone_empty(), plans[0].clone_empty()]
        try:
            for p, b in zip(plans, batches):
                p.Callback(b)
            schema = fd.unify_group_schemas([fd._schema_to_obj(p.group_schema()) for p in plans])
            assert schema.schema.field(key.name).type == pa.bool_()
            for sh in shards:
                sh.seed_groups(schema)
            for p in plans:
                ptr, counts, row_bytes = p.hash_export(shards[0], 2)
                shards[0].hash_import(ptr, counts[0])
                shards[1].hash_import(ptr + counts[0] * row_bytes, counts[1])
            got2 = _concat_results([sh.Finish() for sh in shards])
        finally:
            for p in plans + shards:
                p.Close()
        assert_same_result(got2, want, cols)


def test_distinct_at_scale_dense_and_hash(pp):
    """Distinct label tuples of random data: dense table (2 columns), hash table (12 columns + an int64 key), with a filter,
    two chains merged — vs the oracle."""
    rng = np.random.default_rng(515)
    for n_cols, n_groups, int_key in ((2, None, False), (12, 30_000, True)):
        batches = [many_label_batch(rng, 80_000, n_cols, 4, n_groups=n_groups, int_key=int_key), many_label_batch(rng, 50_000, n_cols, 4, n_groups=n_groups, int_key=int_key)]
        groups = [DynCol("labels")] + ([Col("bucket")] if int_key else [])
        filt = Col("value") >= 0
        want = run_oracle(batches, filt, [], groups, nchains=2)
        p1 = pp.HashAggregatePlan(filt, [], groups)
        p2 = pp.HashAggregatePlan(filt, [], groups)
        try:
            assert "Distinction (labels" in p1.Draw()
            p1.Callback(batches[0]); p2.Callback(batches[1])
            p1.Merge(p2)
            got = arrow_to_pydict(p1.Finish())
        finally:
            p1.Close(); p2.Close()
        cols = key_cols_of(batches, extra=("bucket",) if int_key else ())
        assert_same_result(got, want, cols)


def test_dense_table_too_big_for_lds(pp, variant):
    """Group by (path × instance × code): 2 001 × 17 × 7 ≈ 238 k dense slots — the table does not fit in LDS, so rows go to the global
    table with atomics; the specialised kernel puts a per-workgroup combining cache in front (hot keys: path is Zipf-like here).
    With and without a COUNT aggregation (the count array is only an occupancy flag in the second case)."""
    rng = np.random.default_rng(77)
    batches = []
    for n in (120_000, 90_001):
        b = make_prometheus_batch(rng, n, n_path=2000)
        hot = pa.array(np.minimum(rng.zipf(1.3, size=n) - 1, 1999).astype(np.uint32), type=pa.uint32(), mask=rng.random(n) < 0.01)
        path = pa.DictionaryArray.from_arrays(hot, b.column(1).dictionary)
        batches.append(b.set_column(1, "labels.path", path))
    groups = [Col("labels.path"), Col("labels.instance"), Col("labels.code")]
    for aggs in ([Count(Col("value")), Min(Col("timestamp")), Max(Col("value")), Sum(Col("value"))], [Sum(Col("value")), Min(Col("value"))]):
        want = run_oracle(batches, Col("labels.method") != "PUT", aggs, groups)
        plan = pp.HashAggregatePlan(Col("labels.method") != "PUT", aggs, groups)
        keep = [pp.ResidentBatch(x) for x in batches]
        try:
            plan.CallbackResident(keep)
            assert plan.last_kernel() == ("fdb_plan_kernel" if variant == "specialised" else "scan_dense_kernel") or variant == "interpreted"
            got = arrow_to_pydict(plan.Finish())
        finally:
            plan.Close()
        cols = ["labels.path", "labels.instance", "labels.code"] + [a.Name() for a in aggs]
        assert_same_result(got, want, cols, float_cols={"sum(value)"})


def test_full_size_configs_2_and_3_properties(pp):
    """BASELINE.json's full single-GPU size — 100 M synthetic Prometheus rows in 4 resident records, one launch per query — checked
    through size-independent properties against numpy on the raw columns: Σ count = selected rows, per-path counts and sums
    (cfg 2), min/max of the selected timestamps, group sums add up to the total within 1e-9 (cfg 3)."""
    from concurrent.futures import ThreadPoolExecutor
    from frostdb_amd import synth
    n_rec, rows = 4, 25_000_000
    with ThreadPoolExecutor(4) as ex:
        recs = list(ex.map(lambda i: synth.prometheus_chunk(0, i, rows, row_base=i * rows, cfg3=True), range(n_rec)))
    keep = [pp.ResidentBatch(r) for r in recs]

    def np_idx(col):
        return col.indices.fill_null(len(col.dictionary)).to_numpy(zero_copy_only=False).astype(np.int64)

    code = np.concatenate([np_idx(r.column(0)) for r in recs])
    path = np.concatenate([np_idx(r.column(1)) for r in recs])
    method = np.concatenate([np_idx(r.column(2)) for r in recs])
    inst_valid = np.concatenate([np.asarray(r.column(3).is_valid()) for r in recs])
    ts = np.concatenate([r.column(4).to_numpy() for r in recs])
    val = np.concatenate([r.column(5).to_numpy() for r in recs])
    n_path = len(recs[0].column(1).dictionary)
    try:
        # cfg 2
        plan = pp.HashAggregatePlan(CFG2["filter_expr"], [Sum(Col("value")), Count(Col("value"))], CFG2["groups"])
        plan.CallbackResident(keep)
        assert plan.last_kernel() == "fdb_plan_kernel"
        d = arrow_to_pydict(plan.Finish())
        plan.Close()
        sel = code == synth.CODES.index(b"200")
        want_cnt = np.bincount(path[sel], minlength=n_path + 1)
        want_sum = np.bincount(path[sel], weights=val[sel], minlength=n_path + 1)
        names = synth.PATHS + [None]
        got = {p: (c, s) for p, c, s in zip(d["labels.path"], d["count(value)"], d["sum(value)"])}
        assert sum(c for c, _ in got.values()) == int(sel.sum())
        for i, p in enumerate(names):
            if want_cnt[i] == 0:
                assert p not in got
            else:
                assert got[p][0] == want_cnt[i] and math.isclose(got[p][1], want_sum[i], rel_tol=REL_TOL), p
        # cfg 3
        plan = pp.HashAggregatePlan(CFG3["filter_expr"], CFG3["aggs"], CFG3["groups"])
        plan.CallbackResident(keep)
        d = arrow_to_pydict(plan.Finish())
        plan.Close()
        sel = ((code == synth.CODES.index(b"200")) | (code == synth.CODES.index(b"500"))) & (method == synth.METHODS.index(b"GET")) & inst_valid
        assert sum(d["count(value)"]) == int(sel.sum())
        assert min(d["min(timestamp)"]) == int(ts[sel].min()) and max(d["max(timestamp)"]) == int(ts[sel].max())
        assert math.isclose(math.fsum(d["sum(value)"]), math.fsum(val[sel]), rel_tol=REL_TOL)
        want_cnt = np.bincount(path[sel], minlength=n_path + 1)
        got_cnt = dict(zip(d["labels.path"], d["count(value)"]))
        assert all(got_cnt.get(p, 0) == want_cnt[i] for i, p in enumerate(names))
    finally:
        for k in keep:
            k.close()


def test_concurrent_chains_on_threads(pp):
    """N chains = N plans driven from N OS threads at once (≙ the reference's N scan workers calling Callback concurrently, one
    chain each, then Finish on all chains concurrently — table.go:783-860, physicalplan.go:157-165; ctypes drops the GIL during the
    C calls). Then Synchronizer + final stage = fdb_plan_merge. Dense and hash tables, several rounds to shake out races in the
    shared caches (contexts, compiled kernels, dictionaries, pinned result blocks)."""
    import threading
    rng = np.random.default_rng(4242)
    n_chains = 6
    for make, filt, aggs, groups, cols in (
            (lambda: make_prometheus_batch(rng, 30_000, n_path=100), CFG3["filter_expr"], CFG3["aggs"], CFG3["groups"],
             ["labels.path"] + [a.Name() for a in CFG3["aggs"]]),
            (lambda: many_label_batch(rng, 20_000, 10, 3, n_groups=3000), None, [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")], None)):
        for _ in range(3):
            shards = [[make() for _ in range(3)] for _ in range(n_chains)]
            want = run_oracle([b for s in shards for b in s], filt, aggs, groups, nchains=n_chains)
            plans = [pp.HashAggregatePlan(filt, aggs, groups) for _ in range(n_chains)]
            errors = []

            def work(i):
                try:
                    for b in shards[i]:
                        plans[i].Callback(b)
                    plans[i].num_groups()
                except Exception as e:  # noqa: BLE001
                    errors.append(e)

            threads = [threading.Thread(target=work, args=(i,)) for i in range(n_chains)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            try:
                assert not errors, errors
                for p in plans[1:]:
                    plans[0].Merge(p)
                got = arrow_to_pydict(plans[0].Finish())
            finally:
                for p in plans:
                    p.Close()
            c = cols or (key_cols_of([b for s in shards for b in s]) + [a.Name() for a in aggs])
            assert_same_result(got, want, c, float_cols={"sum(value)"})


def test_projection_with_schema_drift_between_queued_records(pp):
    """Small host records are queued and scanned together; records of different SHAPES (the filter column is missing in one —
    missing-column rules turn its leaf into a constant) cannot share one specialised kernel, and a plan with computed columns
    has no interpreting fallback: such a launch is split per record instead of failing."""
    rng = np.random.default_rng(93)
    b1 = many_label_batch(rng, 5000, 3, 4, int_key=True)
    b2 = many_label_batch(rng, 4000, 3, 4, int_key=True).drop_columns(["labels.l02"])
    b3 = many_label_batch(rng, 3000, 3, 4, int_key=True)
    filt = Col("labels.l02") != "v2_1"
    aggs = [Sum(Col("value") * Col("bucket")), Count(Col("value"))]
    groups = [Col("labels.l00")]
    want = run_oracle([b1, b2, b3], filt, aggs, groups)
    got = run_gpu(pp, [b1, b2, b3], filt, aggs, groups)
    assert_same_result(got, want, ["labels.l00"] + [a.Name() for a in aggs])


def test_maximum_group_columns(pp, variant):
    """64 group-by columns is the limit of the key tuple's valid mask: 64 work (hash path), 65 are refused loudly."""
    rng = np.random.default_rng(64)
    b = many_label_batch(rng, 20_000, 64, 2, n_groups=3000)
    aggs = [Sum(Col("value")), Count(Col("value"))]
    want = run_oracle([b], None, aggs, [DynCol("labels")])
    got = run_gpu(pp, [b], None, aggs, [DynCol("labels")])
    assert_same_result(got, want, key_cols_of([b]) + [a.Name() for a in aggs])
    b65 = many_label_batch(rng, 1000, 65, 2, n_groups=100)
    plan = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
    try:
        with pytest.raises(pp.UnsupportedError):
            plan.Callback(b65)
    finally:
        plan.Close()


def test_large_dictionary_columns(pp, variant):
    """A label with 60 000 distinct values: its predicate truth table (regex → byte LUT) and its key-id LUT are too big for LDS and
    are gathered from global memory; the dense table (60 001 × 3 slots) does not fit LDS either. Second query: group by it alone."""
    rng = np.random.default_rng(606)
    n, card = 150_000, 60_000
    names = [b"series-%06d" % i for i in range(card)]
    recs = []
    for _ in range(2):
        idx = pa.array(rng.integers(0, card, size=n).astype(np.uint32), type=pa.uint32(), mask=rng.random(n) < 0.02)
        small = pa.array(rng.integers(0, 2, size=n).astype(np.uint32), type=pa.uint32())
        recs.append(pa.RecordBatch.from_arrays(
            [pa.DictionaryArray.from_arrays(idx, pa.array(names, type=pa.binary())),
             pa.DictionaryArray.from_arrays(small, pa.array([b"x", b"y"], type=pa.binary())),
             pa.array(rng.integers(0, 1000, size=n).astype(np.int64)), pa.array(rng.uniform(0, 1, size=n))],
            names=["labels.series", "labels.kind", "value", "floatvalue"]))
    filt = And(Col("labels.series").RegexMatch("series-0[0-3]"), Col("labels.kind") != "y")
    for aggs, groups in (([Sum(Col("value")), Count(Col("value")), Max(Col("floatvalue"))], [Col("labels.series"), Col("labels.kind")]),
                         ([Sum(Col("floatvalue"))], [Col("labels.series")])):
        want = run_oracle(recs, filt, aggs, groups)
        got = run_gpu(pp, recs, filt, aggs, groups, resident=True)
        assert_same_result(got, want, [g.name for g in groups] + [a.Name() for a in aggs], float_cols={"sum(floatvalue)"})


def test_mixed_shape_parts_in_one_scan(pp):
    """Resident parts of one table whose shapes differ — a predicate value missing from one part's dictionary (its leaf folds to
    a constant), a label column absent from another, NULL-free columns in a third — pushed in ONE call: the scan is split per
    shape and every group still runs a specialised kernel; the result equals the oracle's."""
    rng = np.random.default_rng(808)
    parts = [make_prometheus_batch(rng, 30_000, n_path=40) for _ in range(5)]
    # part 1: no row has code 500 and its dictionary does not even contain it
    codes = parts[1].column(0)
    keep = pa.array([b"200", b"404"], type=pa.binary())
    parts[1] = parts[1].set_column(0, "labels.code", pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 2, size=30_000).astype(np.uint32)), keep))
    parts[2] = parts[2].drop_columns(["labels.instance"])
    parts[3] = make_prometheus_batch(rng, 30_000, n_path=40, null_frac=0.0)
    want = run_oracle(parts, **CFG3)
    plan = pp.HashAggregatePlan(CFG3["filter_expr"], CFG3["aggs"], CFG3["groups"])
    keep_alive = [pp.ResidentBatch(p) for p in parts]
    try:
        plan.CallbackResident(keep_alive)
        assert plan.last_kernel() == "fdb_plan_kernel"
        got = arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
    assert_same_result(got, want, ["labels.path"] + [a.Name() for a in CFG3["aggs"]], float_cols={"sum(value)"})


# ---- UniqueAggregation / AndAggregation (aggregate.go:635-732) ------------------------------------------

def test_golden_unique_and_and_aggregations(pp, variant):
    """query/engine_test.go TestUniqueAggregation (:19-74) and TestAndAggregation (:76-131), one chain and two chains merged."""
    from tests.test_oracle_golden import unique_and_cases
    for case in unique_and_cases():
        rec = case["rec"]
        d = run_gpu(pp, [rec], None, [case["agg"]], [Col("timestamp")])
        assert dict(zip(d["timestamp"], d[case["out"]])) == case["expected"], case["cite"]
        p1 = pp.HashAggregatePlan(None, [case["agg"]], [Col("timestamp")])
        p2 = pp.HashAggregatePlan(None, [case["agg"]], [Col("timestamp")])
        try:
            p1.Callback(rec.slice(0, 1)); p2.Callback(rec.slice(1))
            p1.Merge(p2)
            d = arrow_to_pydict(p1.Finish())
        finally:
            p1.Close(); p2.Close()
        assert dict(zip(d["timestamp"], d[case["out"]])) == case["expected"], case["cite"]


def test_unique_and_and_vs_oracle_at_scale(pp, variant):
    """UNIQUE over int64 with NULLs (groups with one value, several values, a NULL among equal values) and AND over a nullable bool
    column, next to ordinary aggregates, dense table (label keys) and hash table (int64 key), against the oracle; plus the
    reference's type errors."""
    from frostdb_amd.logicalplan import AndAgg, Unique
    rng = np.random.default_rng(1919)
    n = 60_000
    lab = rng.integers(0, 40, size=n)
    recs = []
    for lo in (0, n // 2):
        sl = slice(lo, lo + n // 2)
        labs = lab[sl]
        u = np.where(labs % 3 == 0, labs * 7, rng.integers(0, 3, size=n // 2) + labs)  # a third of the groups have ONE value
        umask = (rng.random(n // 2) < 0.02) & (labs % 5 == 0)                                # … some of those also see a NULL
        flag = (labs % 2 == 0) | (rng.random(n // 2) < 0.7)
        recs.append(pa.RecordBatch.from_arrays(
            [pa.DictionaryArray.from_arrays(pa.array(labs.astype(np.uint32)), pa.array([b"g%02d" % i for i in range(40)], type=pa.binary())),
             pa.array((labs * 10).astype(np.int64)), pa.array(u.astype(np.int64), mask=umask), pa.array(flag, mask=rng.random(n // 2) < 0.1),
             pa.array(rng.integers(0, 100, size=n // 2).astype(np.int64))],
            names=["labels.g", "bucket", "u", "flag", "value"]))
    aggs = [Unique(Col("u")), AndAgg(Col("flag")), Sum(Col("value")), Count(Col("value")), Unique(Col("bucket"))]
    for groups in ([Col("labels.g")], [Col("bucket")]):
        want = run_oracle(recs, Col("value") >= 5, aggs, groups)
        got = run_gpu(pp, recs, Col("value") >= 5, aggs, groups)
        assert_same_result(got, want, [g.name for g in groups] + [a.Name() for a in aggs])
        assert any(v is None for v in got["unique(u)"]) and any(v is not None for v in got["unique(u)"])
        assert any(v is False for v in got["and(flag)"]) and any(v is True for v in got["and(flag)"])
    for bad in (Unique(Col("flag")), AndAgg(Col("value"))):
        plan = pp.HashAggregatePlan(None, [bad], [Col("labels.g")])
        try:
            with pytest.raises(pp.UnsupportedError):
                plan.Callback(recs[0])
        finally:
            plan.Close()


def test_hash_mode_partial_keys_and_states_line_up(pp):
    """Slot-order compaction of a hash table: fdb_plan_partial_keys and one fdb_plan_partial_state call per aggregation each run
    their own compaction — rows must mean the same group in all of them (the cross-rank key-unification merge relies on it), and
    repeated calls must return the same order. 30 000 int64-keyed groups (hash mode), checked against Finish of a twin plan."""
    import ctypes
    rng = np.random.default_rng(123)
    n = 200_000
    rec = pa.RecordBatch.from_arrays(
        [pa.array(rng.integers(1, 30_000, n).astype(np.int64)), dict_array([b"a", b"b", None][i % 3] for i in range(n)),
         pa.array(rng.integers(-50, 50, n).astype(np.int64)), pa.array(rng.uniform(0, 1, n))],
        names=["bucket", "labels.x", "ival", "value"])
    aggs = [Sum(Col("value")), Min(Col("ival")), Count(Col("value")), Max(Col("ival"))]
    groups = [Col("bucket"), Col("labels.x")]
    plan = pp.HashAggregatePlan(None, aggs, groups)
    twin = pp.HashAggregatePlan(None, aggs, groups)
    try:
        plan.Callback(rec)
        twin.Callback(rec)
        want = arrow_to_pydict(twin.Finish())
        k1, k2 = arrow_to_pydict(plan.partial_keys()), arrow_to_pydict(plan.partial_keys())
        assert k1 == k2
        g = len(k1["bucket"])
        assert g == len(want["bucket"]) and g > 30_000
        got = dict(k1)
        for j, a in enumerate(aggs):
            buf = np.zeros(g, dtype=np.float64 if plan.agg_format(j) == "g" else np.int64)
            plan.partial_state_into(j, buf.ctypes.data, g * 8)
            got[a.Name()] = buf.tolist()
        cols = ["bucket", "labels.x"] + [a.Name() for a in aggs]
        assert_same_result(got, want, cols, float_cols={"sum(value)"})
    finally:
        plan.Close()
        twin.Close()


def test_out_of_range_dictionary_index_is_an_error_not_a_fault(pp):
    """A malformed record — a VALID row whose dictionary index is past the dictionary — is refused with FDB_ERR_INVALID at import
    (small pushed record: checked on the host; big pushed record and resident import: checked by a device pass before anything
    scans it), ≙ the reference's recovered panic (recovery/recovery.go:13-30). A NULL row may hold any index (Arrow leaves the
    slot undefined) and is accepted. The process keeps working afterwards."""
    rng = np.random.default_rng(8)

    def rec(n, bad_at=None, bad_is_null=False):
        idx = rng.integers(0, 3, n).astype(np.uint32)
        mask = np.zeros(n, dtype=bool)
        if bad_at is not None:
            idx[bad_at] = 1_000_000
            mask[bad_at] = bad_is_null
        path = pa.DictionaryArray.from_arrays(pa.array(idx, mask=mask if bad_is_null else None), pa.array([b"a", b"b", b"c"], type=pa.binary()), safe=False)
        code = pa.DictionaryArray.from_arrays(pa.array(np.zeros(n, dtype=np.uint32)), pa.array([b"200"], type=pa.binary()))
        return pa.RecordBatch.from_arrays([code, path, pa.array(rng.uniform(0, 1, n))], names=["labels.code", "labels.path", "value"])

    for n in (5_000, 3_000_000):  # via the pinned ring / direct copy + device check
        plan = pp.HashAggregatePlan(**CFG2)
        with pytest.raises(pp.FdbError) as ei:
            plan.Callback(rec(n, bad_at=n // 2))
        assert ei.value.code == pp.FDB_ERR_INVALID and "labels.path" in str(ei.value) and "out of range" in str(ei.value)
        good = rec(n, bad_at=n // 3, bad_is_null=True)  # the same wild index under a NULL: fine
        plan.Callback(good)
        d = arrow_to_pydict(plan.Finish())
        assert_same_result(d, run_oracle([good], **CFG2), ["labels.path", "sum(value)"], float_cols={"sum(value)"})
        plan.Close()
    with pytest.raises(pp.FdbError) as ei:
        pp.ResidentBatch(rec(200_000, bad_at=7))
    assert ei.value.code == pp.FDB_ERR_INVALID
    # filter / select import the record through the same code
    plan = pp.HashAggregatePlan(Col("labels.path") == "a")
    with pytest.raises(pp.FdbError):
        plan.Select(rec(10_000, bad_at=9_999))
    plan.Close()


# ---- BASELINE.json config 5 at its own shape: 32 dynamic label columns, millions of groups -----------------------------------

def _cfg5_canonical(batch):
    """A cfg 5 result in comparable form: rows sorted by group id; per label column the digit (0-3) or 255 for NULL."""
    from frostdb_amd import synth
    gid = synth.cfg5_decode_group_ids(batch)
    order = np.argsort(gid, kind="stable")
    labels = []
    for c in range(synth.CFG5_COLS):
        col = batch.column(batch.schema.get_field_index("labels.l%02d" % c))
        digit_of_entry = np.array([int(v.rsplit(b"=", 1)[1]) for v in col.dictionary.to_pylist()] + [255], dtype=np.uint8)
        idx = col.indices.fill_null(len(col.dictionary)).to_numpy(zero_copy_only=False).astype(np.int64)
        labels.append(digit_of_entry[idx][order])
    out = {"gid": gid[order], "labels": labels}
    for name in batch.schema.names:
        if not name.startswith("labels."):
            out[name] = batch.column(batch.schema.get_field_index(name)).to_numpy(zero_copy_only=False)[order]
    return out


def test_cfg5_shape_against_the_oracle_over_a_million_groups(pp):
    """cfg 5's own shape — SUM / COUNT grouped by ALL 32 labels.* columns (DynCol), ≈3 % NULL digits — with 1.2 M distinct
    groups over 3 M rows (3 resident records, so the cardinality estimate and the one-step table growth are exercised):
    every group's 32 key values (NULLs included), count and sum against the oracle."""
    from frostdb_amd import synth
    from oracle import OraclePlan
    n_groups, per = 1_200_000, 1_000_000
    recs = [synth.cfg5_chunk(3, i, per, n_groups=n_groups) for i in range(3)]
    aggs, groups = [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")]
    keep = [pp.ResidentBatch(r) for r in recs]
    plan = pp.HashAggregatePlan(None, aggs, groups)
    try:
        plan.CallbackResident(keep)
        assert plan.last_kernel() == "fdb_hash_kernel"
        got = _cfg5_canonical(plan.Finish())
    finally:
        plan.Close()
        for k in keep:
            k.close()
    o = OraclePlan(None, aggs, groups)
    for r in recs:
        o.push(r)
    ob = o.finish()
    want = _cfg5_canonical(ob.to_arrow())
    ob.close(); o.close()
    assert len(got["gid"]) == len(want["gid"]) > 1_000_000
    assert np.array_equal(got["gid"], want["gid"])
    for c in range(synth.CFG5_COLS):
        assert np.array_equal(got["labels"][c], want["labels"][c]), c
    assert np.array_equal(got["count(value)"], want["count(value)"])
    assert np.allclose(got["sum(value)"], want["sum(value)"], rtol=REL_TOL, atol=0.0)


def test_cfg5_full_size_every_group_checked(pp):
    """BASELINE.json config 5 at full size — 100 M rows, 32 label columns, 10 M distinct groups — with EVERY group checked: the
    generator's group ids give the expected count and sum per group through numpy's bincount (no 32-column group-by on the
    host); the result's ids are decoded from its first 12 label columns, its other 20 label columns (incl. NULLs) must be the
    function of the id the generator used. Also Σ count = rows and the number of groups = numpy's count of distinct ids."""
    from frostdb_amd import synth
    n_groups, rows, per = 10_000_000, 100_000_000, 12_500_000
    exp_cnt = np.zeros(n_groups, dtype=np.int64)
    exp_sum = np.zeros(n_groups, dtype=np.float64)
    keep = []
    try:
        for i in range(rows // per):
            rec = synth.cfg5_chunk(0, i, per, n_groups=n_groups)
            gid = synth.cfg5_group_ids(0, i, per, n_groups=n_groups)
            val = rec.column(rec.schema.get_field_index("value")).to_numpy()
            exp_cnt += np.bincount(gid, minlength=n_groups)
            exp_sum += np.bincount(gid, weights=val, minlength=n_groups)
            keep.append(pp.ResidentBatch(rec))
            del rec, gid, val
        plan = pp.HashAggregatePlan(None, [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")])
        try:
            plan.CallbackResident(keep)
            out = plan.Finish()
        finally:
            plan.Close()
    finally:
        for k in keep:
            k.close()
    got = _cfg5_canonical(out)
    present = np.flatnonzero(exp_cnt)
    assert len(got["gid"]) == len(present) and np.array_equal(got["gid"], present)
    assert int(got["count(value)"].sum()) == rows
    assert np.array_equal(got["count(value)"], exp_cnt[present])
    assert np.allclose(got["sum(value)"], exp_sum[present], rtol=REL_TOL, atol=0.0)
    digits, nulls = synth._cfg5_tables(n_groups)
    for c in range(synth.CFG5_COLS):
        want = np.where(nulls[c][present], 255, digits[c][present]).astype(np.uint8)
        assert np.array_equal(got["labels"][c], want), c


def test_hash_table_growth_when_the_cardinality_estimate_is_too_low(pp):
    """A skewed key distribution (Zipf over 2 M int64 keys, 3 M rows in three records) makes the uniform-draw estimate an
    UNDER-estimate: the table must keep growing on its own and the result must still be exact."""
    rng = np.random.default_rng(99)
    recs = []
    for _ in range(3):
        k = np.minimum(rng.zipf(1.3, 1_000_000), 2_000_000).astype(np.int64)
        recs.append(pa.RecordBatch.from_arrays([pa.array(k), pa.array(rng.integers(0, 100, 1_000_000).astype(np.int64))], names=["k", "v"]))
    aggs, groups = [Sum(Col("v")), Count(Col("v")), Max(Col("v"))], [Col("k")]
    keep = [pp.ResidentBatch(r) for r in recs]
    plan = pp.HashAggregatePlan(None, aggs, groups)
    try:
        plan.CallbackResident(keep)
        out = plan.Finish()
    finally:
        plan.Close()
        for k_ in keep:
            k_.close()
    allk = np.concatenate([r.column(0).to_numpy() for r in recs])
    allv = np.concatenate([r.column(1).to_numpy() for r in recs])
    uk, inv = np.unique(allk, return_inverse=True)
    order = np.argsort(out.column("k").to_numpy())
    assert np.array_equal(out.column("k").to_numpy()[order], uk)
    assert np.array_equal(out.column("sum(v)").to_numpy()[order], np.bincount(inv, weights=allv).astype(np.int64))
    assert np.array_equal(out.column("count(v)").to_numpy()[order], np.bincount(inv))
    mx = np.full(len(uk), -1, dtype=np.int64)
    np.maximum.at(mx, inv, allv)
    assert np.array_equal(out.column("max(v)").to_numpy()[order], mx)


def test_records_carrying_prehashed_columns_give_the_same_result(pp, variant):
    """with `Prehash: true` a table's records carry `hashed.<col>` int64 columns (dynparquet/hashed.go:27-84)
    which the reference uses INSTEAD of hashing the key column (aggregate.go:386-392). Group identity here is the key tuple
    itself, so such records must simply work: no matcher — not even DynCol("labels") — picks the helper columns up as keys,
    and the result equals both the oracle's (which takes the stored hashes, like the reference) and the result without them."""
    import oracle
    rng = np.random.default_rng(17)
    base = [make_prometheus_batch(rng, 30_000, n_path=37), make_prometheus_batch(rng, 20_000, n_path=53)]

    def with_hashes(rec):
        out = rec
        for name in ("labels.path", "labels.code"):
            col = rec.column(rec.schema.get_field_index(name)).dictionary_decode().to_pylist()
            h = np.array([oracle.metro_hash64(v) if v is not None else 0 for v in col], dtype=np.uint64).view(np.int64)
            out = out.append_column("hashed." + name, pa.array(h))
        return out

    hashed = [with_hashes(r) for r in base]
    for q in (CFG2, CFG3, dict(filter_expr=None, aggs=[Count(Col("value")), Sum(Col("value"))], groups=[DynCol("labels")])):
        cols = sorted(c for c in base[0].schema.names if c.startswith("labels.")) if q["groups"][0].dynamic else ["labels.path"]
        cols = cols + [a.Name() for a in q["aggs"]]
        got_h = run_gpu(pp, hashed, q["filter_expr"], q["aggs"], q["groups"], resident=True)
        got = run_gpu(pp, base, q["filter_expr"], q["aggs"], q["groups"], resident=True)
        assert not any(k.startswith("hashed.") for k in got_h)
        assert_same_result(got_h, got, cols, float_cols={"sum(value)"})
        assert_same_result(got_h, run_oracle(hashed, **q), cols, float_cols={"sum(value)"})


# ---- filter() on resident records: one-pass compaction (fdb_plan_filter_batch / fdb_plan_select_batch) -----------------------

def _oracle_filter(rec, filt):
    from oracle import OraclePlan
    o = OraclePlan(filt)
    out, idx = o.filter(rec)
    d = out.to_pydict() if out is not None else None
    if out is not None:
        out.close()
    o.close()
    return d, idx


@pytest.mark.parametrize("n", [0, 1, 5, 1023, 1024, 1025, 4097, 65_537, 1_000_003])
def test_resident_filter_compaction_matches_the_oracle(pp, n, variant):
    """The compacted record of fdb_plan_filter_batch (every column type the path stages: dictionary, nullable dictionary, int64,
    float64 with NULLs, bool, plain string) against the oracle's filter() — values, NULLs and row ORDER — across tile edges."""
    rng = np.random.default_rng(n + 3)
    rec = make_prometheus_batch(rng, n, n_path=300, null_frac=0.03) if n else make_prometheus_batch(rng, 1, n_path=3).slice(0, 0)
    fv = pa.array(rng.uniform(0, 1, n), mask=(rng.random(n) < 0.1) if n else None)
    rec = rec.append_column("fnull", fv).append_column("flag", pa.array(rng.random(n) < 0.5)).append_column(
        "plain", pa.array([b"s%d" % (i % 7) if i % 11 else None for i in range(n)], type=pa.binary()))
    filt = And(Or(Col("labels.code") == "200", Col("labels.code") == "404"), Col("value") > 250.0)
    plan = pp.HashAggregatePlan(filt)
    rb = pp.ResidentBatch(rec)
    try:
        out = plan.FilterResident(rb)
        got = out.to_arrow()
        want, idx = _oracle_filter(rec, filt)
        assert got.num_rows == out.num_rows == len(idx)
        assert got.schema.names == rec.schema.names
        if len(idx):
            g = arrow_to_pydict(got)
            for name in rec.schema.names:
                assert g[name] == want[name], name
            assert got.schema.field("labels.path").type == rec.schema.field("labels.path").type
            assert got.schema.field("plain").type == pa.binary() and got.schema.field("flag").type == pa.bool_()
        out.close()
    finally:
        plan.Close()
        rb.close()


def test_filter_of_many_resident_records_in_one_launch_sequence_matches_the_oracle(pp, variant):
    """fdb_plan_filter_batches: PredicateFilter.Callback for every record of a scan at once (flags kernel generated for the
    predicate over all records, one prefix-sum launch, one compaction launch for every column of every record). Records of very
    different sizes — empty, one row, exact tile multiples, tile edges ± 1, many tiles — with different dictionaries (a LUT class
    per record), NULLs in every nullable column, a record in which nothing qualifies and one in which everything does: every
    output must equal the oracle's filter() of its record — values, NULLs and row ORDER — and what the per-record entry point gives."""
    rng = np.random.default_rng(41)
    sizes = [70_001, 0, 1, 2048, 2047, 2049, 4096, 300_000, 5, 1_048_576 + 17]
    recs = []
    for k, n in enumerate(sizes):
        rec = make_prometheus_batch(rng, n, n_path=40 + 13 * k, null_frac=0.02) if n else make_prometheus_batch(rng, 1, n_path=3).slice(0, 0)
        fv = pa.array(rng.uniform(0, 1, n), mask=(rng.random(n) < 0.1) if n else None)
        recs.append(rec.append_column("fnull", fv).append_column("flag", pa.array(rng.random(n) < 0.5)))
    lo = recs[3].column("value").to_numpy()
    recs[3] = recs[3].set_column(recs[3].schema.get_field_index("value"), "value", pa.array(np.zeros(len(lo))))          # nothing qualifies
    recs[4] = recs[4].set_column(recs[4].schema.get_field_index("value"), "value", pa.array(np.full(2047, 999.0)))       # (value part) everything does
    filt = And(Or(Col("labels.code") == "200", Col("labels.code") == "404", Col("labels.code") == None), Col("value") > 250.0)  # noqa: E711
    plan = pp.HashAggregatePlan(filt)
    rbs = [pp.ResidentBatch(r) for r in recs]
    try:
        outs = plan.FilterResidentMany(rbs)
        assert len(outs) == len(recs)
        kernel = plan.last_kernel()
        assert ("fdb_flags_kernel" in kernel or "fdb_select_kernel" in kernel) == (variant == "specialised"), kernel
        for rec, rb, out in zip(recs, rbs, outs):
            want, idx = _oracle_filter(rec, filt)
            got = out.to_arrow()
            assert got.num_rows == out.num_rows == len(idx), (rec.num_rows, got.num_rows, len(idx))
            assert got.schema.names == rec.schema.names
            if len(idx):
                g = arrow_to_pydict(got)
                for name in rec.schema.names:
                    assert g[name] == want[name], (rec.num_rows, name)
            single = plan.FilterResident(rb)
            assert single.to_arrow().equals(got)
            single.close()
        for o in outs:
            o.close()
        # records whose column sets differ (schema drift) still come back right: the launch falls back to one pass per record
        odd = [recs[0], recs[7].drop_columns(["flag"])]
        rb2 = [pp.ResidentBatch(r) for r in odd]
        outs = plan.FilterResidentMany(rb2)
        for rec, out in zip(odd, outs):
            want, idx = _oracle_filter(rec, filt)
            g = arrow_to_pydict(out.to_arrow())
            assert out.num_rows == len(idx) and all(g[nm] == want[nm] for nm in rec.schema.names)
            out.close()
        for r in rb2:
            r.close()
    finally:
        plan.Close()
        for r in rbs:
            r.close()


def test_filter_batches_whose_records_need_different_predicate_shapes(pp):
    """Parts of one table, same column set, but the filtered dictionary has 10 entries in one part and 200 in the next (the truth
    table rides in a 64-bit immediate for ≤ 64 entries and in a byte LUT above) — and a part whose dictionary lacks the literal:
    the shapes do not merge, the launch falls back to one pass per record, and that fall-back stages its own LUTs (it used to run
    while the multi-record path's staging deferral was still in force, so the device read LUT bytes that had never been shipped)."""
    rng = np.random.default_rng(2024)
    recs = [make_prometheus_batch(rng, 50_000, n_path=10, null_frac=0.02), make_prometheus_batch(rng, 70_001, n_path=200, null_frac=0.02),
            make_prometheus_batch(rng, 30_000, n_path=3, null_frac=0.0), make_prometheus_batch(rng, 2_048, n_path=65, null_frac=0.1)]
    filt = And(Col("labels.path").RegexMatch("p000[1-7]$|p01[0-9]{2}$"), Col("value") > 100.0)
    for trial in range(3):  # (the staging ring restarts between calls: repeat so that stale bytes of an earlier call cannot hide it)
        plan = pp.HashAggregatePlan(filt)
        rbs = [pp.ResidentBatch(r) for r in (recs if trial != 1 else recs[::-1])]
        try:
            outs = plan.FilterResidentMany(rbs)
            for rec, out in zip(recs if trial != 1 else recs[::-1], outs):
                want, idx = _oracle_filter(rec, filt)
                g = arrow_to_pydict(out.to_arrow())
                assert out.num_rows == len(idx), (trial, rec.num_rows, out.num_rows, len(idx))
                assert all(g[nm] == want[nm] for nm in rec.schema.names)
                out.close()
        finally:
            plan.Close()
            for r in rbs:
                r.close()


def test_filter_batches_when_only_some_records_have_nulls_in_a_column(pp):
    """The compaction kernel is specialised per COLUMN (width × nullable) for the whole launch: a column that has a validity bitmap
    in one record and none in another (a part without NULLs drops the bitmap) is compacted as nullable everywhere — the records
    without a bitmap read an all-ones one — and each output keeps a bitmap only if NULLs were actually selected."""
    rng = np.random.default_rng(99)
    recs = []
    for k, (n, nf) in enumerate([(40_000, 0.05), (30_001, 0.0), (2_048, 0.0), (55_555, 0.2), (10, 0.0)]):
        recs.append(make_prometheus_batch(rng, n, n_path=25, null_frac=nf))
    assert recs[1].column("labels.path").null_count == 0 and recs[0].column("labels.path").null_count > 0
    filt = Or(Col("value") > 600.0, Col("labels.code") == "404")
    plan = pp.HashAggregatePlan(filt)
    rbs = [pp.ResidentBatch(r) for r in recs]
    try:
        outs = plan.FilterResidentMany(rbs)
        assert "fdb_flags_kernel" in plan.last_kernel()
        for rec, out in zip(recs, outs):
            want, idx = _oracle_filter(rec, filt)
            got = out.to_arrow()
            assert got.num_rows == len(idx)
            g = arrow_to_pydict(got)
            for name in rec.schema.names:
                assert g[name] == want[name], (rec.num_rows, name)
                assert got.column(name).null_count == sum(v is None for v in want[name])
            out.close()
    finally:
        plan.Close()
        for r in rbs:
            r.close()


def test_filter_batches_without_any_bitmap(pp):
    """No column of any record carries a validity bitmap: the compaction runs its lean instantiation, nothing is zeroed and no NULL
    counts come back — rows, order and values must still be the oracle's, and the outputs carry no bitmaps."""
    rng = np.random.default_rng(123)
    recs = [make_prometheus_batch(rng, n, n_path=9, null_frac=0.0, with_method=False) for n in (70_001, 2_048, 1, 33_333)]
    assert all(c.null_count == 0 for r in recs for c in r.columns)
    filt = Or(Col("value") > 700.0, Col("labels.code") == "500")
    plan = pp.HashAggregatePlan(filt)
    rbs = [pp.ResidentBatch(r) for r in recs]
    try:
        outs = plan.FilterResidentMany(rbs)
        for rec, out in zip(recs, outs):
            want, idx = _oracle_filter(rec, filt)
            got = out.to_arrow()
            assert got.num_rows == len(idx)
            g = arrow_to_pydict(got)
            for name in rec.schema.names:
                if want is not None:  # (None: the oracle selected no row of this record)
                    assert g[name] == want[name], (rec.num_rows, name)
                assert got.column(name).null_count == 0
            out.close()
    finally:
        plan.Close()
        for r in rbs:
            r.close()


def test_the_same_query_over_changing_records_stages_fresh_tables(pp):
    """A query repeated over resident data ships its launch tables (predicate LUTs, argument blocks) only when their bytes changed
    (Context::flush_staging compares with what the device ring already holds). Alternate one plan description over two record
    sets whose dictionaries differ — every run must see its own tables — and repeat one of them back to back (the skipped-copy
    case); each result is the oracle's."""
    rng = np.random.default_rng(321)
    a = [make_prometheus_batch(rng, 30_000, n_path=40), make_prometheus_batch(rng, 12_345, n_path=40)]
    b = [make_prometheus_batch(rng, 30_000, n_path=17), make_prometheus_batch(rng, 12_345, n_path=17)]
    filt = And(Col("labels.code") == "200", Col("value") > 250.0)
    aggs, groups = [Sum(Col("value")), Count(Col("value"))], [Col("labels.path")]
    cols = ["labels.path", "sum(value)", "count(value)"]
    want = {"a": run_oracle(a, filt, aggs, groups), "b": run_oracle(b, filt, aggs, groups)}
    ra, rb = [pp.ResidentBatch(r) for r in a], [pp.ResidentBatch(r) for r in b]
    try:
        for which in ["a", "a", "b", "a", "b", "b", "a"]:
            plan = pp.HashAggregatePlan(filt, aggs, groups)
            try:
                plan.CallbackResident(ra if which == "a" else rb)
                got = arrow_to_pydict(plan.Finish())
            finally:
                plan.Close()
            assert_same_result(got, want[which], cols, float_cols=("sum(value)",))
    finally:
        for r in ra + rb:
            r.close()


def test_filter_batches_properties_at_scale(pp):
    """Size-independent properties of fdb_plan_filter_batches at 48 M rows (4 records, sizes off the tile grid): a predicate and its
    complement partition the non-NULL rows (counts add up, Σ value adds up to the column's own sum); filtering the result again
    selects everything and leaves it unchanged (idempotence); rows keep their order (timestamps stay non-decreasing); NULL counts of
    the outputs add up to the NULLs among the selected rows."""
    from frostdb_amd import synth
    sizes = [12_000_000, 11_999_999, 12_000_001, 12_345_678]
    recs = [synth.prometheus_chunk(3, i, n, row_base=sum(sizes[:i])) for i, n in enumerate(sizes)]
    rbs = [pp.ResidentBatch(r) for r in recs]
    a = pp.HashAggregatePlan(Col("value") > 321.5)
    b = pp.HashAggregatePlan(Col("value") <= 321.5)
    try:
        oa, ob = a.FilterResidentMany(rbs), b.FilterResidentMany(rbs)
        for r, x, y in zip(recs, oa, ob):
            v = r.column(r.schema.get_field_index("value")).to_numpy()
            assert x.num_rows == int((v > 321.5).sum()) and x.num_rows + y.num_rows == r.num_rows
        first = oa[1].to_arrow()
        v1 = recs[1].column(recs[1].schema.get_field_index("value")).to_numpy()
        m = v1 > 321.5
        assert np.array_equal(first.column("value").to_numpy(), v1[m])
        ts = first.column("timestamp").to_numpy()
        assert np.all(np.diff(ts) >= 0)
        code = recs[1].column(0)
        assert first.column("labels.code").null_count == int(np.asarray(code.is_null())[m].sum())
        total = sum(float(o.to_arrow().column("value").to_numpy().sum()) for o in (oa[3], ob[3]))
        assert math.isclose(total, float(recs[3].column(recs[3].schema.get_field_index("value")).to_numpy().sum()), rel_tol=1e-12)
        again = a.FilterResidentMany(oa)
        for x, y in zip(oa, again):
            assert y.num_rows == x.num_rows
        assert again[2].to_arrow().equals(oa[2].to_arrow())
        for o in oa + ob + again:
            o.close()
    finally:
        a.Close(); b.Close()
        for r in rbs:
            r.close()


def test_resident_selection_vector_and_capacity_retry(pp):
    """fdb_plan_select_batch writes the ascending selection vector into a device buffer; a 12 M-row record with a predicate that
    only matches in its second half defeats the strided sample less than it defeats a prefix sample — either way the result must
    be exact (the pass is repeated with the exact size when the estimate was short)."""
    import torch
    n = 12_000_000
    rng = np.random.default_rng(2)
    ts = np.arange(n, dtype=np.int64)
    code = pa.DictionaryArray.from_arrays(pa.array((rng.random(n) < 0.5).astype(np.uint32)), pa.array([b"200", b"500"], type=pa.binary()))
    rec = pa.RecordBatch.from_arrays([code, pa.array(ts), pa.array(rng.uniform(0, 1, n))], names=["labels.code", "timestamp", "value"])
    filt = And(Col("labels.code") == "200", Col("timestamp") >= n // 2 + 12345)
    plan = pp.HashAggregatePlan(filt)
    rb = pp.ResidentBatch(rec)
    try:
        want = np.flatnonzero((code.indices.to_numpy() == 0) & (ts >= n // 2 + 12345)).astype(np.uint32)
        buf = torch.empty(n, dtype=torch.int32, device="cuda:0")
        k = plan.SelectResident(rb, buf.data_ptr(), n)
        assert k == len(want)
        assert np.array_equal(buf[:k].cpu().numpy().view(np.uint32), want)
        out = plan.FilterResident(rb)
        got = out.to_arrow()
        assert got.num_rows == len(want)
        assert np.array_equal(got.column("timestamp").to_numpy(), ts[want])
        assert np.array_equal(got.column("value").to_numpy(), rec.column("value").to_numpy()[want])
        out.close()
    finally:
        plan.Close()
        rb.close()


def test_float_sums_with_heavy_cancellation_state_the_contract(pp, variant):
    """float64 SUM over values of mixed sign and magnitude (±1e12 next to O(1)): the reference adds in row order within a chain
    (possibly in SIMD lanes) and in arrival order across chains — its own result is order-dependent, and so is
    any parallel reduction's (LDS / global atomics here, rank order in the cross-GPU merge). What CAN be promised, and what
    north_star's "1e-9 relative" means once terms cancel, is an error bound relative to the magnitudes that were added:
    |got − exact| ≤ 1e-9 · Σ|x| per group (every summation order is within n·ε·Σ|x| ≈ 1e-11 · Σ|x| at this size). exact = math.fsum."""
    rng = np.random.default_rng(4242)
    n = 200_000
    big = rng.choice([-1.0, 1.0], n) * 1e12 * rng.random(n)
    val = np.where(rng.random(n) < 0.5, big, rng.normal(0, 1, n))
    # make every group's big terms cancel almost exactly: pair each big value with its negation elsewhere in the group
    val[1::2] = -val[0::2]
    val += rng.normal(0, 1e-3, n)
    path = rng.integers(0, 50, n).astype(np.uint32)
    path[1::2] = path[0::2]
    rec = pa.RecordBatch.from_arrays(
        [pa.DictionaryArray.from_arrays(pa.array(path), pa.array([b"p%02d" % i for i in range(50)], type=pa.binary())), pa.array(val)],
        names=["labels.path", "value"])
    got = run_gpu(pp, [rec.slice(0, 120_001), rec.slice(120_001)], None, [Sum(Col("value")), Count(Col("value"))], [Col("labels.path")], resident=True)
    by_path = dict(zip(got["labels.path"], got["sum(value)"]))
    for g in range(50):
        x = val[path == g]
        exact, mag = math.fsum(x), float(np.abs(x).sum())
        assert abs(by_path[b"p%02d" % g] - exact) <= 1e-9 * mag, (g, by_path[b"p%02d" % g], exact, mag)
        assert abs(exact) < 1e-6 * mag  # (the test really is cancellation-heavy)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_oracle_sees_ten_million_rows_of_the_benchmark_workloads(pp, cfg):
    """BASELINE.json configs 2 and 3 at 10 M rows (bench.py's generator, 4 resident records in one launch) against the ORACLE —
    the restatement that carries the reference's quirks — not just against numpy."""
    from frostdb_amd import synth
    q = CFG2 if cfg == "cfg2" else CFG3
    recs = [synth.prometheus_chunk(1, i, 2_500_000, row_base=i * 2_500_000, cfg3=(cfg == "cfg3")) for i in range(4)]
    keep = [pp.ResidentBatch(r) for r in recs]
    plan = pp.HashAggregatePlan(q["filter_expr"], q["aggs"], q["groups"])
    try:
        plan.CallbackResident(keep)
        assert plan.last_kernel() == "fdb_plan_kernel"
        got = arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
        for k in keep:
            k.close()
    want = run_oracle(recs, **q, nchains=4)
    assert_same_result(got, want, ["labels.path"] + [a.Name() for a in q["aggs"]], float_cols={"sum(value)"})


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_oracle_sees_all_hundred_million_rows_of_the_benchmark_workloads(pp, cfg):
    """BASELINE.json configs 2 and 3 at their FULL size — 100 M rows, bench.py's generator and record size (4 × 25 M, one launch) —
    against the ORACLE run the way bench.py's cpu_baseline runs it (one chain per host core → Synchronizer → final stage):
    every group's count / MIN / MAX bit-exact, sums within 1e-9 relative. (Round 4 stopped at 10 M rows for the oracle and used numpy
    at this size.)"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from frostdb_amd import synth
    from oracle import OracleBatch, OraclePlan
    q = CFG2 if cfg == "cfg2" else CFG3
    per = 25_000_000
    with ThreadPoolExecutor(4) as ex:
        recs = list(ex.map(lambda i: synth.prometheus_chunk(0, i, per, row_base=i * per, cfg3=(cfg == "cfg3")), range(4)))
    keep = [pp.ResidentBatch(r) for r in recs]
    plan = pp.HashAggregatePlan(q["filter_expr"], q["aggs"], q["groups"])
    try:
        plan.CallbackResident(keep)
        assert plan.last_kernel() == "fdb_plan_kernel"
        got = arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
        for k in keep:
            k.close()
    threads = os.cpu_count() or 1
    batches = [OracleBatch.from_arrow(r.slice(o, min(1 << 20, per - o))) for r in recs for o in range(0, per, 1 << 20)]
    oplan = OraclePlan(q["filter_expr"], q["aggs"], q["groups"], nchains=threads)
    res = oplan.execute(batches, threads)
    want = res.to_pydict()
    res.close(); oplan.close()
    for b in batches:
        b.close()
    assert sum(want["count(value)"]) > 10_000_000 if cfg == "cfg3" else len(want["labels.path"]) == 1025
    assert_same_result(got, want, ["labels.path"] + [a.Name() for a in q["aggs"]], float_cols={"sum(value)"})


def test_convert_isnull_if_projections_on_the_device(pp):
    """convertProjection / isNullProjection / ifExprProjection (project.go:493-702) fused into the scan: as aggregate inputs
    (`sum(convert(ivalue, float64) * value)`, `sum(if(ivalue > 10) { ivalue } else { 1})`) and as group keys (`isnull(ivalue)`,
    an if-bucket), on columns with NULLs, dense and hash tables — against the oracle's restatement."""
    from frostdb_amd.logicalplan import BinaryExpr, Convert, If, IsNull, Literal, OP_GT, OP_MUL
    rng = np.random.default_rng(77)
    recs = []
    for n in (30_000, 17_001):
        base = make_prometheus_batch(rng, n, n_path=20)
        iv = pa.array(rng.integers(-50, 50, n), mask=rng.random(n) < 0.2)
        recs.append(base.append_column("ivalue", iv))
    conv = BinaryExpr(Convert(Col("ivalue")), OP_MUL, Col("value"))
    pick = If(BinaryExpr(Col("ivalue"), OP_GT, Literal(10)), Col("ivalue"), Literal(1))
    cases = [
        (CFG2["filter_expr"], [Sum(conv), Sum(pick), Count(Col("value"))], [Col("labels.path")]),
        (None, [Sum(pick), Max(pick)], [IsNull(Col("ivalue")), Col("labels.code")]),
        (None, [Sum(conv)], [If(BinaryExpr(Col("ivalue"), OP_GT, Literal(0)), Col("ivalue"), Literal(-1))]),
    ]
    for filt, aggs, groups in cases:
        cols = [g.name for g in groups] + [a.Name() for a in aggs]
        for resident in (False, True):
            got = run_gpu(pp, recs, filt, aggs, groups, resident=resident)
            want = run_oracle(recs, filt, aggs, groups)
            assert_same_result(got, want, cols, float_cols={a.Name() for a in aggs if "convert" in a.Name()})


# ---- NaN / ±Inf / −0.0 in float64 columns --------------------------------------------------------------

def _special_float_record(rng, n, n_path, special_first: bool):
    """`value` float64 with NaN, ±Inf and ±0.0 sprinkled in. With special_first = False the first row of every group is an ordinary
    number — the case in which the reference's first-value-wins loops (aggregate.go:846-857, :924-934) never keep a NaN."""
    path = rng.integers(0, n_path, size=n).astype(np.uint32)
    value = rng.uniform(-1000, 1000, size=n)
    kind = rng.random(n)
    value[kind < 0.05] = np.nan
    value[(kind >= 0.05) & (kind < 0.08)] = np.inf
    value[(kind >= 0.08) & (kind < 0.11)] = -np.inf
    value[(kind >= 0.11) & (kind < 0.13)] = 0.0
    value[(kind >= 0.13) & (kind < 0.15)] = -0.0
    if not special_first:
        first = np.unique(path, return_index=True)[1]
        value[first] = rng.uniform(-1000, 1000, size=len(first))
    names = pa.array([b"/p%03d" % i for i in range(n_path)], type=pa.binary())
    return pa.RecordBatch.from_arrays([pa.DictionaryArray.from_arrays(pa.array(path), names), pa.array(value)], names=["labels.path", "value"])


def _same_float(a, b):
    return (math.isnan(a) and math.isnan(b)) or a == b  # (−0.0 == +0.0: numerically equal, which is the bar for float results)


@pytest.mark.parametrize("resident", [False, True])
def test_float_min_max_with_inf_and_nan_not_first_equal_the_reference(pp, variant, resident):
    """±Inf are ordinary ordered values; a NaN that is not its group's first value is never kept by the reference — and never
    by the device: MIN / MAX equal the oracle's bit for bit (up to the sign of a zero)."""
    rng = np.random.default_rng(77)
    recs = [_special_float_record(rng, 40_000, 37, special_first=False)]
    aggs = [Min(Col("value")), Max(Col("value")), Count(Col("value"))]
    got = run_gpu(pp, recs, None, aggs, [Col("labels.path")], resident=resident)
    want = run_oracle(recs, None, aggs, [Col("labels.path")])
    g = {k: (a, b, c) for k, a, b, c in zip(got["labels.path"], got["min(value)"], got["max(value)"], got["count(value)"])}
    w = {k: (a, b, c) for k, a, b, c in zip(want["labels.path"], want["min(value)"], want["max(value)"], want["count(value)"])}
    assert g.keys() == w.keys() and len(g) == 37
    for k in g:
        assert _same_float(g[k][0], w[k][0]) and _same_float(g[k][1], w[k][1]) and g[k][2] == w[k][2], (k, g[k], w[k])
    assert any(v[0] == -math.inf for v in g.values()) and any(v[1] == math.inf for v in g.values())
    # NaN-free data with infinities: SUM follows IEEE too (+Inf + −Inf = NaN, like the reference's plain adds)
    v = np.array(recs[0].column("value"))
    clean = recs[0].set_column(1, "value", pa.array(np.where(np.isnan(v), 1.5, v)))
    gs = run_gpu(pp, [clean], None, [Sum(Col("value"))], [Col("labels.path")], resident=resident)
    ws = run_oracle([clean], None, [Sum(Col("value"))], [Col("labels.path")])
    gm, wm = dict(zip(gs["labels.path"], gs["sum(value)"])), dict(zip(ws["labels.path"], ws["sum(value)"]))
    for k in wm:
        assert _same_float(gm[k], wm[k]) or math.isclose(gm[k], wm[k], rel_tol=REL_TOL), (k, gm[k], wm[k])
    assert any(math.isnan(x) or math.isinf(x) for x in wm.values())


def test_float_min_max_device_rule_for_nan_and_negative_zero(pp, variant):
    """The documented device rule, stated explicitly because the reference's own answer depends on row order
    (a NaN survives its `if v < minV` loop only as the group's FIRST value, aggregate.go:846-857): the MIN / MAX of a group are those
    of its non-NaN values; a group with nothing but NaNs gives NaN; −0.0 and +0.0 are one value, reported as +0.0. Where the
    reference's order-dependent answer differs (NaN first) the test says so instead of hiding it."""
    rng = np.random.default_rng(78)
    rec = _special_float_record(rng, 30_000, 23, special_first=True)
    path = np.array(rec.column("labels.path").indices)
    v = np.array(rec.column("value"))
    v[path == 5] = np.nan                      # a group of nothing but NaNs
    v[path == 6] = np.where(rng.random((path == 6).sum()) < 0.5, 0.0, -0.0)  # a group of zeros of both signs
    first7 = np.flatnonzero(path == 7)[0]
    v[first7] = np.nan                         # NaN first: the reference answers NaN for this group, the device does not
    rec = rec.set_column(1, "value", pa.array(v))
    aggs = [Min(Col("value")), Max(Col("value"))]
    for resident in (False, True):
        got = run_gpu(pp, [rec], None, aggs, [Col("labels.path")], resident=resident)
        g = {int(k[2:]): (a, b) for k, a, b in zip(got["labels.path"], got["min(value)"], got["max(value)"])}
        for p in range(23):
            x = v[path == p]
            x = x[~np.isnan(x)]
            if len(x) == 0:
                assert math.isnan(g[p][0]) and math.isnan(g[p][1]), (p, g[p])
            else:
                assert g[p] == (x.min(), x.max()), (p, g[p], x.min(), x.max())
        assert math.copysign(1.0, g[6][0]) == 1.0 and math.copysign(1.0, g[6][1]) == 1.0 and g[6] == (0.0, 0.0)
    want = run_oracle([rec], None, aggs, [Col("labels.path")])
    w = {int(k[2:]): (a, b) for k, a, b in zip(want["labels.path"], want["min(value)"], want["max(value)"])}
    assert math.isnan(w[7][0]) and math.isnan(w[7][1]) and not math.isnan(g[7][0])  # the one documented difference
    assert math.isnan(w[5][0]) and math.isnan(g[5][0])
    # two-stage: partial results holding NaN (the all-NaN group) merge by the same rule
    a, b = rec.slice(0, 15_000), rec.slice(15_000)
    p1, p2 = pp.HashAggregatePlan(None, aggs, [Col("labels.path")]), pp.HashAggregatePlan(None, aggs, [Col("labels.path")])
    try:
        p1.Callback(a); p2.Callback(b)
        p1.Merge(p2)
        m = arrow_to_pydict(p1.Finish())
        gm = {int(k[2:]): (x, y) for k, x, y in zip(m["labels.path"], m["min(value)"], m["max(value)"])}
        for p in range(23):
            assert all(_same_float(x, y) for x, y in zip(gm[p], g[p])), (p, gm[p], g[p])
    finally:
        p1.Close(); p2.Close()


def test_float_compare_predicates_with_nan_rows_follow_ieee(pp, variant):
    """`value > x`, `>=`, `<`, `<=`, `==` never select a NaN row, `!=` always does (IEEE compares — what Arrow's compare kernels
    behind binaryscalarexpr.go:119-152 compute); ±Inf compare like numbers. Selection against the oracle, row for row."""
    rng = np.random.default_rng(79)
    rec = _special_float_record(rng, 50_000, 11, special_first=True)
    v = np.array(rec.column("value"))
    n_nan = int(np.isnan(v).sum())
    assert n_nan > 1000
    for expr, npsel in [(Col("value") > 10.0, v > 10.0), (Col("value") >= 0.0, v >= 0.0), (Col("value") < 0.0, v < 0.0), (Col("value") <= -0.0, v <= 0.0),
                        (Col("value") == 0.0, v == 0.0), (Col("value") != 0.0, v != 0.0), (Col("value") > -math.inf, v > -np.inf),
                        (Col("value") == math.inf, v == np.inf), (Col("value") < math.inf, v < np.inf)]:
        plan = pp.HashAggregatePlan(expr)
        try:
            idx = plan.Select(rec)
            want_tbl, want_idx = _oracle_filter(rec, expr)
            assert list(idx) == list(want_idx) == list(np.flatnonzero(npsel)), expr
        finally:
            plan.Close()
        got = run_gpu(pp, [rec], expr, [Count(Col("value"))], [])
        assert got["count(value)"] == [int(npsel.sum())]


@pytest.mark.timeout(600)
def test_eight_chains_pushing_small_host_records_concurrently(pp):
    """The reference's shape of a scan: N chains, each on its own thread, each handed host records of the batch floor (1 024 rows,
    table.go:780) — here 8 chains × 2 500 records, so every chain scans its queue several times (one launch per 1 024 queued
    records, each staging its predicate tables and then 1 024 argument blocks). Every chain's own result and the merged one must
    equal the numpy expectation. (Until round 4 the staging ring restarted — or was freed and re-allocated — in the middle of such a
    launch's staging: with several chains in flight another thread's allocation landed on the freed block and the scan read
    garbage tables.)"""
    import threading

    import bench
    from frostdb_amd import synth
    from frostdb_amd.logicalplan import to_desc
    chains, rec_rows, per_chain = 8, 1024, 2_560_000
    filt, aggs, groups, _ = bench.query(2)
    desc = to_desc(filt, aggs, groups)
    src = synth.prometheus_chunk(0, 0, chains * per_chain)
    names = synth.PATHS + [None]

    def expect(b):
        s, c = bench.expected_cfg2(b)
        return {names[i]: s[i] for i in range(len(names)) if c[i]}
    exported = [[pp.ExportedBatch(src.slice(c * per_chain + o, min(rec_rows, per_chain - o))) for o in range(0, per_chain, rec_rows)] for c in range(chains)]
    plans = [pp.HashAggregatePlan(filt, aggs, groups, desc=desc) for _ in range(chains)]
    errors = []

    def work(c):
        try:
            plans[c].CallbackExportedMany(exported[c])
            plans[c].last_kernel()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    try:
        ts = [threading.Thread(target=work, args=(c,)) for c in range(chains)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        for c in range(chains):
            keys = plans[c].partial_keys().column(0).to_pylist()
            vals = np.zeros(len(keys), dtype=np.float64)
            plans[c].partial_state_into(0, vals.ctypes.data, vals.nbytes)
            want, got = expect(src.slice(c * per_chain, per_chain)), dict(zip(keys, vals.tolist()))
            assert set(got) == set(want), c
            assert all(math.isclose(got[k], want[k], rel_tol=1e-9) for k in want), c
        for p in plans[1:]:
            plans[0].Merge(p)
        res = plans[0].Finish()
        got, want = dict(zip(res.column(0).to_pylist(), res.column(1).to_pylist())), expect(src)
        assert set(got) == set(want) and all(math.isclose(got[k], want[k], rel_tol=1e-9) for k in want)
    finally:
        for p in plans:
            p.Close()
        for ex_list in exported:
            for ex in ex_list:
                ex.close()


def test_and_is_lazy_like_the_reference(pp, variant):
    """AndExpr.Eval (filter.go:172-190; filter_test.go:66-82 TestAndExprShortCircuits): when the left side of an AND selects no row of a
    record, the right side is not evaluated — a right side that could not be evaluated (here `<` on a dictionary column,
    binaryscalarexpr.go:106-108) is then no error for that record. Where the left side does select rows, the error is the reference's."""
    rng = np.random.default_rng(5)
    rec = make_prometheus_batch(rng, 30_000, n_path=20)
    bad = BinaryExpr(Col("labels.path"), 3, Literal("x"))  # OP_LT on a dictionary column: ErrUnsupportedBinaryOperation
    from oracle import OraclePlan
    for left, selects in ((Col("labels.code") == "no such code", False), (Col("value") < -1.0, False), (Col("labels.code") == "200", True)):
        filt = And(left, bad)
        o = OraclePlan(filt)
        if selects:
            with pytest.raises(Exception):
                o.filter(rec)
        else:
            out, idx = o.filter(rec)
            assert out is None and list(idx) == []
        o.close()
        for resident in (False, True):
            plan = pp.HashAggregatePlan(filt, [Count(Col("value"))], [Col("labels.path")])
            rb = pp.ResidentBatch(rec) if resident else None
            try:
                if selects:
                    with pytest.raises(pp.FdbError) as e:
                        plan.Callback(rb if resident else rec)
                        plan.Finish()
                    assert e.value.code == pp.FDB_ERR_UNSUPPORTED
                else:
                    plan.Callback(rb if resident else rec)
                    assert plan.Finish().num_rows == 0
            finally:
                plan.Close()
                if rb is not None:
                    rb.close()
        fp = pp.HashAggregatePlan(filt)
        try:
            if selects:
                with pytest.raises(pp.FdbError):
                    fp.Select(rec)
            else:
                assert list(fp.Select(rec)) == []
        finally:
            fp.Close()
    # a record where the left side selects nothing next to one where it selects something: only the second raises
    quiet = rec.set_column(rec.schema.get_field_index("value"), "value", pa.array(np.full(rec.num_rows, 5.0)))
    filt = And(Col("value") > 100.0, bad)
    plan = pp.HashAggregatePlan(filt, [Count(Col("value"))], [Col("labels.path")])
    try:
        plan.Callback(quiet)
        assert plan.num_groups() == 0
        with pytest.raises(pp.FdbError):
            plan.Callback(rec)
            plan.Finish()
    finally:
        plan.Close()


@pytest.mark.parametrize("case", ["value", "method", "method_and_code", "value_and_method", "timestamp"])
@pytest.mark.parametrize("thresh", [-1.0, 800.0, 2000.0])
def test_filter_in_one_pass_over_the_filter_columns(pp, case, thresh, monkeypatch):
    """fdb_select_kernel: the wave that evaluates a tile's predicate also places it (decoupled look-back inside the record, tiles handed
    out by tickets) and writes the compacted values of the filter columns it holds — `value` alone (8 bytes: the whole LDS budget), one
    or two dictionary columns without NULLs (4 + 4 bytes), `value` with a dictionary column that no longer fits, an int64 column — over
    records from one row to many tiles, with thresholds that select everything (worst-case blocks kept), ≈ 20 % (repacked into the
    exact arena) and nothing. Every output equals the oracle's filter() of its record and, bit for bit, what the
    three-launch path (bitmap → prefix sums → compaction, FDB_SELECT_TWO_PASS) returns."""
    rng = np.random.default_rng(7)
    sizes = [1, 2047, 2048, 2049, 8192, 70_001, 0, 90_000, 300_000]
    recs = []
    for k, n in enumerate(sizes):
        rec = make_prometheus_batch(rng, n, n_path=20 + k, null_frac=0.0 if k % 2 == 0 else 0.03) if n else make_prometheus_batch(rng, 1, n_path=3).slice(0, 0)
        if case == "method_and_code":  # `labels.code` without NULLs in EVERY record, or it is not fused
            ci = rec.schema.get_field_index("labels.code")
            c = rec.column(ci)
            rec = rec.set_column(ci, "labels.code", pa.DictionaryArray.from_arrays(pa.array(np.asarray(c.indices.fill_null(0)), type=pa.uint32()), c.dictionary))
        recs.append(rec)
    v = Col("value") > thresh
    m = Or(Col("labels.method") == "GET", Col("labels.method") == "PUT", Col("labels.method") == "DELETE") if thresh < 2000 else (Col("labels.method") == "PATCH")
    filt = {"value": v, "method": m, "method_and_code": And(m, Or(Col("labels.code") == "200", Col("labels.code") == "500")),
            "value_and_method": And(v, m), "timestamp": Col("timestamp") > int(1_700_000_000_000 + 15_000 * (thresh / 1000.0) * 40_000)}[case]
    results = {}
    for passes in ("one", "two"):
        if passes == "two":
            monkeypatch.setenv("FDB_SELECT_TWO_PASS", "1")
        else:  # (by default only predicates whose filter columns are all fused and 8 bytes wide take the one-pass kernel)
            monkeypatch.delenv("FDB_SELECT_TWO_PASS", raising=False)
            monkeypatch.setenv("FDB_SELECT_ONE_PASS", "1")
        plan = pp.HashAggregatePlan(filt)
        rbs = [pp.ResidentBatch(r) for r in recs]
        try:
            outs = plan.FilterResidentMany(rbs)
            assert ("fdb_select_kernel" in plan.last_kernel()) == (passes == "one"), plan.last_kernel()
            results[passes] = [o.to_arrow() for o in outs]
            assert [o.num_rows for o in outs] == [t.num_rows for t in results[passes]]
            for o in outs:
                o.close()
        finally:
            plan.Close()
            for r in rbs:
                r.close()
    for rec, one, two in zip(recs, results["one"], results["two"]):
        want, idx = _oracle_filter(rec, filt)
        assert one.num_rows == two.num_rows == len(idx), (case, thresh, rec.num_rows, one.num_rows, two.num_rows, len(idx))
        assert one.equals(two)
        if len(idx):
            g = arrow_to_pydict(one)
            for name in rec.schema.names:
                assert g[name] == want[name], (case, thresh, rec.num_rows, name)


def test_filter_in_one_pass_many_scans_at_once(pp):
    """Eight threads, one plan each, filter() over the same resident records concurrently: every scan's tiles are handed out by its own
    ticket counter and its look-back only ever waits for waves that already run, whatever else occupies the GPU; the contexts' control
    blocks are re-used from call to call (a status word of an earlier launch carries an earlier epoch)."""
    import threading
    rng = np.random.default_rng(17)
    recs = [make_prometheus_batch(rng, n, n_path=30, null_frac=0.02) for n in (150_000, 3, 400_001, 2048)]
    rbs = [pp.ResidentBatch(r) for r in recs]
    filt = Col("value") > 500.0  # (one 8-byte filter column: the one-pass kernel by default)
    want = [_oracle_filter(r, filt) for r in recs]
    errors = []

    def chain(t):
        try:
            for rep in range(6):
                plan = pp.HashAggregatePlan(filt)
                try:
                    outs = plan.FilterResidentMany(rbs)
                    assert "fdb_select_kernel" in plan.last_kernel()
                    for (w, idx), o in zip(want, outs):
                        assert o.num_rows == len(idx)
                        if rep == 5:
                            g = arrow_to_pydict(o.to_arrow())
                            assert g["value"] == w["value"] and g["labels.path"] == w["labels.path"]
                        o.close()
                finally:
                    plan.Close()
        except BaseException as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=chain, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r in rbs:
        r.close()
    assert not errors, errors


def test_deterministic_float_sums_are_bit_identical_from_run_to_run(pp):
    """fdb_plan_set_deterministic: SUM(float64) with a table per wave and fixed-order folds — the same records pushed in the same calls
    give the same BITS on every run (20 runs; values spread over 12 orders of magnitude, so that a different order of additions shows),
    within the usual tolerance of the oracle; MIN / MAX / COUNT ride along. Both table shapes: ≤ 8 slots (lane-private registers first)
    and 1 000 groups in LDS, and no group-by at all. A scan that cannot have per-wave tables is refused (at the call that launches it), not
    answered approximately."""
    rng = np.random.default_rng(2026)
    n = 600_000
    recs = []
    for k in range(3):
        rec = make_prometheus_batch(rng, n, n_path=1000, null_frac=0.02)
        v = rng.uniform(-1.0, 1.0, n) * np.power(10.0, rng.integers(-6, 7, n))
        recs.append(rec.set_column(rec.schema.get_field_index("value"), "value", pa.array(v)))
    aggs = [Sum(Col("value")), Min(Col("value")), Max(Col("value")), Count(Col("value"))]
    filt = Col("labels.code") != "404"
    for groups in ([Col("labels.path")], [Col("labels.method")], []):
        keys = [g.name for g in groups]
        cols = keys + [a.Name() for a in aggs]
        want = run_oracle(recs, filt, aggs, groups)
        first = None
        for run in range(20):
            plan = pp.HashAggregatePlan(filt, aggs, groups)
            plan.set_deterministic(True)
            rbs = [pp.ResidentBatch(r) for r in recs]
            try:
                plan.CallbackResident(rbs)
                out = plan.Finish()
                assert plan.last_kernel() == "fdb_plan_kernel"
            finally:
                plan.Close()
                for r in rbs:
                    r.close()
            d = arrow_to_pydict(out)
            if first is None:
                assert_same_result(d, want, cols, float_cols=("sum(value)",))
            bits = dict(d)
            bits["sum(value)"] = np.asarray(d["sum(value)"], dtype=np.float64).view(np.uint64).tolist()
            rows = rows_of(bits, cols)
            if first is None:
                first = rows
            assert rows == first, (keys, run)
    # refused, not approximated: 32 group columns need the hash table
    from frostdb_amd import synth
    wide = synth.cfg5_chunk(0, 0, 50_000, n_groups=10_000)
    plan = pp.HashAggregatePlan(None, [Sum(Col("value"))], [DynCol("labels")])
    plan.set_deterministic(True)
    try:
        with pytest.raises(pp.UnsupportedError):  # (a small host record is queued: the scan — and its refusal — happens at the latest at Finish)
            plan.Callback(wide)
            plan.Finish()
    finally:
        plan.Close()


def test_filter_in_one_pass_survives_the_epoch_wrap(pp, monkeypatch):
    """The one-pass kernel's control block is never cleared between launches: status and place words carry the 24-bit epoch of the
    launch that wrote them. When the epochs run out the context clears the block and starts over — exercised here by jumping to the end
    of the range (FDB_TEST_SELECT_EPOCH_JUMP) so that every few calls cross the wrap; every result still equals the oracle's."""
    monkeypatch.setenv("FDB_TEST_SELECT_EPOCH_JUMP", "1")
    rng = np.random.default_rng(5)
    recs = [make_prometheus_batch(rng, n, n_path=12, null_frac=0.01) for n in (100_000, 9_000, 300_000)]
    rbs = [pp.ResidentBatch(r) for r in recs]
    filt = Col("value") > 420.0
    want = [_oracle_filter(r, filt) for r in recs]
    try:
        for rep in range(9):
            plan = pp.HashAggregatePlan(filt)
            try:
                outs = plan.FilterResidentMany(rbs)
                assert "fdb_select_kernel" in plan.last_kernel()
                for (w, idx), o in zip(want, outs):
                    assert o.num_rows == len(idx), rep
                    g = arrow_to_pydict(o.to_arrow())
                    assert g["value"] == w["value"] and g["labels.code"] == w["labels.code"], rep
                    o.close()
            finally:
                plan.Close()
    finally:
        for r in rbs:
            r.close()


@pytest.mark.parametrize("run_len", [1, 37, 300, 5000, 1 << 30])
def test_dense_kernel_folds_waves_whose_rows_share_a_slot(pp, run_len, monkeypatch):
    """A table sorted by its group column: the rows of a wave fall into one slot and the specialised dense kernel folds them across the
    lanes before ONE LDS update (fdb_jit.cpp, "Sorted input") — single-phase shapes, SUM / MIN / MAX / COUNT, float64 and int64, with a
    filter that empties some lanes and whole waves, NULL group keys, runs of equal keys from 1 row (nothing folds) to the whole record
    (everything folds), ragged record sizes. Equal to the oracle, and bit-identical in the integer columns to the kernel without the
    fold (FDB_NO_UNIFORM_FOLD)."""
    from frostdb_amd.logicalplan import Max, Min
    rng = np.random.default_rng(run_len % 1000 + 7)
    n = 400_003
    n_path = 50
    starts = np.arange(0, n, min(run_len, n))
    key_of_run = rng.integers(0, n_path + 1, len(starts))  # n_path = NULL
    path = np.repeat(key_of_run, np.diff(np.append(starts, n)))[:n]
    code = rng.integers(0, 3, n).astype(np.uint32)
    code[100_000:103_000] = 1  # a stretch the filter removes altogether (whole waves without a selected row)
    rec = pa.RecordBatch.from_arrays(
        [pa.DictionaryArray.from_arrays(pa.array(code), pa.array([b"200", b"404", b"500"], type=pa.binary())),
         pa.DictionaryArray.from_arrays(pa.array(np.where(path == n_path, 0, path).astype(np.uint32), mask=path == n_path), pa.array([b"/p%02d" % i for i in range(n_path)], type=pa.binary())),
         pa.array(rng.uniform(-5, 1000, n))], names=["labels.code", "labels.path", "value"])
    filt = Col("labels.code") != "404"
    aggs = [Sum(Col("value")), Min(Col("value")), Max(Col("value")), Count(Col("value"))]
    groups = [Col("labels.path")]
    recs = [rec.slice(0, 150_001), rec.slice(150_001)]
    cols = ["labels.path"] + [a.Name() for a in aggs]
    want = run_oracle(recs, filt, aggs, groups)
    got = run_gpu(pp, recs, filt, aggs, groups, resident=True)
    assert_same_result(got, want, cols, float_cols={"sum(value)"})
    monkeypatch.setenv("FDB_NO_UNIFORM_FOLD", "1")
    plain = run_gpu(pp, recs, filt, aggs, groups, resident=True)
    assert_same_result(plain, want, cols, float_cols={"sum(value)"})
    for c in ("min(value)", "max(value)", "count(value)"):
        assert dict(zip(got["labels.path"], got[c])) == dict(zip(plain["labels.path"], plain[c]))


def test_filter_retries_through_the_three_launch_path_when_the_one_pass_kernel_stalls(pp, monkeypatch):
    """The one-pass select kernel's waits are bounded; running into the bound (a GPU shared with long kernels, a preempted queue —
    pretended here with FDB_TEST_SELECT_STALL) is not the caller's error: the same call filters the records again through
    flags → prefix sums → compaction and returns the same rows (it used to be FDB_ERR_DEVICE)."""
    rng = np.random.default_rng(23)
    recs = [make_prometheus_batch(rng, n, n_path=30, null_frac=0.02) for n in (300_000, 5, 70_001)]
    rbs = [pp.ResidentBatch(r) for r in recs]
    filt = Col("value") > 400.0
    want = [_oracle_filter(r, filt) for r in recs]
    try:
        for stall in (False, True):
            if stall:
                monkeypatch.setenv("FDB_TEST_SELECT_STALL", "1")
            plan = pp.HashAggregatePlan(filt)
            try:
                outs = plan.FilterResidentMany(rbs)
                assert ("fdb_select_kernel" in plan.last_kernel()) == (not stall), plan.last_kernel()
                for (w, idx), o in zip(want, outs):
                    g = arrow_to_pydict(o.to_arrow())
                    assert o.num_rows == len(idx)
                    if w is not None:  # (the oracle returns no record for an empty selection)
                        assert g["value"] == w["value"] and g["labels.path"] == w["labels.path"]
                    o.close()
            finally:
                plan.Close()
    finally:
        for r in rbs:
            r.close()
    assert pp.live_allocations()["device_blocks"] == 0
