"""OrderedAggregate on the device (fdb_plan_desc.ordered) against the Python restatement of the reference's operator
(tests/ordered_oracle.py, pinned on the reference's vectors in the CPU suite): the reference's vectors themselves, randomised
partially ordered streams with NULL keys, dynamic columns that come and go, MIN / MAX / COUNT, a partial-stage plan's column
naming, and — at a size the Python restatement cannot reach — sortedness + equality with the hash aggregate."""
import numpy as np
import pyarrow as pa
import pytest

from frostdb_amd.logicalplan import Col, Count, DynCol, Max, Min, Sum
from tests.golden.ordered_cases import ORDERED_CASES
from tests.ordered_oracle import COUNT, MAX, MIN, SUM, OrderedAggregate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import physicalplan
    assert physicalplan.device_count() >= 1
    return physicalplan


def to_record(rec):
    arrays, names = [], []
    for name, vals in rec.items():
        if all(isinstance(v, (bytes, type(None))) for v in vals) and any(isinstance(v, bytes) for v in vals) or name.startswith(("group", "labels")):
            arrays.append(pa.array(vals, type=pa.binary()))
        else:
            arrays.append(pa.array(vals, type=pa.int64()))
        names.append(name)
    return pa.RecordBatch.from_arrays(arrays, names=names)


def device_rows(pp, records, agg, groups, final_stage=True):
    plan = pp.HashAggregatePlan(None, [agg], groups, ordered=True, final_stage=False)
    try:
        for r in records:
            plan.Callback(to_record(r))
        out = plan.Finish()
    finally:
        plan.Close()
    cols = [c.to_pylist() for c in out.columns]
    return out.schema.names, [tuple(c[i] for c in cols) for i in range(out.num_rows)]


@pytest.mark.parametrize("case", ORDERED_CASES, ids=[c["id"] for c in ORDERED_CASES])
def test_reference_vectors(pp, case):
    recs = []
    for groups, vals in case["records"]:
        rec = {"group%d" % i: [g.encode() if g else None for g in col] for i, col in enumerate(groups) if col}
        rec["vals"] = [v or None for v in vals]
        recs.append(rec)
    names, rows = device_rows(pp, recs, Sum(Col("vals")), [Col("group%d" % i) for i in range(case["ncols"])])
    want = [tuple(x.encode() if isinstance(x, str) else x for x in r) for r in case["expected"]]
    assert rows == want, case["cite"]
    assert names[-1] == "vals"  # a partial-stage OrderedAggregate names its result after the column (ordered_aggregate.go:551-557)


@pytest.mark.parametrize("func,agg", [(SUM, Sum), (MIN, Min), (MAX, Max), (COUNT, Count)])
def test_random_partially_ordered_streams_with_null_keys(pp, func, agg):
    rng = np.random.default_rng(int(func) * 7)
    keys_a = [None, b"a", b"b", b"c", b"d"]
    recs = []
    for _ in range(5):
        n = int(rng.integers(1, 60))
        a = sorted(rng.integers(0, 5, n).tolist()) if rng.random() < 0.7 else rng.integers(0, 5, n).tolist()
        b = rng.integers(0, 3, n).tolist()
        recs.append({"group0": [keys_a[i] for i in a], "group1": [None if i == 0 else b"k%d" % i for i in b],
                     "vals": [int(v) for v in rng.integers(1, 100, n)]})
    o = OrderedAggregate(func, "vals", [("group0", False), ("group1", False)], final_stage=False)
    for r in recs:
        o.callback(r)
    want = o.finish()["rows"]
    _, rows = device_rows(pp, recs, agg(Col("vals")), [Col("group0"), Col("group1")])
    # Reference quirk (restated faithfully by the Python oracle): cursorHeap.Less returns "not less" as soon as BOTH sides are NULL in
    # a column (merge.go:91-98) without looking at the later columns, so among rows whose FIRST key is NULL the merged order is not
    # sorted by the second key, equal keys are not adjacent, and the final re-grouping leaves them as SEVERAL rows per group. The
    # device merges by key tuple, so those groups come out whole: the oracle's NULL-first rows are folded per key before comparing
    # (documented difference); everything before them must agree row for row, in order.
    def fold(rs):
        head = [r for r in rs if r[0] is not None]
        tail = {}
        for r in (r for r in rs if r[0] is None):
            tail.setdefault(r[:-1], []).append(r[-1])
        red = {SUM: sum, COUNT: sum, MIN: min, MAX: max}[func]
        return head, sorted(((k + (red(v),)) for k, v in tail.items()), key=lambda r: (r[1] is None, r[1] or b""))
    assert fold(rows) == fold(want)


def test_dynamic_columns_come_and_go(pp):
    recs = []
    for i in range(4):
        rec = {"labels.0": [b"group"] * 10}
        if i:
            rec["labels.%d" % i] = [b"group"] * 10
        rec["value"] = [1] * 10
        recs.append(rec)
    names, rows = device_rows(pp, recs, Sum(Col("value")), [DynCol("labels")])
    assert len(rows) == 4 and len(names) == 5 and all(r[-1] == 10 for r in rows)  # TestOrderedAggregateDynCols
    o = OrderedAggregate(SUM, "value", [("labels", True)], final_stage=False)
    for r in recs:
        o.callback(r)
    assert sorted(rows, key=repr) == sorted(o.finish()["rows"], key=repr)


def test_large_sorted_input_is_the_hash_aggregate_in_key_order(pp):
    """2 M rows sorted by (int64 bucket, label): the ordered plan's record equals the hash aggregate's rows and is sorted by key,
    NULL labels last within a bucket; the final-stage naming is the aggregation's."""
    rng = np.random.default_rng(1)
    n = 2_000_000
    bucket = np.sort(rng.integers(1, 5000, n)).astype(np.int64)
    lab = rng.integers(0, 6, n)
    order = np.lexsort((lab, bucket))
    bucket, lab = bucket[order], lab[order]
    labels = pa.DictionaryArray.from_arrays(pa.array(lab.astype(np.uint32), mask=lab == 5), pa.array([b"e", b"d", b"c", b"b", b"a"], type=pa.binary()))
    rec = pa.RecordBatch.from_arrays([pa.array(bucket), labels, pa.array(rng.integers(0, 1000, n).astype(np.int64))], names=["bucket", "labels.x", "v"])
    res = {}
    for ordered in (False, True):
        plan = pp.HashAggregatePlan(None, [Sum(Col("v"))], [Col("bucket"), Col("labels.x")], ordered=ordered, final_stage=False)
        rb = pp.ResidentBatch(rec)
        try:
            plan.Callback(rb)
            out = plan.Finish()
        finally:
            plan.Close()
            rb.close()
        res[ordered] = out
    o, h = res[True], res[False]
    assert o.schema.names == ["bucket", "labels.x", "v"] and h.schema.names == ["bucket", "labels.x", "sum(v)"]
    orows = list(zip(o.column(0).to_pylist(), o.column(1).dictionary_decode().to_pylist(), o.column(2).to_pylist()))
    hrows = list(zip(h.column(0).to_pylist(), h.column(1).dictionary_decode().to_pylist(), h.column(2).to_pylist()))
    key = lambda r: (r[0], r[1] is None, r[1] or b"")  # noqa: E731
    assert orows == sorted(hrows, key=key)
    assert len(orows) > 20_000
