"""OrderedAggregate on the device (fdb_plan_desc.ordered) against the Python restatement of the reference's operator
(tests/ordered_oracle.py, pinned on the reference's vectors in the CPU suite): the reference's vectors themselves, randomised
partially ordered streams with NULL keys, dynamic columns that come and go, MIN / MAX / COUNT, a partial-stage plan's column
naming, and — at a size the Python restatement cannot reach — sortedness + equality with the hash aggregate."""
import numpy as np
import pyarrow as pa
import pytest

from frostdb_amd.logicalplan import Col, Count, DynCol, Max, Min, Sum, Unique
from tests.golden.ordered_cases import ORDERED_CASES
from tests.ordered_oracle import COUNT, MAX, MIN, SUM, OrderedAggregate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import physicalplan
    assert physicalplan.device_count() >= 1
    return physicalplan


@pytest.fixture(autouse=True)
def _runs_on_small_shapes(monkeypatch):
    """Ordered plans over SMALL key spaces (≤ 8 192 dense slots) keep the dense table since round 5 (Plan::runs_wanted); these tests are
    about the run store, so they ask for it on small shapes too. test_small_key_spaces_keep_the_dense_table checks the default."""
    monkeypatch.setenv("FDB_RUNS_ALWAYS", "1")


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


@pytest.mark.parametrize("records", ["narrow", "medium", "wide"])
@pytest.mark.parametrize("case", ORDERED_CASES, ids=[c["id"] for c in ORDERED_CASES])
def test_reference_vectors(pp, case, records, monkeypatch):
    """ordered_aggregate_test.go's vectors through the run store with each of its three run records (FDB_RUNS_WIDE forces the wider ones)."""
    if records != "narrow":
        monkeypatch.setenv("FDB_RUNS_WIDE", "m" if records == "medium" else "1")
    recs = []
    for groups, vals in case["records"]:
        rec = {"group%d" % i: [g.encode() if g else None for g in col] for i, col in enumerate(groups) if col}
        rec["vals"] = [v or None for v in vals]
        recs.append(rec)
    names, rows = device_rows(pp, recs, Sum(Col("vals")), [Col("group%d" % i) for i in range(case["ncols"])])
    want = [tuple(x.encode() if isinstance(x, str) else x for x in r) for r in case["expected"]]
    assert rows == want, case["cite"]
    assert names[-1] == "vals"  # a partial-stage OrderedAggregate names its result after the column (ordered_aggregate.go:551-557)


@pytest.mark.parametrize("fallback", ["sort", "table", "table-sorted-on-the-device"])
@pytest.mark.parametrize("records", ["narrow", "medium", "wide"])
@pytest.mark.parametrize("func,agg", [(SUM, Sum), (MIN, Min), (MAX, Max), (COUNT, Count)])
def test_random_partially_ordered_streams_with_null_keys(pp, func, agg, records, fallback, monkeypatch):
    if fallback != "sort":  # (records out of order: the runs are sorted on the device by default; with this they go through the hash table …)
        monkeypatch.setenv("FDB_RUNS_NO_SORT", "1")
    if fallback == "table-sorted-on-the-device":  # (… whose groups an ordered Finish sorts on the host when there are few, on the device from 4 096 on: here always)
        monkeypatch.setenv("FDB_ORDERED_SORT_MIN", "0")
    if records != "narrow":
        monkeypatch.setenv("FDB_RUNS_WIDE", "m" if records == "medium" else "1")
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


# ---- the table-free path (round 4): runs of equal keys collected by the scan, merged and checked for order at Finish ------------------

def _sorted_label_records(rng, n_total, n_records, card=(5, 7, 3), null_frac=0.03, sort=True):
    """Rows over three dictionary label columns, ordered by (l0, l1, l2) — values ascending bytewise, NULLs last, the order an
    OrderedAggregate's input has — cut into records at arbitrary rows (groups straddle record boundaries)."""
    cols = []
    for c, k in enumerate(card):
        v = rng.integers(0, k + 1, n_total)  # k = NULL
        cols.append(np.where(rng.random(n_total) < null_frac, k, v))
    if sort:
        order = np.lexsort(tuple(reversed(cols)))
        cols = [c[order] for c in cols]
    val = rng.integers(-50, 1000, n_total).astype(np.int64)
    fval = rng.uniform(0, 100, n_total)
    cuts = [0] + sorted(rng.integers(1, n_total, n_records - 1).tolist()) + [n_total]
    recs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        arrays, names = [], []
        for c, k in enumerate(card):
            # dictionary in DESCENDING value order, so that key ids / dictionary indices do not happen to be ranks
            d = pa.array([b"v%02d" % (k - 1 - i) for i in range(k)], type=pa.binary())
            x = cols[c][a:b]
            idx = pa.array(np.where(x == k, 0, k - 1 - x).astype(np.uint32), mask=x == k)
            arrays.append(pa.DictionaryArray.from_arrays(idx, d)); names.append("labels.l%d" % c)
        arrays += [pa.array(val[a:b]), pa.array(fval[a:b])]
        names += ["v", "f"]
        recs.append(pa.RecordBatch.from_arrays(arrays, names=names))
    return recs


def _run_plan(pp, recs, agg, groups, ordered, resident=False, filt=None, finish_resident=False):
    plan = pp.HashAggregatePlan(filt, [agg], groups, ordered=ordered, final_stage=False)
    keep = []
    try:
        if resident:
            keep = [pp.ResidentBatch(r) for r in recs]
            plan.CallbackResident(keep)
        else:
            for r in recs:
                plan.Callback(r)
        kernel = plan.last_kernel()
        if finish_resident:
            rb = plan.FinishResident()
            out = rb.to_arrow()
            rb.close()
        else:
            out = plan.Finish()
        _run_plan.after_finish = plan.last_kernel()  # (an ordered Finish that had to sort its runs names the kernels it ran)
        return out, kernel
    finally:
        plan.Close()
        for k in keep:
            k.close()


def _rows(out):
    cols = [(c.dictionary_decode() if pa.types.is_dictionary(c.type) else c).to_pylist() for c in out.columns]
    return [tuple(c[i] for c in cols) for i in range(out.num_rows)]


def _key_order(r, nkeys=3):
    return tuple((x is None, x or b"") for x in r[:nkeys])


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("agg_name", ["sum_i", "sum_f", "min", "max", "count"])
def test_table_free_ordered_aggregate_of_sorted_records(pp, agg_name, resident):
    """Sorted label columns, 6 records cut at arbitrary rows: the scan collects runs (no hash kernel ran: last_kernel says so),
    Finish merges the runs that wave and record boundaries cut and emits the groups in key order — equal to the hash aggregate's
    groups sorted by key (values bit-exact for integers; float sums to 1e-9, the fold order differs)."""
    rng = np.random.default_rng(11)
    recs = _sorted_label_records(rng, 300_000, 6)
    agg = {"sum_i": Sum(Col("v")), "sum_f": Sum(Col("f")), "min": Min(Col("v")), "max": Max(Col("f")), "count": Count(Col("v"))}[agg_name]
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    o, kernel = _run_plan(pp, recs, agg, groups, ordered=True, resident=resident)
    assert kernel == "fdb_hash_kernel(runs)", kernel
    h, hk = _run_plan(pp, recs, agg, groups, ordered=False, resident=resident)
    assert "runs" not in hk
    orows, hrows = _rows(o), sorted(_rows(h), key=_key_order)
    assert len(orows) == len(hrows) > 100
    assert [r[:3] for r in orows] == [r[:3] for r in hrows]  # same groups, in key order (NULLs last in every column)
    for a, b in zip(orows, hrows):
        assert a[3] == b[3] or (isinstance(a[3], float) and abs(a[3] - b[3]) <= 1e-9 * max(1.0, abs(b[3]))), (a, b)
    assert o.schema.names[:3] == ["labels.l0", "labels.l1", "labels.l2"]
    assert o.schema.names[3] == ("v" if agg_name not in ("sum_f", "max") else "f")  # partial-stage naming (ordered_aggregate.go:551-557)


def test_table_free_ordered_aggregate_with_a_filter_and_a_resident_finish(pp):
    rng = np.random.default_rng(12)
    recs = _sorted_label_records(rng, 200_000, 3)
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    filt = Col("v") > 300
    o, kernel = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True, resident=True, filt=filt, finish_resident=True)
    assert kernel == "fdb_hash_kernel(runs)"
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False, resident=True, filt=filt)
    assert _rows(o) == sorted(_rows(h), key=_key_order)


SORTED_FINISH = "runs_sort_keys_kernel + runs_expand_kernel"


@pytest.mark.parametrize("fallback", ["sort", "table"])
def test_input_that_breaks_the_order_is_sorted_on_the_device_or_falls_back_to_the_table(pp, fallback, monkeypatch):
    """One record out of order (and one that is not sorted at all): Finish notices that a new key does not sort after its
    predecessor and sorts the runs by key on the device (round 5) — or, with $FDB_RUNS_NO_SORT, puts every run into the hash table and
    takes the ordinary ordered Finish (round 4). Same groups, same order."""
    if fallback == "table":
        monkeypatch.setenv("FDB_RUNS_NO_SORT", "1")
    rng = np.random.default_rng(13)
    recs = _sorted_label_records(rng, 120_000, 4)
    recs = [recs[2], recs[0], recs[3], recs[1]] + _sorted_label_records(rng, 30_000, 1, sort=False)
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    o, kernel = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True)
    assert kernel == "fdb_hash_kernel(runs)"
    assert (_run_plan.after_finish == SORTED_FINISH) == (fallback == "sort"), _run_plan.after_finish
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False)
    assert _rows(o) == sorted(_rows(h), key=_key_order)


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("records", ["narrow", "medium", "wide"])
@pytest.mark.parametrize("agg_name", ["sum_i", "sum_f", "min", "max", "count"])
def test_several_ordered_sets_are_merged_without_the_table(pp, agg_name, records, resident, monkeypatch):
    """Four ordered sets (≙ the sets OrderedAggregate merges at Finish, ordered_aggregate.go:449-470): every record sorted, pushed in an
    order that is not the key order, groups shared between the sets. No hash kernel: the runs are sorted by key on the device and the
    runs of one group folded where they meet. Equal to the hash aggregate's groups sorted by key — with each of the three run records."""
    if records != "narrow":
        monkeypatch.setenv("FDB_RUNS_WIDE", "m" if records == "medium" else "1")
    rng = np.random.default_rng(31)
    sets = [_sorted_label_records(rng, 60_000, 2, card=(6, 9, 4)) for _ in range(4)]  # (each set: two records that continue each other)
    recs = [r for st in (sets[2], sets[0], sets[3], sets[1]) for r in st]
    agg = {"sum_i": Sum(Col("v")), "sum_f": Sum(Col("f")), "min": Min(Col("v")), "max": Max(Col("f")), "count": Count(Col("v"))}[agg_name]
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    o, kernel = _run_plan(pp, recs, agg, groups, ordered=True, resident=resident, finish_resident=resident)
    assert kernel.startswith("fdb_hash_kernel(runs"), kernel
    assert _run_plan.after_finish == SORTED_FINISH, _run_plan.after_finish
    h, _ = _run_plan(pp, recs, agg, groups, ordered=False, resident=resident)
    orows, hrows = _rows(o), sorted(_rows(h), key=_key_order)
    assert len(orows) == len(hrows) > 300
    assert [r[:3] for r in orows] == [r[:3] for r in hrows]
    for a, b in zip(orows, hrows):
        assert a[3] == b[3] or (isinstance(a[3], float) and abs(a[3] - b[3]) <= 1e-9 * max(1.0, abs(b[3]))), (a, b)


TABLE_SORTED_FINISH = "hash_gather_rows_kernel + runs_sort_keys_kernel"


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("agg_name", ["sum_i", "sum_f", "min", "count"])
def test_ordered_plans_merged_like_chains_finish_in_key_order_on_the_device(pp, agg_name, resident):
    """Two ordered plans (two chains over halves of a sorted table), merged (≙ OrderedSynchronizer + the final ordered aggregate,
    ordered_synchronizer.go:59-116): the second plan's runs are re-keyed into the first plan's key ids (runs_translate_kernel) and become
    one more ordered set of its run store — no hash table is built (round 6; round 5 inserted every run into the table) — and Finish brings
    the sets into key order on the device. Equal to the hash aggregate sorted by key; int64 and NULL keys included (wide run records)."""
    rng = np.random.default_rng(41)
    recs = _wide_sorted_records(rng, 200_000, 6, cards=(300, 2_000), int_key=1)
    agg = {"sum_i": Sum(Col("v")), "sum_f": Sum(Col("v")), "min": Min(Col("v")), "count": Count(Col("v"))}[agg_name]
    groups = [Col("labels.l0"), Col("bucket")]
    p1 = pp.HashAggregatePlan(None, [agg], groups, ordered=True, final_stage=False)
    p2 = pp.HashAggregatePlan(None, [agg], groups, ordered=True, final_stage=False)
    try:
        for r in recs[:3]:
            p1.Callback(r)
        for r in recs[3:]:
            p2.Callback(r)
        p1.Merge(p2)
        assert p1.last_kernel() == "runs_translate_kernel", p1.last_kernel()
        if resident:
            rb = p1.FinishResident()
            out = rb.to_arrow()
            rb.close()
        else:
            out = p1.Finish()
        # (the second chain's keys all sort behind the first chain's: the appended set continues the order and Finish needs no sort — no kernel of the table path ran either way)
        assert p1.last_kernel() in ("runs_translate_kernel", SORTED_FINISH), p1.last_kernel()
    finally:
        p1.Close(); p2.Close()
    h, _ = _run_plan(pp, recs, agg, groups, ordered=False)
    key = lambda r: tuple((x is None, x if x is not None else 0) for x in r[:2])  # noqa: E731
    orows = _rows(out)
    assert len(orows) > 50_000 and orows == sorted(_rows(h), key=key)


@pytest.mark.parametrize("shape", ["narrow", "medium", "drifting"])
def test_ordered_plans_with_their_own_dictionaries_merge_as_runs(pp, shape, monkeypatch):
    """The merge of ordered plans as runs across the record formats: three plans over thirds of a sorted table whose records carry
    DIFFERENT dictionaries (every record's dictionary in its own order, so the plans' key ids disagree), merged pairwise. `narrow`: a byte
    per key id stays a byte; `medium`: 300-value dictionaries (two bytes); `drifting`: the third plan has a label column the others lack and
    lacks one they have (the merged record is wide where a narrow one cannot say "absent"... it can: id 0). Equal to the hash aggregate
    over all records, in key order; the merged plan never runs a hash kernel."""
    rng = np.random.default_rng({"narrow": 51, "medium": 52, "drifting": 53}[shape])
    n = 120_000
    cards = (300, 40, 9) if shape == "medium" else (40, 9, 5)
    cols = [rng.integers(0, k + 1, n) for k in cards]  # k = NULL
    order = np.lexsort(tuple(reversed(cols)))
    cols = [c[order] for c in cols]
    val = rng.integers(-50, 1000, n).astype(np.int64)
    cuts = [0, n // 3, 2 * n // 3, n]
    recs = []
    for part, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        arrays, names = [], []
        for c, k in enumerate(cards):
            if shape == "drifting" and part == 2 and c == 1:
                continue  # the third plan never sees labels.l1 …
            perm = rng.permutation(k)  # this record's dictionary order: entry perm[v] holds value v
            d = [None] * k
            for v in range(k):
                d[perm[v]] = b"v%03d" % v
            x = cols[c][a:b]
            idx = pa.array(np.where(x == k, 0, perm[np.minimum(x, k - 1)]).astype(np.uint32), mask=x == k)
            arrays.append(pa.DictionaryArray.from_arrays(idx, pa.array(d, type=pa.binary()))); names.append("labels.l%d" % c)
        if shape == "drifting" and part == 2:  # … and brings a column of its own (constant: the rows stay in key order)
            arrays.append(pa.DictionaryArray.from_arrays(pa.array(np.zeros(b - a, dtype=np.uint32)), pa.array([b"only-here"], type=pa.binary()))); names.append("labels.l9")
        arrays.append(pa.array(val[a:b])); names.append("v")
        recs.append(pa.RecordBatch.from_arrays(arrays, names=names))
    groups = [DynCol("labels")]
    plans = [pp.HashAggregatePlan(None, [Sum(Col("v"))], groups, ordered=True, final_stage=False) for _ in recs]
    keep = [pp.ResidentBatch(r) for r in recs]
    try:
        for p, k in zip(plans, keep):
            p.CallbackResident([k])
            assert p.last_kernel().startswith("fdb_hash_kernel(runs"), p.last_kernel()
        plans[0].Merge(plans[2])
        plans[0].Merge(plans[1])
        assert plans[0].last_kernel() == "runs_translate_kernel"
        out = plans[0].Finish()
        assert plans[0].last_kernel() == SORTED_FINISH, plans[0].last_kernel()
    finally:
        for p in plans:
            p.Close()
        for k in keep:
            k.close()
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False, resident=True)
    assert out.schema.names[:-1] == h.schema.names[:-1]
    nk = out.num_columns - 1
    orows = _rows(out)
    assert len(orows) > 1000 and orows == sorted(_rows(h), key=lambda r: _key_order(r, nk))


def test_small_ordered_results_out_of_the_table_are_still_sorted_on_the_host(pp, monkeypatch):
    """Below 4 096 groups the ordered Finish out of the table keeps the host sort (a handful of launches cost more than sorting a few rows);
    $FDB_ORDERED_SORT_MIN=0 sends the same result through the device sort — both equal the hash aggregate sorted by key."""
    rng = np.random.default_rng(42)
    recs = _sorted_label_records(rng, 50_000, 4)
    recs = [recs[1], recs[3], recs[0], recs[2]]
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    monkeypatch.setenv("FDB_RUNS_NO_SORT", "1")
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False)
    want = sorted(_rows(h), key=_key_order)
    o, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True)
    assert _run_plan.after_finish not in (SORTED_FINISH, TABLE_SORTED_FINISH), _run_plan.after_finish
    assert _rows(o) == want
    monkeypatch.setenv("FDB_ORDERED_SORT_MIN", "0")
    for finish_resident in (False, True):
        o, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True, resident=finish_resident, finish_resident=finish_resident)
        assert _run_plan.after_finish == TABLE_SORTED_FINISH, _run_plan.after_finish
        assert _rows(o) == want


@pytest.mark.parametrize("pos", [0, 1])
def test_ordered_sets_with_int64_keys_and_big_dictionaries_are_sorted_by_value(pp, pos):
    """The sort's int64 passes (value with the sign bit flipped, then NULL last) and a dictionary of 70 000 values (17 bits of rank) next to
    them: three ordered sets with negative, positive and NULL buckets, pushed out of order."""
    rng = np.random.default_rng(33 + pos)
    cards = (3_000, 70_000) if pos == 0 else (70_000, 3_000)
    sets = [_wide_sorted_records(rng, 80_000, 2, cards=cards, int_key=pos) for _ in range(3)]
    recs = [r for st in (sets[1], sets[2], sets[0]) for r in st]
    groups = [Col("bucket"), Col("labels.l1")] if pos == 0 else [Col("labels.l0"), Col("bucket")]
    o, kernel = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True, resident=True)
    assert kernel == "fdb_hash_kernel(runs, wide)", kernel
    assert _run_plan.after_finish == SORTED_FINISH, _run_plan.after_finish
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False, resident=True)
    key = lambda r: tuple((x is None, x if x is not None else 0) for x in r[:2])  # noqa: E731
    orows = _rows(o)
    assert len(orows) > 100_000 and orows == sorted(_rows(h), key=key)


def test_ordered_unique_aggregation_finishes_out_of_the_table_in_key_order(pp):
    """UNIQUE is a composite reducer (MIN and MAX side by side, NULL where they differ): such a plan never collects runs, its groups sit in
    the hash table with THREE value arrays (count, min, max), and the ordered Finish sorts them on the device like any other — the value
    arrays are gathered into key order together. Host Arrow result and resident result (which takes the host route for composites)."""
    rng = np.random.default_rng(53)
    recs = _wide_sorted_records(rng, 120_000, 3, cards=(60, 70_000), int_key=None)  # (4 M key combinations: the hash table, not the dense one)
    # v: constant within most groups (UNIQUE keeps it), different within some (UNIQUE → NULL)
    fixed = []
    for r in recs:
        l0 = r.column(0).indices.to_numpy(zero_copy_only=False)  # (dictionary indices; NaN where the key is NULL)
        l1 = r.column(1).indices.to_numpy(zero_copy_only=False)
        base = (np.nan_to_num(l0.astype(np.float64), nan=-1).astype(np.int64) * 100_000 + np.nan_to_num(l1.astype(np.float64), nan=-1).astype(np.int64))
        v = np.where(base % 5 == 0, rng.integers(0, 3, len(base)), 7) + base
        fixed.append(pa.RecordBatch.from_arrays([r.column(0), r.column(1), pa.array(v.astype(np.int64))], names=["labels.l0", "labels.l1", "v"]))
    groups = [Col("labels.l0"), Col("labels.l1")]
    h, _ = _run_plan(pp, fixed, Unique(Col("v")), groups, ordered=False)
    want = sorted(_rows(h), key=lambda r: _key_order(r, 2))
    assert len(want) > 20_000 and any(r[2] is None for r in want) and any(r[2] is not None for r in want)
    for finish_resident in (False, True):
        o, kernel = _run_plan(pp, [fixed[1], fixed[0], fixed[2]], Unique(Col("v")), groups, ordered=True, resident=finish_resident, finish_resident=finish_resident)
        assert kernel in ("fdb_hash_kernel", "scan_hash_kernel"), kernel
        assert _run_plan.after_finish == TABLE_SORTED_FINISH, _run_plan.after_finish
        assert _rows(o) == want


def test_tiny_ordered_results_through_every_sort(pp, monkeypatch):
    """Zero, one and two groups through the run store's sort (two runs out of order), the table fallback sorted on the device
    ($FDB_ORDERED_SORT_MIN=0) and a filter that selects nothing."""
    d = pa.array([b"b", b"a"], type=pa.binary())

    def rec(ids, vals):
        return pa.RecordBatch.from_arrays([pa.DictionaryArray.from_arrays(pa.array(np.array(ids, dtype=np.uint32)), d), pa.array(np.array(vals, dtype=np.int64))], names=["labels.k", "v"])
    groups = [Col("labels.k")]
    for env in ({}, {"FDB_RUNS_NO_SORT": "1", "FDB_ORDERED_SORT_MIN": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o, _ = _run_plan(pp, [rec([0, 0], [1, 2]), rec([1, 1], [10, 20])], Sum(Col("v")), groups, ordered=True)  # b, b then a, a: out of order
        assert _rows(o) == [(b"a", 30), (b"b", 3)]
        o, _ = _run_plan(pp, [rec([0], [5])], Sum(Col("v")), groups, ordered=True)
        assert _rows(o) == [(b"b", 5)]
        o, _ = _run_plan(pp, [rec([0, 1, 0], [1, 2, 4])], Sum(Col("v")), groups, ordered=True, filt=Col("v") > 100)
        assert o.num_rows == 0
        o, _ = _run_plan(pp, [rec([0, 1, 0], [1, 2, 4])], Sum(Col("v")), groups, ordered=True)  # b a b: three runs of two groups in one record
        assert _rows(o) == [(b"a", 2), (b"b", 5)]


def test_ordered_sets_with_uint64_keys_beyond_the_sign_bit(pp):
    """A uint64 group key (the bytes schema's timestamps, logic_test.go:110-146) with values on both sides of 2^63 and NULLs, behind a dictionary
    column: the sort's value pass must NOT flip the sign bit for it (unsigned order), the order check compares unsigned, NULLs last. Three
    ordered sets pushed out of order."""
    rng = np.random.default_rng(51)
    n = 90_000
    lab = rng.integers(0, 41, n)  # 40 = NULL
    ts = rng.integers(1, 1 << 62, n).astype(np.uint64) * np.uint64(4) + np.uint64(1)  # spread over [1, 2^64): never 0 (0 and NULL are one group in the hash table)
    ts = (ts // np.uint64(1 << 50)) * np.uint64(1 << 50) + np.uint64(1)              # ≈ 16 k distinct values
    tnull = rng.random(n) < 0.03
    sets = []
    for part in np.array_split(rng.permutation(n), 3):
        o = part[np.lexsort((np.where(tnull[part], np.uint64(0xFFFFFFFFFFFFFFFF), ts[part]), lab[part]))]
        d = pa.array([b"v%02d" % (39 - i) for i in range(40)], type=pa.binary())
        sets.append(pa.RecordBatch.from_arrays(
            [pa.DictionaryArray.from_arrays(pa.array(np.where(lab[o] == 40, 0, 39 - lab[o]).astype(np.uint32), mask=lab[o] == 40), d),
             pa.array(np.where(tnull[o], np.uint64(0), ts[o]), type=pa.uint64(), mask=tnull[o]), pa.array(rng.integers(1, 100, len(o)).astype(np.int64))],
            names=["labels.x", "ts", "v"]))
    groups = [Col("labels.x"), Col("ts")]
    o, kernel = _run_plan(pp, sets, Sum(Col("v")), groups, ordered=True)
    assert kernel == "fdb_hash_kernel(runs, wide)", kernel
    assert _run_plan.after_finish == SORTED_FINISH, _run_plan.after_finish
    h, _ = _run_plan(pp, sets, Sum(Col("v")), groups, ordered=False)
    key = lambda r: (r[0] is None, r[0] or b"", r[1] is None, r[1] if r[1] is not None else 0)  # noqa: E731
    orows = _rows(o)
    assert len(orows) > 40_000 and any(r[1] is not None and r[1] >= 1 << 63 for r in orows) and orows == sorted(_rows(h), key=key)


def test_forty_group_columns_out_of_order_are_sorted_through_wide_records(pp):
    """More group columns than a narrow or medium run record holds (40 > 32): wide records, whose sort keys come out of the records' key
    tuples word by word; rows not sorted at all, columns that are constant, columns with NULLs, one column that only the second record has."""
    rng = np.random.default_rng(52)
    n, n_cols = 60_000, 40
    g = rng.integers(0, 9_000, n)
    arrays, names = [], []
    for c in range(n_cols):
        k = 1 if c % 7 == 3 else 6
        x = (g // (1 + c)) % (k + 1) if k > 1 else np.zeros(n, dtype=np.int64)  # k = NULL (never for the constant columns)
        d = pa.array([b"c%02d-%d" % (c, k - 1 - i) for i in range(k)], type=pa.binary())
        arrays.append(pa.DictionaryArray.from_arrays(pa.array(np.where(x == k, 0, k - 1 - x).astype(np.uint32), mask=(x == k) if k > 1 else None), d))
        names.append("labels.l%02d" % c)
    arrays.append(pa.array(rng.integers(1, 50, n).astype(np.int64))); names.append("v")
    rec = pa.RecordBatch.from_arrays(arrays, names=names)
    first = rec.slice(0, 25_000).drop_columns(["labels.l39"])  # (the first record does not know the last column: NULL there)
    recs = [first, rec.slice(25_000)]
    groups = [DynCol("labels")]
    o, kernel = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True)
    assert kernel == "fdb_hash_kernel(runs, wide)", kernel
    assert _run_plan.after_finish == SORTED_FINISH, _run_plan.after_finish
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False)
    onames = o.schema.names[:-1]
    hidx = [h.schema.names.index(nm) for nm in onames] + [h.num_columns - 1]
    hrows = [tuple(r[i] for i in hidx) for r in _rows(h)]
    orows = _rows(o)
    assert len(onames) == n_cols and len(orows) > 8_000 and orows == sorted(hrows, key=lambda r: _key_order(r, n_cols))


def test_ordered_sets_whose_key_ranks_need_several_sort_passes(pp):
    """Fourteen label columns of 90–130 values (7 bits of rank each: 98 bits, two 64-bit passes) over rows that are NOT sorted at all —
    every row a run of its own, 200 000 runs into ≈ 60 000 groups: the least significant columns must be sorted first and the passes
    must be stable for the result to come out in key order."""
    rng = np.random.default_rng(35)
    n, n_cols = 200_000, 14
    cards = [int(c) for c in rng.integers(90, 131, n_cols)]
    base = rng.integers(0, 60_000, n)  # the group of every row; its key: a fixed random tuple per group
    tuples = np.stack([rng.integers(0, k + 1, 60_000) for k in cards], axis=1)  # k = NULL
    arrays, names = [], []
    for c, k in enumerate(cards):
        x = tuples[base, c]
        d = pa.array([b"v%03d" % (k - 1 - i) for i in range(k)], type=pa.binary())
        arrays.append(pa.DictionaryArray.from_arrays(pa.array(np.where(x == k, 0, k - 1 - x).astype(np.uint32), mask=x == k), d)); names.append("labels.l%02d" % c)
    arrays.append(pa.array(rng.integers(1, 100, n).astype(np.int64))); names.append("v")
    rec = pa.RecordBatch.from_arrays(arrays, names=names)
    groups = [Col("labels.l%02d" % c) for c in range(n_cols)]
    o, kernel = _run_plan(pp, [rec.slice(0, 120_000), rec.slice(120_000)], Sum(Col("v")), groups, ordered=True)
    assert kernel.startswith("fdb_hash_kernel(runs"), kernel
    assert _run_plan.after_finish == SORTED_FINISH, _run_plan.after_finish
    h, _ = _run_plan(pp, [rec], Sum(Col("v")), groups, ordered=False)
    orows = _rows(o)
    assert len(orows) > 50_000 and orows == sorted(_rows(h), key=lambda r: _key_order(r, n_cols))


def test_runs_become_table_entries_for_every_other_consumer(pp):
    """Merge of two ordered plans, the group count, the partial keys: anything but Finish first inserts the collected runs into the
    hash table (Plan::runs_to_table) — results equal the hash aggregate's."""
    rng = np.random.default_rng(14)
    recs = _sorted_label_records(rng, 150_000, 4)
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    p1 = pp.HashAggregatePlan(None, [Sum(Col("v"))], groups, ordered=True)
    p2 = pp.HashAggregatePlan(None, [Sum(Col("v"))], groups, ordered=True)
    try:
        for r in recs[:2]:
            p1.Callback(r)
        for r in recs[2:]:
            p2.Callback(r)
        assert p1.last_kernel() == "fdb_hash_kernel(runs)" and p2.last_kernel() == "fdb_hash_kernel(runs)"
        n2 = p2.num_groups()
        p1.Merge(p2)
        out = p1.Finish()
    finally:
        p1.Close(); p2.Close()
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False)
    assert _rows(out) == sorted(_rows(h), key=_key_order)
    h2, _ = _run_plan(pp, recs[2:], Sum(Col("v")), groups, ordered=False)
    assert n2 == h2.num_rows


def test_records_that_do_not_fit_the_narrow_run_record_take_the_wide_one(pp):
    """A dictionary that outgrows one byte of key ids, and a record that lacks one of the plan's group columns: round 4 left the
    table-free path there; now such a record's launch writes WIDE run records (the table's own key tuple, from re-loaded columns)
    and Finish merges narrow and wide segments alike."""
    rng = np.random.default_rng(15)
    recs = _sorted_label_records(rng, 60_000, 2)
    n = 5_000
    wide = pa.RecordBatch.from_arrays(
        [pa.DictionaryArray.from_arrays(pa.array(np.sort(rng.integers(0, 400, n)).astype(np.uint32)), pa.array([b"w%03d" % i for i in range(400)], type=pa.binary())),
         pa.DictionaryArray.from_arrays(pa.array(np.zeros(n, dtype=np.uint32)), pa.array([b"v00"], type=pa.binary())),
         pa.DictionaryArray.from_arrays(pa.array(np.zeros(n, dtype=np.uint32)), pa.array([b"v00"], type=pa.binary())),
         pa.array(rng.integers(0, 9, n).astype(np.int64)), pa.array(rng.uniform(0, 1, n))], names=["labels.l0", "labels.l1", "labels.l2", "v", "f"])
    missing = recs[1].drop_columns(["labels.l1"])
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    for extra, last in ((wide, "fdb_hash_kernel(runs, medium)"), (missing, "fdb_hash_kernel(runs)")):
        seq = [recs[0], extra, recs[1]]
        o, kernel = _run_plan(pp, seq, Sum(Col("v")), groups, ordered=True)
        # (the first sequence's last record meets a plan whose l0 has 400 values: medium records — two bytes per key id; in the second the
        # record that lacks l1 wrote wide records and the last one narrow ones again)
        assert kernel == last, kernel
        h, _ = _run_plan(pp, seq, Sum(Col("v")), groups, ordered=False)
        assert _rows(o) == sorted(_rows(h), key=_key_order)


def _wide_sorted_records(rng, n_total, n_records, cards=(700, 3, 70_000), int_key=None, null_frac=0.02):
    """Rows over dictionary label columns of the given cardinalities (and, `int_key` = position, an int64 key column there), ordered by
    the columns in plan order — values ascending bytewise / numerically, NULLs last — cut into records at arbitrary rows. Dictionaries are
    in DESCENDING value order so that neither indices nor key ids happen to be ranks."""
    n_cols = len(cards)
    cols = []
    for c, k in enumerate(cards):
        v = rng.integers(0, k, n_total) if c != int_key else rng.integers(-k, k, n_total)
        if c == int_key:
            # (no key 0 next to NULL keys: the hash table — the other side of this comparison — files an int64 key 0 and a NULL key under one
            # fingerprint, as the reference's HashAggregate does, dynparquet/hashed.go:254-272; the run path keeps them apart like
            # ordered_aggregate.go's group ranges)
            v = np.where(v == 0, k, v)
        cols.append(np.where(rng.random(n_total) < null_frac, np.iinfo(np.int64).max, v))
    order = np.lexsort(tuple(reversed(cols)))
    cols = [c[order] for c in cols]
    val = rng.integers(-50, 1000, n_total).astype(np.int64)
    cuts = [0] + sorted(rng.integers(1, n_total, n_records - 1).tolist()) + [n_total]
    recs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        arrays, names = [], []
        for c, k in enumerate(cards):
            x = cols[c][a:b]
            null = x == np.iinfo(np.int64).max
            if c == int_key:
                arrays.append(pa.array(np.where(null, 0, x), mask=null)); names.append("bucket")
            else:
                d = pa.array([b"v%06d" % (k - 1 - i) for i in range(k)], type=pa.binary())
                arrays.append(pa.DictionaryArray.from_arrays(pa.array(np.where(null, 0, k - 1 - x).astype(np.uint32), mask=null), d)); names.append("labels.l%d" % c)
        arrays.append(pa.array(val[a:b])); names.append("v")
        recs.append(pa.RecordBatch.from_arrays(arrays, names=names))
    return recs


def _key_order_mixed(r, nkeys):
    return tuple((x is None, (x if x is not None else 0) if isinstance(x, (int, type(None))) and not isinstance(x, bytes) else x) for x in r[:nkeys])


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("agg_name", ["sum", "min", "count"])
@pytest.mark.parametrize("cards,expect", [((700, 3, 70_000), "fdb_hash_kernel(runs, wide)"), ((700, 3, 40_000), "fdb_hash_kernel(runs, medium)")])
def test_wide_run_records_dictionaries_beyond_one_byte_of_key_ids(pp, agg_name, resident, cards, expect):
    """Group columns of 700, 3 and 40 000 / 70 000 distinct values: medium run records (two bytes per key id, kept in registers) up to
    65 534 values, wide ones (the table's key tuple, from re-loaded columns) beyond — no hash kernel ran — equal the hash aggregate's
    groups in key order."""
    rng = np.random.default_rng(21)
    recs = _wide_sorted_records(rng, 400_000, 5, cards=cards)
    agg = {"sum": Sum(Col("v")), "min": Min(Col("v")), "count": Count(Col("v"))}[agg_name]
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    o, kernel = _run_plan(pp, recs, agg, groups, ordered=True, resident=resident)
    assert kernel == expect, kernel
    h, hk = _run_plan(pp, recs, agg, groups, ordered=False, resident=resident)
    assert "runs" not in hk
    orows, hrows = _rows(o), sorted(_rows(h), key=_key_order)
    assert len(orows) > 100_000 and orows == hrows


@pytest.mark.parametrize("pos", [0, 1])
def test_wide_run_records_int64_keys(pp, pos):
    """An int64 group key (a time bucket — the `window` vectors' shape) in front of / behind a dictionary column, with NULLs in both:
    runs hold the raw value, the order check compares values (NULL last); equal to the hash aggregate sorted by key."""
    rng = np.random.default_rng(22 + pos)
    cards = (5_000, 40) if pos == 0 else (40, 5_000)
    recs = _wide_sorted_records(rng, 300_000, 4, cards=cards, int_key=pos)
    groups = [Col("bucket"), Col("labels.l1")] if pos == 0 else [Col("labels.l0"), Col("bucket")]
    o, kernel = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True, resident=True)
    assert kernel == "fdb_hash_kernel(runs, wide)", kernel
    h, _ = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=False, resident=True)
    key = lambda r: tuple((x is None, x if x is not None else 0) for x in r[:2])  # noqa: E731
    orows = _rows(o)
    assert len(orows) > 50_000 and orows == sorted(_rows(h), key=key)


def test_wide_run_records_group_columns_that_come_and_go(pp):
    """Dynamic label columns that appear later and disappear again (a record that lacks a group column: id 0 = NULL there; a plan that
    gains a column: later segments have longer tuples) stay on the table-free path; equal to the hash aggregate."""
    rng = np.random.default_rng(23)
    n = 20_000
    def col(k):
        x = np.sort(rng.integers(0, k, n))
        return pa.DictionaryArray.from_arrays(pa.array(x.astype(np.uint32)), pa.array([b"k%04d" % i for i in range(k)], type=pa.binary()))
    v = lambda: pa.array(rng.integers(0, 100, n).astype(np.int64))  # noqa: E731
    recs = [pa.RecordBatch.from_arrays([col(300), v()], names=["labels.a", "v"]),
            pa.RecordBatch.from_arrays([col(300), col(7), v()], names=["labels.a", "labels.b", "v"]),
            pa.RecordBatch.from_arrays([col(5), v()], names=["labels.b", "v"]),
            pa.RecordBatch.from_arrays([col(300), col(7), col(2), v()], names=["labels.a", "labels.b", "labels.c", "v"])]
    o, kernel = _run_plan(pp, recs, Sum(Col("v")), [DynCol("labels")], ordered=True)
    assert kernel == "fdb_hash_kernel(runs, medium)", kernel  # (the last record carries every column again: ids in registers; the ones before it wrote wide records)
    h, _ = _run_plan(pp, recs, Sum(Col("v")), [DynCol("labels")], ordered=False)
    assert o.schema.names == h.schema.names[:-1] + ["v"]
    assert sorted(_rows(o), key=repr) == sorted(_rows(h), key=repr) and o.num_rows > 300


def test_wide_run_records_written_before_the_plan_gains_a_column_read_null_there(pp, monkeypatch):
    """Canonical wide records of a 3-column plan pad their 7-word tuple to 8 words; a record that adds a 4th group column puts it on
    exactly that word. The older runs must read NULL there (the padding is written as zeros), not whatever the LDS stage / the
    segment's block held (advisor finding, round 5). Several passes so that recycled blocks carry old tuples."""
    monkeypatch.setenv("FDB_RUNS_WIDE", "1")
    rng = np.random.default_rng(61)
    n = 30_000
    def col(k, tag):
        x = np.sort(rng.integers(0, k, n))
        return pa.DictionaryArray.from_arrays(pa.array(x.astype(np.uint32)), pa.array([b"%s%03d" % (tag, i) for i in range(k)], type=pa.binary()))
    v = lambda: pa.array(rng.integers(1, 100, n).astype(np.int64))  # noqa: E731
    for _ in range(3):
        three = [pa.RecordBatch.from_arrays([col(40, b"a"), col(9, b"b"), col(5, b"c"), v()], names=["labels.a", "labels.b", "labels.c", "v"]) for _ in range(2)]
        four = pa.RecordBatch.from_arrays([col(40, b"a"), col(9, b"b"), col(5, b"c"), col(6, b"d"), v()], names=["labels.a", "labels.b", "labels.c", "labels.d", "v"])
        recs = three + [four]
        # one launch per record (a launch over all three would know labels.d from the start and write the 3-column records non-canonically)
        plan = pp.HashAggregatePlan(None, [Sum(Col("v"))], [DynCol("labels")], ordered=True, final_stage=False)
        keep = [pp.ResidentBatch(r) for r in recs]
        try:
            kernels = []
            for k in keep:
                plan.CallbackResident([k])
                kernels.append(plan.last_kernel())
            o = plan.Finish()
        finally:
            plan.Close()
            for k in keep:
                k.close()
        assert kernels == ["fdb_hash_kernel(runs, wide)"] * 3, kernels
        h, _ = _run_plan(pp, recs, Sum(Col("v")), [DynCol("labels")], ordered=False, resident=True)
        assert sorted(_rows(o), key=repr) == sorted(_rows(h), key=repr) and o.num_rows > 100  # (columns sorted one by one: ≈ the sum of their cardinalities)
        # every group that came from the 3-column records has labels.d = NULL
        d = o.column(o.schema.get_field_index("labels.d"))
        assert d.null_count > 0


@pytest.mark.parametrize("force,name", [("1", "fdb_hash_kernel(runs, wide)"), ("m", "fdb_hash_kernel(runs, medium)")])
def test_every_table_free_test_shape_with_wide_records_forced(pp, monkeypatch, force, name):
    """FDB_RUNS_WIDE: the narrow-record shapes above through the wide-record / medium-record kernels — same answers."""
    monkeypatch.setenv("FDB_RUNS_WIDE", force)
    rng = np.random.default_rng(24)
    recs = _sorted_label_records(rng, 250_000, 5)
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    for agg in (Sum(Col("v")), Sum(Col("f")), Max(Col("f")), Count(Col("v"))):
        o, kernel = _run_plan(pp, recs, agg, groups, ordered=True, resident=True, filt=Col("v") > 100)
        assert kernel == name
        h, _ = _run_plan(pp, recs, agg, groups, ordered=False, resident=True, filt=Col("v") > 100)
        orows, hrows = _rows(o), sorted(_rows(h), key=_key_order)
        assert [r[:3] for r in orows] == [r[:3] for r in hrows]
        for a, b in zip(orows, hrows):
            assert a[3] == b[3] or (isinstance(a[3], float) and abs(a[3] - b[3]) <= 1e-9 * max(1.0, abs(b[3]))), (a, b)
    # every row its own run (more runs in a tile than a wave's stage holds: they go straight into the chunk), unsorted input (falls back)
    n = 70_000
    uniq = pa.RecordBatch.from_arrays(
        [pa.DictionaryArray.from_arrays(pa.array(np.arange(n, dtype=np.uint32)), pa.array([b"u%06d" % i for i in range(n)], type=pa.binary())),
         pa.array(rng.integers(0, 9, n).astype(np.int64))], names=["labels.u", "v"])
    for rec in (uniq, uniq.take(pa.array(rng.permutation(n)))):
        o, kernel = _run_plan(pp, [rec], Sum(Col("v")), [Col("labels.u")], ordered=True, resident=True)
        assert kernel == "fdb_hash_kernel(runs, wide)"  # (70 000 distinct values: beyond two bytes whatever is forced)
        h, _ = _run_plan(pp, [rec], Sum(Col("v")), [Col("labels.u")], ordered=False, resident=True)
        assert _rows(o) == sorted(_rows(h), key=lambda r: _key_order(r, 1)) and o.num_rows == n


def test_benchmark_schema_query_over_a_table_sorted_by_path(pp):
    """BASELINE.json config 2's query — labels.code == '200', SUM(value) GROUP BY labels.path (1 024 values: beyond the narrow record) —
    over bench.py's generator with the rows of every record sorted by labels.path: an ordered plan takes the table-free path and
    equals the hash aggregate; sums within 1e-9 (the fold order differs)."""
    from frostdb_amd import synth
    recs = []
    for i in range(3):
        b = synth.prometheus_chunk(3, i, 400_000, row_base=i * 400_000)
        path = b.column(b.schema.get_field_index("labels.path"))
        vals = np.array([v if v is not None else b"\xff" for v in path.dictionary.to_pylist()], dtype=object)
        idx = path.indices.to_numpy(zero_copy_only=False)
        keys = np.where(np.isnan(idx.astype(np.float64)), b"\xff\xff", vals[np.nan_to_num(idx.astype(np.float64)).astype(np.int64)])
        recs.append(b.take(pa.array(np.argsort(keys, kind="stable"))))
    # (each record is sorted; across records the keys restart, so Finish finds the order broken and merges through the table — still one
    # run kernel per record; a single sorted record stays table-free to the end)
    filt, aggs, groups = Col("labels.code") == "200", Sum(Col("value")), [Col("labels.path")]
    for seq in (recs[:1], recs):
        o, kernel = _run_plan(pp, seq, aggs, groups, ordered=True, resident=True, filt=filt)
        assert kernel == "fdb_hash_kernel(runs, medium)", kernel  # (1 024 path values: two bytes per key id; FDB_RUNS_ALWAYS — by default this key space keeps the dense table)
        h, _ = _run_plan(pp, seq, aggs, groups, ordered=False, resident=True, filt=filt)
        orows, hrows = _rows(o), sorted(_rows(h), key=lambda r: _key_order(r, 1))
        assert [r[0] for r in orows] == [r[0] for r in hrows] and len(orows) > 1000
        for a, b in zip(orows, hrows):
            assert abs(a[1] - b[1]) <= 1e-9 * abs(b[1]), (a, b)


@pytest.mark.parametrize("seed", range(8))
def test_run_path_against_the_hash_path_on_random_shapes(pp, seed):
    """Differential: an ordered plan (run kernel + run-store Finish, or its fall-backs) against the plain hash aggregate sorted by key, over
    shapes that stress the run store's bookkeeping — ragged records (1 row … tens of thousands, not multiples of the 256-row wave tile or
    the 1 024-row workgroup tile), every row its own run (a wave's LDS stage fills and flushes inside one record, chunks of 4 096 runs
    change), one run spanning whole records, NULL keys, a filter that drops most rows, more records than run segments (the path is left
    mid-scan), sorted and unsorted input."""
    rng = np.random.default_rng(1000 + seed)
    n_rec = int(rng.integers(1, 80 if seed % 4 == 3 else 12))
    card = [int(rng.integers(1, 200)), int(rng.integers(1, 9)), int(rng.integers(1, 4))]
    n_total = int(rng.integers(n_rec, 150_000))
    sort = seed % 3 != 2
    cols = []
    for k in card:
        v = rng.integers(0, k + 1, n_total)
        cols.append(np.where(rng.random(n_total) < 0.05, k, v))
    if seed % 4 == 1:  # every row its own key: the first column counts up
        card[0] = 250
        cols[0] = np.sort(rng.integers(0, 250, n_total))
        cols[1] = np.arange(n_total) % (card[1] + 1)
    if seed % 4 == 2:  # one key for (almost) everything
        cols = [np.zeros(n_total, dtype=np.int64) for _ in card]
        cols[2][-3:] = 1 if card[2] > 1 else 0
    if sort:
        order = np.lexsort(tuple(reversed(cols)))
        cols = [c[order] for c in cols]
    val = rng.integers(-1000, 1000, n_total).astype(np.int64)
    cuts = [0] + sorted(rng.integers(0, n_total + 1, n_rec - 1).tolist()) + [n_total]
    recs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == b and rng.random() < 0.5:
            continue
        arrays, names = [], []
        for c, k in enumerate(card):
            d = pa.array([b"k%03d" % i for i in range(k)], type=pa.binary())
            x = cols[c][a:b]
            arrays.append(pa.DictionaryArray.from_arrays(pa.array(np.where(x >= k, 0, x).astype(np.uint32), mask=x >= k), d)); names.append("labels.l%d" % c)
        arrays.append(pa.array(val[a:b])); names.append("v")
        recs.append(pa.RecordBatch.from_arrays(arrays, names=names))
    if not recs:
        return
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    filt = (Col("v") > 900) if seed % 2 == 1 else None
    agg = [Sum(Col("v")), Min(Col("v")), Max(Col("v")), Count(Col("v"))][seed % 4]
    for resident in (False, True):
        o, _ = _run_plan(pp, recs, agg, groups, ordered=True, resident=resident, filt=filt)
        h, _ = _run_plan(pp, recs, agg, groups, ordered=False, resident=resident, filt=filt)
        assert _rows(o) == sorted(_rows(h), key=_key_order), (seed, resident, n_rec, n_total)


def test_small_key_spaces_keep_the_dense_table(pp, monkeypatch):
    """Without FDB_RUNS_ALWAYS an ordered plan whose key space fits the LDS-resident dense table (here 6 × 8 × 4 slots; BASELINE.json
    config 2's 1 025 paths) scans with the dense kernel and sorts its groups at Finish — same record as the run path's, in key order."""
    monkeypatch.delenv("FDB_RUNS_ALWAYS", raising=False)
    rng = np.random.default_rng(31)
    recs = _sorted_label_records(rng, 200_000, 3)
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    o, kernel = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True, resident=True)
    assert "runs" not in kernel, kernel
    monkeypatch.setenv("FDB_RUNS_ALWAYS", "1")
    r, kernel = _run_plan(pp, recs, Sum(Col("v")), groups, ordered=True, resident=True)
    assert kernel == "fdb_hash_kernel(runs)"
    assert _rows(o) == _rows(r) and o.schema.names == r.schema.names


def test_wide_run_records_computed_int64_key(pp):
    """`(timestamp / 1000) * 1000 as bucket` — the `window` vectors' computed key — next to a label column in an ordered plan: the key exists
    only inside the kernel (projection), the run record is the wide one and its tuple's bucket words are recomputed by the lanes that end a
    run. The plan's key order is (labels.x, bucket) — stored columns first — so rows sorted by (bucket, label) BREAK it and the runs go through
    the table (int64 valid bits must survive that road: they did not before this test), rows sorted by (label, bucket) stay table-free; both
    equal the hash aggregate."""
    rng = np.random.default_rng(41)
    n = 300_000
    ts = np.sort(rng.integers(1000, 4_000_000, n)).astype(np.int64)  # (no bucket 0: an int64 key 0 and a NULL key are one group in the hash table, see _wide_sorted_records)
    lab = rng.integers(0, 6, n)
    bucket = (Col("timestamp") / 1000 * 1000).Alias("bucket")
    vals = rng.integers(0, 1000, n).astype(np.int64)
    for keys in ("bucket_first", "label_first"):
      order = np.lexsort((lab, ts // 1000)) if keys == "bucket_first" else np.lexsort((ts // 1000, lab))
      ts2, lab2 = ts[order], lab[order]
      labels = pa.DictionaryArray.from_arrays(pa.array(np.where(lab2 == 5, 0, lab2).astype(np.uint32), mask=lab2 == 5), pa.array([b"a", b"b", b"c", b"d", b"e"], type=pa.binary()))
      rec = pa.RecordBatch.from_arrays([pa.array(ts2), labels, pa.array(vals)], names=["timestamp", "labels.x", "v"])
      res = {}
      for ordered in (True, False):
          plan = pp.HashAggregatePlan(None, [Sum(Col("v"))], [bucket, Col("labels.x")], ordered=ordered, final_stage=False)
          rbs = [pp.ResidentBatch(rec.slice(0, 100_001)), pp.ResidentBatch(rec.slice(100_001))]
          try:
              plan.CallbackResident(rbs)
              kernel = plan.last_kernel()
              out = plan.Finish()
              names = out.schema.names
              ib, il = names.index("bucket"), names.index("labels.x")
              iv = [i for i in range(3) if i not in (ib, il)][0]
              res[ordered] = [(r[ib], r[il], r[iv]) for r in _rows(out)]  # (bucket, label, sum) whatever order the columns come back in
          finally:
              plan.Close()
              for r in rbs:
                  r.close()
          if ordered:
              assert kernel == "fdb_hash_kernel(runs, wide)", kernel
      key = lambda r: (r[1] is None, r[1] or b"", r[0])  # noqa: E731  (the plan's key order: labels.x, then bucket)
      assert len(res[True]) > 10_000 and res[True] == sorted(res[False], key=key), keys


def test_deterministic_ordered_plans_do_not_collect_runs(pp):
    """fdb_plan_set_deterministic on an ordered plan: groups cut by wave or record boundaries are folded with atomics at the
    run store's Finish, so such a plan keeps the dense kernel (whose float sums ARE reproducible) — and two passes give the same bits."""
    rng = np.random.default_rng(42)
    recs = _sorted_label_records(rng, 200_000, 3)
    groups = [Col("labels.l0"), Col("labels.l1"), Col("labels.l2")]
    outs = []
    for _ in range(2):
        plan = pp.HashAggregatePlan(None, [Sum(Col("f"))], groups, ordered=True, final_stage=False)
        plan.set_deterministic(True)
        keep = [pp.ResidentBatch(r) for r in recs]
        try:
            plan.CallbackResident(keep)
            assert "runs" not in plan.last_kernel(), plan.last_kernel()
            out = plan.Finish()
        finally:
            plan.Close()
            for k in keep:
                k.close()
        outs.append(out)
    assert outs[0].column(3).to_numpy().tobytes() == outs[1].column(3).to_numpy().tobytes()
    h, _ = _run_plan(pp, recs, Sum(Col("f")), groups, ordered=False, resident=True)
    a, b = _rows(outs[0]), sorted(_rows(h), key=_key_order)
    assert [r[:3] for r in a] == [r[:3] for r in b]
    assert all(abs(x[3] - y[3]) <= 1e-9 * max(1.0, abs(y[3])) for x, y in zip(a, b))
