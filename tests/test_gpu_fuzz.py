"""Differential fuzzing of the device path against the oracle: random plans (filter trees over dictionary / numeric / NULL /
missing columns, 0-3 group columns incl. dynamic sets, int64 and computed keys, 0-5 aggregations incl. computed inputs) over
random records with NULLs, schema drift and different dictionaries per record. Fixed seeds — every case is reproducible by its id."""
import math

import numpy as np
import pyarrow as pa
import pytest

from frostdb_amd.logicalplan import OP_LT_EQ, And, AndAgg, BinaryExpr, Col, Count, DynCol, Max, Min, Or, Sum, Unique
from tests.util import arrow_to_pydict, batch_rows, sort_key

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import physicalplan
    assert physicalplan.device_count() >= 1
    return physicalplan


def random_batch(rng, n, drop=()):
    def darr(card, nf, prefix):
        names = [b"%s%03d" % (prefix, i) for i in range(card)]
        perm = rng.permutation(card)  # every record has its own dictionary ORDER (and may miss values)
        keep = perm[: max(1, int(card * rng.uniform(0.6, 1.0)))]
        idx = rng.integers(0, len(keep), size=n).astype(np.uint32)
        mask = rng.random(n) < nf if nf > 0 else None
        return pa.DictionaryArray.from_arrays(pa.array(idx, type=pa.uint32(), mask=mask), pa.array([names[k] for k in keep], type=pa.binary()))

    cols = {
        "labels.a": darr(5, 0.05, b"a"),
        "labels.b": darr(40, 0.0, b"b"),
        "labels.c": darr(300, 0.2, b"c"),
        "labels.d": darr(3, 0.5, b"d"),
        "ts": pa.array((1000 + rng.integers(0, 50, size=n) * 10).astype(np.int64)),
        "ival": pa.array(rng.integers(-20, 20, size=n).astype(np.int64), mask=rng.random(n) < 0.1),
        "fval": pa.array(rng.uniform(-5, 5, size=n), mask=rng.random(n) < 0.1),
        "small": pa.array(rng.integers(0, 4, size=n).astype(np.int64)),
        "flag": pa.array(rng.random(n) < 0.9, mask=rng.random(n) < 0.15),
    }
    for d in drop:
        cols.pop(d, None)
    return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))


def random_leaf(rng):
    k = rng.integers(0, 9)
    if k == 0:
        return Col("labels.a") == ("a%03d" % rng.integers(0, 6))
    if k == 1:
        return Col("labels.b") != ("b%03d" % rng.integers(0, 45))
    if k == 2:
        return Col("labels.c").RegexMatch("c0[0-%d]." % rng.integers(1, 9))
    if k == 3:
        return Col("labels.d") == None if rng.random() < 0.5 else Col("labels.d") != None  # noqa: E711
    if k == 4:
        return [Col("ts") >= 1200, Col("ts") < 1300, Col("ts") == 1250, Col("ts") != 1010][rng.integers(0, 4)]
    if k == 5:
        return [Col("fval") > 0.5, Col("fval") <= -1.0, Col("ival") >= 3, Col("ival") < -2.5][rng.integers(0, 4)]
    if k == 6:
        return Col("labels.c").Contains("c1")
    if k == 7:
        return Col("labels.missing") == "x" if rng.random() < 0.5 else Col("labels.missing") != "x"
    return Col("labels.b").NotContains("b01")


def random_filter(rng, depth=0):
    r = rng.random()
    if depth >= 3 or r < 0.4:
        return random_leaf(rng)
    if r < 0.7:
        return And(random_filter(rng, depth + 1), random_filter(rng, depth + 1))
    return Or(random_filter(rng, depth + 1), random_filter(rng, depth + 1))


def random_plan(rng):
    filt = random_filter(rng) if rng.random() < 0.8 else None
    I, F, T = Col("ival"), Col("fval"), Col("ts")
    agg_pool = [Sum(I), Min(I), Max(I), Count(I), Sum(F), Min(F), Max(F), Count(F), Sum(T), Max(T),
                Sum(I * T), Min(I - T), Max(T / Col("small")), Sum(F * 2.0), Min(F / F), Unique(Col("small")), Unique(I), AndAgg(Col("flag"))]
    n_aggs = int(rng.integers(0, 6))
    aggs = [agg_pool[i] for i in rng.choice(len(agg_pool), size=n_aggs, replace=False)]
    group_pool = [[], [Col("labels.a")], [Col("labels.d")], [Col("labels.a"), Col("labels.d")], [Col("labels.b"), Col("labels.a")],
                  [Col("labels.c")], [DynCol("labels")], [Col("ts")], [Col("labels.a"), Col("ts")], [(T / 100 * 100).Alias("bucket")],
                  [Col("labels.b"), (Col("small") - 1).Alias("sm1")], [Col("labels.c"), Col("labels.b")]]
    groups = group_pool[int(rng.integers(0, len(group_pool)))]
    if not aggs and not groups:
        aggs = [Count(I)]
    return filt, aggs, groups


def canon(d, key_cols, computed_keys):
    for k in computed_keys:  # int64 key 0 ≡ NULL (reference hash identity): fold for comparison
        if k in d:
            d[k] = [0 if v is None else v for v in d[k]]
    return sorted(batch_rows(d, key_cols + [c for c in d if c not in key_cols]), key=lambda r: sort_key(r[:len(key_cols)]))


@pytest.mark.parametrize("seed", range(96))
def test_fuzz_plan_vs_oracle(pp, seed, monkeypatch):
    rng = np.random.default_rng(10_000 + seed)
    if seed % 4 == 3:
        monkeypatch.setenv("FDB_NO_JIT", "1")
    filt, aggs, groups = random_plan(rng)
    if seed % 4 == 3 and any(not hasattr(a.expr, "dynamic") or a.expr.__class__.__name__ != "Column" for a in aggs):
        aggs = [a for a in aggs if a.expr.__class__.__name__ == "Column"]  # the interpreting kernels do not evaluate expressions
    if seed % 4 == 3:
        groups = [g for g in groups if g.__class__.__name__ == "Column"]
        if not aggs and not groups:
            aggs = [Count(Col("ival"))]
    n_rec = int(rng.integers(1, 5))
    recs = []
    for r in range(n_rec):
        drop = [c for c in ("labels.c", "labels.d") if rng.random() < 0.15]
        recs.append(random_batch(rng, int(rng.integers(1, 30_000)), drop=drop))
    check_against_oracle(pp, filt, aggs, groups, recs, resident=seed % 2 == 0)


def check_against_oracle(pp, filt, aggs, groups, recs, resident):
    from oracle import OraclePlan
    o = OraclePlan(filt, aggs, groups, nchains=1)
    oracle_err = None
    try:
        for r in recs:
            o.push(r)
        want = o.finish().to_pydict()
    except Exception as e:  # noqa: BLE001
        oracle_err = e
    finally:
        o.close()
    plan = pp.HashAggregatePlan(filt, aggs, groups)
    try:
        if oracle_err is not None:
            with pytest.raises(pp.FdbError):
                for r in recs:
                    plan.Callback(r)
                plan.Finish()
            return
        keep = []
        for r in recs:
            if resident:
                keep.append(pp.ResidentBatch(r))
                plan.Callback(keep[-1])
            else:
                plan.Callback(r)
        got = arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
    if not want or all(len(v) == 0 for v in want.values()):  # nothing selected: the reference emits no record at all (aggregate.go:547-549)
        assert all(len(v) == 0 for v in got.values()), (got, str(filt))
        return
    agg_names = [a.Name() for a in aggs]
    key_cols = [c for c in want if c not in agg_names]
    assert sorted(got.keys()) == sorted(want.keys()), (got.keys(), want.keys())
    computed = [g.name for g in groups if g.__class__.__name__ == "AliasExpr"] + [g.name for g in groups if g.name in ("ts", "small", "u")]
    a, b = canon(got, key_cols, computed), canon(want, key_cols, computed)
    assert len(a) == len(b), (len(a), len(b), str(filt), agg_names, [g.name for g in groups])
    cols = key_cols + [c for c in got if c not in key_cols]
    cols_w = key_cols + [c for c in want if c not in key_cols]
    for ra, rb in zip(a, b):
        da, db = dict(zip(cols, ra)), dict(zip(cols_w, rb))
        for c in cols:
            x, y = da[c], db[c]
            if isinstance(x, float) or isinstance(y, float):
                assert (x is None and y is None) or math.isclose(x, y, rel_tol=REL_TOL, abs_tol=1e-12), (c, x, y, str(filt))
            else:
                assert x == y, (c, x, y, str(filt), agg_names)


# ---- second family: plain string / binary columns, boolean columns and boolean projections ---------------------------------

WORDS = ["", "a", "ab", "abc", "b", "ba", "zeta", "Zeta", "é", "value1", "value10", "value2"] + ["w%02d" % k for k in range(30)]


def random_batch2(rng, n, drop=()):
    def sarr(typ, nf, lo, hi):
        pick = rng.integers(lo, hi, size=n)
        return pa.array([WORDS[k] for k in pick], type=pa.string(), mask=rng.random(n) < nf).cast(typ)

    lo = int(rng.integers(0, 10))
    cols = {
        "s.name": sarr([pa.string(), pa.large_string()][int(rng.integers(0, 2))], 0.1, lo, len(WORDS)),
        "s.raw": sarr(pa.binary(), 0.0, 0, 8),
        "labels.a": pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 4, size=n).astype(np.uint32), mask=rng.random(n) < 0.05),
                                                   pa.array([b"a0", b"a1", b"a2", b"a3"], type=pa.binary())),
        "ts": pa.array((1000 + rng.integers(0, 30, size=n) * 10).astype(np.int64)),
        "ival": pa.array(rng.integers(-20, 20, size=n).astype(np.int64), mask=rng.random(n) < 0.1),
        "fval": pa.array(rng.uniform(-5, 5, size=n), mask=rng.random(n) < 0.1),
        "flag": pa.array(rng.random(n) < 0.6, mask=rng.random(n) < 0.15),
        "u": pa.array(rng.integers(0, 6, size=n).astype(np.uint64), mask=rng.random(n) < 0.1),
    }
    for d in drop:
        cols.pop(d, None)
    return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))


def random_leaf2(rng):
    k = int(rng.integers(0, 10))
    w = WORDS[int(rng.integers(0, len(WORDS)))]
    N = Col("s.name")
    if k == 0:
        return [N == w, N != w, N < w, N <= w, N > w, N >= w][int(rng.integers(0, 6))]
    if k == 1:
        return N.RegexMatch("^w[0-%d]" % rng.integers(0, 3)) if rng.random() < 0.5 else N.RegexNotMatch("a")
    if k == 2:
        return N.Contains(w) if rng.random() < 0.5 else N.NotContains("a")
    if k == 3:
        return [N == None, N != None, N.Contains(None)][int(rng.integers(0, 3))]  # noqa: E711
    if k == 4:
        return [Col("s.raw") == b"ab", Col("s.raw") > b"a", Col("s.raw").RegexMatch("^b")][int(rng.integers(0, 3))]
    if k == 5:
        return [Col("flag") == True, Col("flag") != True, Col("flag") == False, Col("flag") > False][int(rng.integers(0, 4))]  # noqa: E712
    if k == 6:
        return Col("s.missing") == "x" if rng.random() < 0.5 else Col("s.missing") < "x"
    if k == 7:
        return Col("labels.a") == "a1"
    if k == 8:
        return Col("ts") >= 1100
    return Col("fval") > 0.0


def random_filter2(rng, depth=0):
    r = rng.random()
    if depth >= 3 or r < 0.4:
        return random_leaf2(rng)
    if r < 0.7:
        return And(random_filter2(rng, depth + 1), random_filter2(rng, depth + 1))
    return Or(random_filter2(rng, depth + 1), random_filter2(rng, depth + 1))


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_plain_strings_and_bools_vs_oracle(pp, seed):
    rng = np.random.default_rng(20_000 + seed)
    filt = random_filter2(rng) if rng.random() < 0.8 else None
    I, F, T = Col("ival"), Col("fval"), Col("ts")
    agg_pool = [Sum(I), Min(I), Max(F), Count(I), Sum(F), Sum(I * T), AndAgg(Col("flag")), Unique(I)]
    aggs = [agg_pool[i] for i in rng.choice(len(agg_pool), size=int(rng.integers(0, 4)), replace=False)]
    group_pool = [[Col("s.name")], [Col("s.raw")], [DynCol("s")], [Col("s.name"), Col("labels.a")], [DynCol("s"), Col("ts")],
                  [Col("flag")], [Col("flag"), Col("s.name")], [Col("u")], [Col("u"), Col("labels.a")],
                  [Col("labels.a"), T > 1100], [Col("s.name"), And(I > 0, F < 1.5)], [Or(T == 1000, BinaryExpr(I, OP_LT_EQ, F))], [Col("s.raw"), (T / 100).Alias("h")]]
    groups = group_pool[int(rng.integers(0, len(group_pool)))]
    recs = []
    for r in range(int(rng.integers(1, 4))):
        drop = [c for c in ("s.raw",) if rng.random() < 0.15]
        recs.append(random_batch2(rng, int(rng.integers(1, 20_000)), drop=drop))
    check_against_oracle(pp, filt, aggs, groups, recs, resident=seed % 2 == 0)
