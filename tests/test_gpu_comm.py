"""The cross-GPU merge behind the C ABI (fdb_comm_*, fdb_plan_allreduce, fdb_plan_exchange) on a 1-GPU box.

RCCL refuses two ranks of one communicator on the same device (probed: ncclCommInitRank → "invalid usage"), so the
multi-rank behaviour is driven through the in-process peer-to-peer transport (fdb_comm_init_local: N ranks = N threads, here
all on device 0) — the plan-level code (layout probe, grouped in-place all-reduce, schema agreement, re-key + partition,
exchange, import) is the same for both transports — and the RCCL transport itself runs with one rank (ncclCommInitRank /
ncclCommInitAll, every collective on a 1-rank communicator). No torch, no gloo: everything goes through ctypes.
"""
import math
import threading

import numpy as np
import pyarrow as pa
import pytest

from frostdb_amd.logicalplan import And, Col, Count, DynCol, Max, Min, Or, Sum
from tests.test_gpu_parity import CFG2, CFG3, assert_same_result, run_oracle
from tests.util import arrow_to_pydict, dict_array, make_prometheus_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import physicalplan
    assert physicalplan.device_count() >= 1, "no HIP device visible"
    return physicalplan


@pytest.fixture(scope="module")
def fcomm():
    from frostdb_amd import comm
    return comm


def run_ranks(n, fn):
    """fn(rank) on n threads (one per rank, like one goroutine per chain); re-raises the first failure."""
    out, errs = [None] * n, []

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective"
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_local_ranks_aligned_allreduce_in_place(pp, fcomm, world, cfg):
    """Parts of one table (same dictionaries) on `world` ranks: layouts agree, the table arrays are all-reduced in place on each
    plan's own stream, and EVERY rank then holds the merged table: each one's Finish equals the oracle over all shards."""
    q = CFG2 if cfg == "cfg2" else CFG3
    rng = np.random.default_rng(5 + world)
    shards = [[make_prometheus_batch(rng, 20_000 + 1_000 * r, n_path=40, null_frac=0.0)] for r in range(world)]
    want = run_oracle([b for s in shards for b in s], **q)
    cols = ["labels.path"] + [a.Name() for a in q["aggs"]]
    comms = fcomm.Comm.init_local([0] * world)
    assert [c.rank for c in comms] == list(range(world)) and all(c.size == world for c in comms)

    def rank_fn(r):
        plan = pp.HashAggregatePlan(q["filter_expr"], q["aggs"], q["groups"])
        keep = [pp.ResidentBatch(b) for b in shards[r]]
        try:
            plan.CallbackResident(keep)
            assert comms[r].allreduce(plan) is True
            return arrow_to_pydict(plan.Finish())
        finally:
            plan.Close()
            for k in keep:
                k.close()

    for got in run_ranks(world, rank_fn):
        assert_same_result(got, want, cols, float_cols={"sum(value)"})
    for c in comms:
        c.close()


def drifting_shard(rng, r, n=15_000):
    """Rank r's part: its own dictionary order, some values only it has, and one extra label column on odd ranks."""
    paths = [b"/p/%03d" % i for i in range(30 + 7 * r)]
    rng.shuffle(paths)
    code = [b"200", b"500", b"404"][:: 1 if r % 2 == 0 else -1]
    cols = {
        "labels.code": pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 3, n).astype(np.uint32)), pa.array(code, type=pa.binary())),
        "labels.path": pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, len(paths), n).astype(np.uint32), mask=rng.random(n) < 0.02),
                                                      pa.array(paths, type=pa.binary())),
    }
    if r % 2 == 1:
        cols["labels.zone"] = dict_array([None if rng.random() < 0.3 else b"z%d" % rng.integers(0, 3) for _ in range(n)])
    cols["timestamp"] = pa.array(rng.integers(0, 10**6, n).astype(np.int64))
    cols["value"] = pa.array(rng.uniform(0, 100, n))
    return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))


@pytest.mark.parametrize("world", [2, 3])
def test_local_ranks_unaligned_layouts_take_the_exchange(pp, fcomm, world):
    """Different dictionaries and column sets per rank: allreduce reports 'not aligned' and changes nothing; the exchange gives
    every rank a disjoint shard and the union of the shards equals the oracle over all parts."""
    rng = np.random.default_rng(77)
    shards = [drifting_shard(rng, r) for r in range(world)]
    filt = Or(Col("labels.code") == "200", Col("labels.code") == "500")
    aggs = [Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp")), Sum(Col("value"))]
    groups = [DynCol("labels")]
    want = run_oracle(shards, filt, aggs, groups)
    comms = fcomm.Comm.init_local([0] * world)

    def rank_fn(r):
        plan = pp.HashAggregatePlan(filt, aggs, groups)
        try:
            plan.Callback(shards[r])
            assert comms[r].allreduce(plan) is False
            shard = comms[r].merge_alltoall(plan)
            try:
                return arrow_to_pydict(shard.Finish())
            finally:
                shard.Close()
        finally:
            plan.Close()

    parts = run_ranks(world, rank_fn)
    cols = sorted({c for p in parts for c in p if c.startswith("labels.")}) + [a.Name() for a in aggs]
    merged = {c: [] for c in cols}
    for p in parts:
        n = len(next(iter(p.values()))) if p else 0
        for c in cols:
            merged[c] += p.get(c, [None] * n)
    assert sum(len(next(iter(p.values()))) if p else 0 for p in parts) == len(want["count(value)"])  # shards are disjoint
    assert_same_result(merged, want, cols, float_cols={"sum(value)"})
    for c in comms:
        c.close()


def test_exchange_adds_the_ranks_partial_sums_in_rank_order(pp, fcomm):
    """The hash-partitioned exchange merges what an owner received RANK BY RANK: a group's float64 partial sums are added in rank order,
    so the merged bits are a function of the ranks' partial sums alone — here every rank holds every group exactly once
    (its partial sum IS its row's value), values spread over 12 orders of magnitude so that (a + b) + c ≠ a + (b + c) nearly everywhere;
    the result must equal the left-to-right float64 sum over the ranks bit for bit, on several passes."""
    world, n_groups = 4, 30_000
    rng = np.random.default_rng(8)
    vals = [rng.uniform(-1, 1, n_groups) * 10.0 ** rng.integers(-6, 7, n_groups) for _ in range(world)]

    def shard(r):
        order = rng.permutation(n_groups)  # (every rank meets the groups in its own order)
        g = np.arange(n_groups)[order]
        arrays, names = [], []
        for c in range(10):
            digit = ((g >> (2 * c)) & 3).astype(np.uint32)
            arrays.append(pa.DictionaryArray.from_arrays(pa.array(digit), pa.array([b"c%02d=%d" % (c, k) for k in range(4)], type=pa.binary())))
            names.append("labels.l%02d" % c)
        arrays.append(pa.array(vals[r][order]))
        names.append("value")
        return pa.RecordBatch.from_arrays(arrays, names=names)

    shards = [shard(r) for r in range(world)]
    want = vals[0].copy()
    for r in range(1, world):
        want = want + vals[r]  # left to right: ((v0 + v1) + v2) + v3
    aggs, groups = [Sum(Col("value"))], [DynCol("labels")]
    for _ in range(3):
        comms = fcomm.Comm.init_local([0] * world)

        def rank_fn(r):
            plan = pp.HashAggregatePlan(None, aggs, groups)
            rb = pp.ResidentBatch(shards[r])
            try:
                plan.Callback(rb)
                return comms[r].merge(plan)
            finally:
                plan.Close()
                rb.close()

        parts = run_ranks(world, rank_fn)
        got = np.full(n_groups, np.nan)
        for out in parts:
            gid = np.zeros(out.num_rows, dtype=np.int64)
            for c in range(10):
                col = out.column(out.schema.get_field_index("labels.l%02d" % c))
                digit = np.array([int(v.rsplit(b"=", 1)[1]) for v in col.dictionary.to_pylist()], dtype=np.int64)[col.indices.to_numpy()]
                gid |= digit << (2 * c)
            got[gid] = out.column(out.schema.get_field_index("sum(value)")).to_numpy()
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
        for c in comms:
            c.close()


def test_local_ranks_high_cardinality_exchange(pp, fcomm):
    """Hash-mode tables (12 label columns, ≈60 k groups) on 4 ranks: shards are disjoint, their union is the oracle's result."""
    world, n_cols, n = 4, 12, 60_000
    rng = np.random.default_rng(3)

    def shard(r):
        g = rng.integers(0, 40_000, n)
        arrays, names = [], []
        for c in range(n_cols):
            digit = ((g >> (2 * c)) & 3).astype(np.uint32) if c < 8 else ((g * (c + 3)) % 5).astype(np.uint32)
            vals = [b"c%02d=%d" % (c, k) for k in range(5)]
            arrays.append(pa.DictionaryArray.from_arrays(pa.array(digit), pa.array(vals, type=pa.binary())))
            names.append("labels.l%02d" % c)
        arrays.append(pa.array(rng.uniform(0, 10, n)))
        names.append("value")
        return pa.RecordBatch.from_arrays(arrays, names=names)

    shards = [shard(r) for r in range(world)]
    aggs, groups = [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")]
    want = run_oracle(shards, None, aggs, groups)
    comms = fcomm.Comm.init_local([0] * world)

    def rank_fn(r):
        plan = pp.HashAggregatePlan(None, aggs, groups)
        rb = pp.ResidentBatch(shards[r])
        try:
            plan.Callback(rb)
            out = comms[r].merge(plan)  # hash mode is never "aligned": merge() falls through to the exchange
            return arrow_to_pydict(out)
        finally:
            plan.Close()
            rb.close()

    parts = run_ranks(world, rank_fn)
    cols = ["labels.l%02d" % c for c in range(n_cols)] + [a.Name() for a in aggs]
    merged = {c: sum((p[c] for p in parts), []) for c in cols}
    assert len(merged["count(value)"]) == len(want["count(value)"])
    assert_same_result(merged, want, cols, float_cols={"sum(value)"})
    for c in comms:
        c.close()


@pytest.mark.parametrize("world", [2, 4])
def test_ordered_plans_on_several_ranks_exchange_and_finish_their_shards_in_key_order(pp, fcomm, world, monkeypatch):
    """Ordered plans (fdb_plan_desc.ordered) across ranks: every rank scans its part of a table sorted by (label, bucket) into a run store,
    the exchange re-keys and partitions the runs through the hash table, and every rank finishes ITS shard of the groups in key order —
    sorted on the device (round 5: the ordered Finish out of the table). Shards are disjoint, their union is the hash aggregate's result."""
    monkeypatch.setenv("FDB_RUNS_ALWAYS", "1")
    rng = np.random.default_rng(17)
    n = 240_000
    lab = rng.integers(0, 301, n)           # 300 = NULL
    bucket = rng.integers(-2_000, 2_000, n)
    bucket = np.where(bucket == 0, 2_000, bucket)  # (no int64 key 0 next to NULL keys: the hash table files them under one fingerprint)
    bnull = rng.random(n) < 0.02
    order = np.lexsort((np.where(bnull, np.iinfo(np.int64).max, bucket), lab))
    lab, bucket, bnull = lab[order], bucket[order], bnull[order]
    d = pa.array([b"v%03d" % (299 - i) for i in range(300)], type=pa.binary())  # (descending: ids are not ranks)
    rec = pa.RecordBatch.from_arrays(
        [pa.DictionaryArray.from_arrays(pa.array(np.where(lab == 300, 0, 299 - lab).astype(np.uint32), mask=lab == 300), d),
         pa.array(np.where(bnull, 0, bucket).astype(np.int64), mask=bnull), pa.array(rng.integers(1, 100, n).astype(np.int64))], names=["labels.x", "bucket", "v"])
    per = n // world
    shards = [rec.slice(r * per, per if r + 1 < world else n - r * per) for r in range(world)]
    aggs, groups = [Sum(Col("v"))], [Col("labels.x"), Col("bucket")]
    comms = fcomm.Comm.init_local([0] * world)

    def rank_fn(r):
        plan = pp.HashAggregatePlan(None, aggs, groups, ordered=True)
        try:
            plan.Callback(shards[r])
            assert "runs" in plan.last_kernel(), plan.last_kernel()
            shard = comms[r].merge_alltoall(plan)
            try:
                out = shard.Finish()
                ran = shard.last_kernel()
            finally:
                shard.Close()
            return out, ran
        finally:
            plan.Close()

    parts = run_ranks(world, rank_fn)
    key = lambda r: (r[0] is None, r[0] or b"", r[1] is None, r[1] if r[1] is not None else 0)  # noqa: E731
    rows = []
    for out, ran in parts:
        assert ran == "hash_gather_rows_kernel + runs_sort_keys_kernel", ran
        cols = [(c.dictionary_decode() if pa.types.is_dictionary(c.type) else c).to_pylist() for c in out.columns]
        mine = list(zip(*cols))
        assert len(mine) > 20_000 and mine == sorted(mine, key=key)  # this rank's shard, in key order
        rows += mine
    h = pp.HashAggregatePlan(None, aggs, groups)
    try:
        h.Callback(rec)
        want = h.Finish()
    finally:
        h.Close()
    wcols = [(c.dictionary_decode() if pa.types.is_dictionary(c.type) else c).to_pylist() for c in want.columns]
    assert sorted(rows, key=key) == sorted(zip(*wcols), key=key)
    for c in comms:
        c.close()


def test_rccl_transport_single_rank_through_the_c_abi(pp, fcomm):
    """RCCL bound inside the library: unique id → ncclCommInitRank (1 rank) → layout probe + grouped in-place all-reduce on the
    plan's stream, and the send/recv exchange, each reproducing the plain Finish; then the same through ncclCommInitAll."""
    rng = np.random.default_rng(31)
    batches = [make_prometheus_batch(rng, 40_000, n_path=50), make_prometheus_batch(rng, 30_000, n_path=80)]
    cols = ["labels.path"] + [a.Name() for a in CFG3["aggs"]]
    want = run_oracle(batches, **CFG3)
    uid = fcomm.unique_id()
    assert len(uid) == fcomm.UNIQUE_ID_BYTES
    for make in (lambda: fcomm.Comm(uid, 1, 0, 0), lambda: fcomm.Comm.init_all([0])[0]):
        c = make()
        assert (c.rank, c.size) == (0, 1)
        p1 = pp.HashAggregatePlan(**{"filter_expr": CFG3["filter_expr"], "aggs": CFG3["aggs"], "groups": CFG3["groups"]})
        for b in batches:
            p1.Callback(b)
        assert c.allreduce(p1) is True
        assert_same_result(arrow_to_pydict(p1.Finish()), want, cols, float_cols={"sum(value)"})
        p1.Close()
        p2 = pp.HashAggregatePlan(**{"filter_expr": CFG3["filter_expr"], "aggs": CFG3["aggs"], "groups": CFG3["groups"]})
        for b in batches:
            p2.Callback(b)
        shard = c.merge_alltoall(p2)
        assert_same_result(arrow_to_pydict(shard.Finish()), want, cols, float_cols={"sum(value)"})
        shard.Close()
        p2.Close()
        c.close()
        uid = fcomm.unique_id()


def test_empty_rank(pp, fcomm):
    """A rank whose shard is empty has no table: the layout probe says 'not aligned' on EVERY rank (nobody's table is touched) and
    the exchange still merges."""
    rng = np.random.default_rng(9)
    rec = make_prometheus_batch(rng, 9_000, n_path=20)
    want = run_oracle([rec], **CFG2)
    comms = fcomm.Comm.init_local([0, 0])

    def rank_fn(r):
        plan = pp.HashAggregatePlan(CFG2["filter_expr"], CFG2["aggs"], CFG2["groups"])
        try:
            if r == 0:
                plan.Callback(rec)
            assert comms[r].allreduce(plan) is False
            out = comms[r].merge(plan)
            return arrow_to_pydict(out)
        finally:
            plan.Close()

    parts = run_ranks(2, rank_fn)
    cols = ["labels.path", "sum(value)"]
    merged = {c: sum((p.get(c, []) for p in parts), []) for c in cols}
    assert_same_result(merged, want, cols, float_cols={"sum(value)"})
    for c in comms:
        c.close()


@pytest.mark.parametrize("how", ["allreduce", "exchange"])
def test_a_rank_that_fails_before_the_merge_takes_every_rank_out_with_an_error(pp, fcomm, how, monkeypatch):
    """A rank that raises BEFORE a merge's first collective (a pending record that fails when it is scanned, a wrong device) must
    not leave its peers blocked inside the collective: the failure is voted through that first collective — the layout probe of
    fdb_plan_allreduce, the schema all-gather of fdb_plan_exchange — and every rank returns an error. Rank 1 fails here through
    the library's test hook; ranks 0 and 2 must come back (run_ranks asserts nobody is stuck) with FDB_ERR_STATE."""
    world = 3
    rng = np.random.default_rng(11)
    shards = [make_prometheus_batch(rng, 5_000, n_path=20, null_frac=0.0) for _ in range(world)]
    comms = fcomm.Comm.init_local([0] * world)
    monkeypatch.setenv("FDB_TEST_FAIL_MERGE_RANK", "1")
    errors = [None] * world

    def rank_fn(r):
        plan = pp.HashAggregatePlan(CFG2["filter_expr"], CFG2["aggs"], CFG2["groups"])
        try:
            plan.Callback(shards[r])
            try:
                if how == "allreduce":
                    comms[r].allreduce(plan)
                else:
                    comms[r].merge_alltoall(plan).Close()
            except pp.FdbError as e:
                errors[r] = e
        finally:
            plan.Close()

    run_ranks(world, rank_fn)
    assert all(e is not None for e in errors), errors
    assert "injected failure" in str(errors[1])
    for r in (0, 2):
        assert errors[r].code == pp.FDB_ERR_STATE and "another rank failed" in str(errors[r]), str(errors[r])
    monkeypatch.delenv("FDB_TEST_FAIL_MERGE_RANK")
    # the communicator is still usable afterwards (nobody is half way through a collective)

    def again(r):
        plan = pp.HashAggregatePlan(CFG2["filter_expr"], CFG2["aggs"], CFG2["groups"])
        try:
            plan.Callback(shards[r])
            assert comms[r].allreduce(plan) is True
            return arrow_to_pydict(plan.Finish())
        finally:
            plan.Close()

    want = run_oracle(shards, **CFG2)
    for got in run_ranks(world, again):
        assert_same_result(got, want, ["labels.path", "sum(value)"], float_cols={"sum(value)"})
    for c in comms:
        c.close()


@pytest.mark.timeout(900)
def test_bench_two_ranks_over_the_in_process_transport_is_cfg4_sharded():
    """`bench.py --gpus 2 --force-local`: cfg 4 as BASELINE.json states it (1 B rows in total, sharded — here 2 × 500 M on the one
    GPU of this box, merged through fdb_plan_allreduce over the in-process transport), one JSON line, shard sizes sum to 1 B, the
    merged result checked against the numpy restatement summed over both ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--force-local", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=850, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["workload"].startswith("cfg4: 1000000000 rows sharded over 2 GPUs")
    assert line["config"]["rows_per_gpu"] == [500_000_000, 500_000_000] and sum(line["config"]["rows_per_gpu"]) == 1_000_000_000
    assert line["config"]["total_rows"] == 1_000_000_000
    assert line["checked"]["groups_out"] == 1025 and line["merge_ms"] > 0
    assert len(line["per_rank"]) == 2 and all(r["kernel_frac"] > 0.3 for r in line["per_rank"])
    assert line["roofline"]["kernel"] == "fdb_plan_kernel" and line["devices_used"] == 1
    assert math.isclose(line["value"], 1e9 * 3 / (line["ms_per_step"] * 3e-3), rel_tol=1e-6)
