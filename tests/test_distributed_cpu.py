"""world_size-2 (and 3) gloo tests of the cross-rank merge (frostdb_amd/distributed.py) on CPU.

Each rank aggregates its own shard of records (here: with the oracle, because there is no GPU in this
container — on the GPU box the same merge code receives tensors filled by fdb_plan_partial_state), then the
ranks unify keys and all-reduce. The merged record must equal the oracle run over ALL records.
"""
import os
import socket

import numpy as np
import pyarrow as pa
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from frostdb_amd.logicalplan import AGG_COUNT, Col, Count, DynCol, Max, Min, Sum
from tests.util import make_prometheus_batch, sort_key, batch_rows, dict_array


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_records(world, seed=123):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(2 * world + 1):
        b = make_prometheus_batch(rng, 3000 + 500 * i, n_path=10 + 7 * i, with_method=False)
        if i % 2 == 1:  # some records carry an extra dynamic label column
            b = b.append_column("labels.zone", dict_array([None if rng.random() < 0.4 else b"z%d" % rng.integers(3) for _ in range(b.num_rows)]))
        recs.append(b)
    return recs


def _oracle(records, filter_expr, aggs, groups):
    from oracle import OraclePlan
    p = OraclePlan(filter_expr, aggs, groups)
    for r in records:
        p.push(r)
    d = p.finish().to_pydict()
    p.close()
    return d


def _partial_as_arrow(d, aggs):
    """Oracle partial result → (keys RecordBatch, tensors) in the shape fdb_plan_partial_keys/state produce."""
    key_names = [k for k in d.keys() if not any(k == a.Name() for a in aggs)]
    n = len(next(iter(d.values()))) if d else 0
    keys = pa.RecordBatch.from_arrays([dict_array(d[k]) for k in key_names], names=key_names) if key_names else \
        pa.RecordBatch.from_arrays([pa.array([0] * n)], names=["_"]).select([])
    tensors = []
    for a in aggs:
        v = d.get(a.Name(), [])
        isf = any(isinstance(x, float) for x in v)
        tensors.append(torch.tensor(v, dtype=torch.float64 if isf else torch.int64))
    return keys, tensors


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frostdb_amd.distributed import merge_partials
        recs = _shard_records(world)
        mine = [r for i, r in enumerate(recs) if i % world == rank]
        f = Col("labels.code") == "200"
        aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp"))]
        groups = [DynCol("labels")]
        if rank == world - 1 and world == 3:
            mine = []  # one rank with an empty shard
        d = _oracle(mine, f, aggs, groups) if mine else {}
        keys, tensors = _partial_as_arrow(d, aggs)
        if not mine:
            keys = pa.RecordBatch.from_arrays([], names=[])
            tensors = [torch.zeros(0, dtype=torch.int64) for _ in aggs]
        out = merge_partials(keys, tensors, aggs)
        if rank == 0:
            cols = out.schema.names
            res = {}
            for n, c in zip(cols, out.columns):
                if pa.types.is_dictionary(c.type):
                    c = c.dictionary_decode()
                res[n] = c.to_pylist()
            q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_merge_partials_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    recs = _shard_records(world)
    if world == 3:
        recs = [r for i, r in enumerate(recs) if i % world != world - 1]
    f = Col("labels.code") == "200"
    aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp"))]
    want = _oracle(recs, f, aggs, [DynCol("labels")])
    cols = ["labels.code", "labels.path", "labels.zone"] + [a.Name() for a in aggs]
    g = sorted(batch_rows(res, cols), key=lambda r: sort_key(r[:3]))
    w = sorted(batch_rows(want, cols), key=lambda r: sort_key(r[:3]))
    assert len(g) == len(w)
    for a, b in zip(g, w):
        assert a[:3] == b[:3] and a[4:] == b[4:]
        assert abs(a[3] - b[3]) <= 1e-9 * abs(b[3])


def test_unify_keys_handles_missing_columns_and_order():
    from frostdb_amd.distributed import unify_keys
    names, keys, perms = unify_keys([["a"], ["b", "a"]], [[(b"x",), (None,)], [(b"q", b"x"), (None, b"x"), (None, None)]])
    assert names == ["a", "b"]
    assert keys == [(b"x", None), (None, None), (b"x", b"q")]
    assert perms == [[0, 1], [2, 0, 1]]


# ---- hash-partitioned all-to-all (high-cardinality merge): schema agreement + exchange bookkeeping on gloo ----------------

def _alltoall_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frostdb_amd.distributed import exchange_rows, unify_group_schemas
        # (1) schema agreement: every rank contributes different dictionaries / columns and all must derive the same global schema
        mine = [("labels.a", "dict", "binary", [b"r%d" % rank, b"shared"]), ("ts", "plain", "int64", None)]
        if rank % 2 == 1:
            mine.append(("labels.odd", "dict", "binary", [b"o%d" % rank]))
        if rank == 0:
            mine.append(("sum(value)", "plain", "double", None))
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        schema = unify_group_schemas(gathered)
        sig = [(f.name, str(f.type), c.dictionary.to_pylist() if pa.types.is_dictionary(f.type) else None) for f, c in zip(schema.schema, schema.columns)]
        # (2) exchange: rows [key, count, sum-as-int], key k is owned by rank k % world; rows are sent grouped by owner
        rng = np.random.default_rng(1000 + rank)
        n = 5000 + 700 * rank
        keys = rng.integers(0, 3000, size=n)
        order = np.argsort(keys % world, kind="stable")
        keys = keys[order]
        rows = np.stack([keys, np.ones(n, dtype=np.int64), keys * 3 + rank], axis=1).astype(np.int64)
        counts = np.bincount(keys % world, minlength=world).tolist()
        send = torch.from_numpy(rows.reshape(-1).copy())
        recv, rc = exchange_rows(send, counts, 3)
        recv_sliced, rc2 = exchange_rows(send, counts, 3, chunk_bytes=24 * 700)  # many slices, uneven tails
        assert rc2 == rc and torch.equal(recv, recv_sliced)
        got = recv.numpy().reshape(-1, 3)
        assert sum(rc) == got.shape[0]
        assert np.all(got[:, 0] % world == rank)  # only keys this rank owns arrive here
        q.put((rank, sig, rc, int(got[:, 1].sum()), int(got[:, 2].sum()), counts))
    finally:
        dist.destroy_process_group()


def test_group_schema_agreement_keeps_key_column_types():
    """Schema agreement for the hash-partitioned exchange (host logic): dictionary columns unite their value sets; a PLAIN string /
    binary key column travels as its value set too (signed index type is the marker fdb_plan_group_schema / fdb_plan_seed_groups
    use) and keeps its own type, large or not; int64 / uint64 / bool key columns and the aggregate value types pass through."""
    from frostdb_amd.distributed import _schema_to_obj, unify_group_schemas

    def rec(cols):
        return pa.RecordBatch.from_arrays([c for _, c in cols], names=[n for n, _ in cols])

    def dic(index_type, values, value_type):
        return pa.DictionaryArray.from_arrays(pa.array([], type=index_type), pa.array(values, type=value_type))

    r0 = rec([("labels.a", dic(pa.uint32(), [b"x", b"y"], pa.binary())), ("user", dic(pa.int32(), ["u1", "u2"], pa.string())),
              ("raw", dic(pa.int32(), [b"r"], pa.large_binary())), ("flag", pa.array([], type=pa.bool_())),
              ("shard", pa.array([], type=pa.uint64())), ("sum(value)", pa.array([], type=pa.float64()))])
    r1 = rec([("user", dic(pa.int32(), ["u2", "u3"], pa.string())), ("labels.a", dic(pa.uint32(), [b"y", b"z"], pa.binary())),
              ("ts", pa.array([], type=pa.int64())), ("flag", pa.array([], type=pa.bool_()))])
    objs = [_schema_to_obj(r0), _schema_to_obj(r1)]
    assert [o[1] for o in objs[0]] == ["dict", "strs", "strs", "plain", "plain", "plain"]
    s = unify_group_schemas(objs)
    assert s.num_rows == 0 and s.schema.names == ["labels.a", "user", "raw", "flag", "shard", "sum(value)", "ts"]
    t = {f.name: f.type for f in s.schema}
    assert t["labels.a"] == pa.dictionary(pa.uint32(), pa.binary()) and s.column(0).dictionary.to_pylist() == [b"x", b"y", b"z"]
    assert t["user"] == pa.dictionary(pa.int32(), pa.string()) and s.column(1).dictionary.to_pylist() == ["u1", "u2", "u3"]
    assert t["raw"] == pa.dictionary(pa.int32(), pa.large_binary())
    assert (t["flag"], t["shard"], t["sum(value)"], t["ts"]) == (pa.bool_(), pa.uint64(), pa.float64(), pa.int64())
    # a column that is a dictionary on one rank and plain strings on another cannot be merged
    bad = rec([("user", dic(pa.uint32(), ["u9"], pa.string()))])
    with pytest.raises(ValueError):
        unify_group_schemas([objs[0], _schema_to_obj(bad)])


@pytest.mark.parametrize("world", [2, 3])
def test_alltoall_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_alltoall_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sigs = [r[1] for r in res]
    assert all(s == sigs[0] for s in sigs)  # identical global schema everywhere
    names = [s[0] for s in sigs[0]]
    assert names[:2] == ["labels.a", "ts"] and "sum(value)" in names and "labels.odd" in names
    a_vals = sigs[0][0][2]
    assert a_vals == [b"r0", b"shared"] + [b"r%d" % r for r in range(1, world)]
    # what rank r received from rank s is what s counted for r; nothing is lost
    for r in range(world):
        assert res[r][2] == [res[s][5][r] for s in range(world)]
    total_rows = sum(5000 + 700 * r for r in range(world))
    assert sum(r[3] for r in res) == total_rows
    expect_sum = 0
    for r in range(world):
        k = np.random.default_rng(1000 + r).integers(0, 3000, size=5000 + 700 * r)
        expect_sum += int((k * 3 + r).sum())
    assert sum(r[4] for r in res) == expect_sum


def test_unify_keys_merges_integer_zero_with_null():
    """int64 key 0 ≡ NULL (both hash to 0 in the reference, dynparquet/hashed.go:254-272; one group on a single GPU): ranks that
    print the group differently still land on ONE global id; strings are untouched ('' stays distinct from NULL)."""
    from frostdb_amd.distributed import unify_keys
    names = [["bucket", "labels.a"], ["bucket", "labels.a"]]
    rows = [[(0, b"x"), (5, b""), (7, None)], [(None, b"x"), (5, None), (7, None)]]
    gnames, gkeys, perms = unify_keys(names, rows)
    assert gnames == ["bucket", "labels.a"]
    assert perms[0] == [0, 1, 2] and perms[1] == [0, 3, 2]
    assert gkeys == [(0, b"x"), (5, b""), (7, None), (5, None)]
