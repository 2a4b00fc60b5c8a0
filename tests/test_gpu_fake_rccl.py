"""The RCCL transport of the cross-GPU merge (frostdb_amd/csrc/fdb_comm.cpp: RcclComm) with MORE THAN ONE RANK on the 1-GPU box.

The real librccl refuses two ranks of a communicator on one device, so until round 4 this transport — unique id →
ncclCommInitRank / ncclCommInitAll, the grouped in-place all-reduce, the all-gathers of the schema agreement, the grouped and
sliced ncclSend / ncclRecv of the exchange, the failure vote — had only ever met one rank. Here the library binds a test-only
stand-in (tests/fake_rccl/fake_rccl.cpp, loaded through $FDB_RCCL_LIB) that implements the same entry points over a
memory-mapped file, and 2 / 4 / 8 ranks — threads for fdb_comm_init_all, separate PROCESSES for fdb_comm_init_rank — run the same
plan-level merges as tests/test_gpu_comm.py, against the oracle. The exchange's slice size is forced down to 1 MiB
(FDB_EXCHANGE_SLICE_BYTES) so that slicing happens. Everything runs in child processes: the binding of librccl is per process.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_lib():
    sys.path.insert(0, os.path.join(ROOT, "tests", "fake_rccl"))
    import importlib
    build = importlib.import_module("build")
    return build.build()


def child_env(fake_lib, **extra):
    env = dict(os.environ)
    env.update({"FDB_RCCL_LIB": fake_lib, "FDB_EXCHANGE_SLICE_BYTES": str(1 << 20), "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", ""),
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra)
    return env


THREADS_CHILD = r'''
import json, sys, threading
import numpy as np
import pyarrow as pa
sys.path.insert(0, %(root)r)
from frostdb_amd import physicalplan as pp, comm as fcomm
from frostdb_amd.logicalplan import Col, Count, DynCol, Max, Min, Or, Sum
from tests.test_gpu_parity import CFG2, CFG3, assert_same_result, run_oracle
from tests.test_gpu_comm import run_ranks, drifting_shard
from tests.util import arrow_to_pydict, make_prometheus_batch

world = int(sys.argv[1])
report = {"world": world}
comms = fcomm.Comm.init_all([0] * world)   # ncclCommInitAll of the bound library: `world` ranks on device 0
assert [c.rank for c in comms] == list(range(world)) and all(c.size == world for c in comms)
report["transport_ranks"] = [c.transport_ranks for c in comms]

# 1. aligned dense tables: grouped in-place all-reduce
for name, q in (("cfg2", CFG2), ("cfg3", CFG3)):
    rng = np.random.default_rng(5 + world)
    shards = [[make_prometheus_batch(rng, 20_000 + 1_000 * r, n_path=40, null_frac=0.0)] for r in range(world)]
    want = run_oracle([b for s in shards for b in s], **q)
    cols = ["labels.path"] + [a.Name() for a in q["aggs"]]
    def rank_fn(r):
        plan = pp.HashAggregatePlan(q["filter_expr"], q["aggs"], q["groups"])
        keep = [pp.ResidentBatch(b) for b in shards[r]]
        try:
            plan.CallbackResident(keep)
            assert comms[r].allreduce(plan) is True
            return arrow_to_pydict(plan.Finish())
        finally:
            plan.Close()
            for k in keep: k.close()
    for got in run_ranks(world, rank_fn):
        assert_same_result(got, want, cols, float_cols={"sum(value)"})
report["allreduce"] = "ok"

# 2. different dictionaries / column sets: schema all-gather + exchange, shards disjoint, union = oracle
rng = np.random.default_rng(77)
shards = [drifting_shard(rng, r) for r in range(world)]
filt = Or(Col("labels.code") == "200", Col("labels.code") == "500")
aggs = [Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp")), Sum(Col("value"))]
groups = [DynCol("labels")]
want = run_oracle(shards, filt, aggs, groups)
def rank_fn(r):
    plan = pp.HashAggregatePlan(filt, aggs, groups)
    try:
        plan.Callback(shards[r])
        assert comms[r].allreduce(plan) is False
        shard = comms[r].merge_alltoall(plan)
        try: return arrow_to_pydict(shard.Finish())
        finally: shard.Close()
    finally:
        plan.Close()
parts = run_ranks(world, rank_fn)
cols = sorted({c for p in parts for c in p if c.startswith("labels.")}) + [a.Name() for a in aggs]
merged = {c: [] for c in cols}
for p in parts:
    n = len(next(iter(p.values()))) if p else 0
    for c in cols: merged[c] += p.get(c, [None] * n)
assert sum(len(next(iter(p.values()))) if p else 0 for p in parts) == len(want["count(value)"])
assert_same_result(merged, want, cols, float_cols={"sum(value)"})
report["exchange"] = "ok"

# 3. hash-mode tables big enough that a rank's partition for one peer exceeds the forced 1 MiB slice: several send/recv rounds
n_cols, n = 12, 120_000
rng = np.random.default_rng(3)
def big(r):
    g = rng.integers(0, 90_000, n)
    arrays, names = [], []
    for c in range(n_cols):
        digit = ((g >> (2 * c)) & 3).astype(np.uint32) if c < 9 else ((g * (c + 3)) %% 5).astype(np.uint32)
        arrays.append(pa.DictionaryArray.from_arrays(pa.array(digit), pa.array([b"c%%02d=%%d" %% (c, k) for k in range(5)], type=pa.binary())))
        names.append("labels.l%%02d" %% c)
    arrays.append(pa.array(rng.uniform(0, 10, n))); names.append("value")
    return pa.RecordBatch.from_arrays(arrays, names=names)
shards = [big(r) for r in range(world)]
aggs, groups = [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")]
want = run_oracle(shards, None, aggs, groups)
def rank_fn(r):
    plan = pp.HashAggregatePlan(None, aggs, groups)
    rb = pp.ResidentBatch(shards[r])
    try:
        plan.Callback(rb)
        return arrow_to_pydict(comms[r].merge(plan))
    finally:
        plan.Close(); rb.close()
parts = run_ranks(world, rank_fn)
cols = ["labels.l%%02d" %% c for c in range(n_cols)] + [a.Name() for a in aggs]
merged = {c: sum((p[c] for p in parts), []) for c in cols}
assert len(merged["count(value)"]) == len(want["count(value)"])
assert_same_result(merged, want, cols, float_cols={"sum(value)"})
report["sliced_exchange"] = {"groups": len(want["count(value)"]), "rows_bytes_per_rank_estimate": len(want["count(value)"]) * (16 + n_cols * 4 + 24)}

# 4. the failure vote: rank 1 fails before the merge's first collective; nobody hangs, everybody errors; the communicator survives
if world >= 3:
    import os
    shards = [make_prometheus_batch(np.random.default_rng(11 + r), 5_000, n_path=20, null_frac=0.0) for r in range(world)]
    for how in ("allreduce", "exchange"):
        os.environ["FDB_TEST_FAIL_MERGE_RANK"] = "1"
        errors = [None] * world
        def rank_fn(r):
            plan = pp.HashAggregatePlan(CFG2["filter_expr"], CFG2["aggs"], CFG2["groups"])
            try:
                plan.Callback(shards[r])
                try:
                    if how == "allreduce": comms[r].allreduce(plan)
                    else: comms[r].merge_alltoall(plan).Close()
                except pp.FdbError as e:
                    errors[r] = e
            finally:
                plan.Close()
        run_ranks(world, rank_fn)
        del os.environ["FDB_TEST_FAIL_MERGE_RANK"]
        assert all(e is not None for e in errors), errors
        assert "injected failure" in str(errors[1])
        assert all(errors[r].code == pp.FDB_ERR_STATE for r in range(world) if r != 1)
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
    report["failure_vote"] = "ok"
# 5. the rank-ordered merge of small dense tables (one all-gather + a local fold in rank order): float64 sums are bit-identical on
#    every rank and in every run whatever order the ranks reach the merge in, and equal the ranks' partial sums added in rank order
import time
shards = []
for r in range(world):
    b = make_prometheus_batch(np.random.default_rng(40 + r), 30_000, n_path=40, null_frac=0.0)
    g = np.random.default_rng(90 + r)
    wild = g.standard_normal(b.num_rows) * 10.0 ** g.integers(-8, 9, b.num_rows)  # magnitudes 1e-8 … 1e8: the sum's bits depend on its order
    shards.append(b.set_column(b.schema.get_field_index("value"), "value", pa.array(wild)))
def bits(out):
    key = out.column(out.schema.names.index("labels.path")).dictionary_decode().to_pylist()
    return dict(zip(key, out.column(out.schema.names.index("sum(value)")).to_numpy().view(np.uint64).tolist()))
def one_pass(seed, merge=True):
    delays = np.random.default_rng(seed).permutation(world) * 0.03
    def rank_fn(r):
        plan = pp.HashAggregatePlan(CFG2["filter_expr"], CFG2["aggs"], CFG2["groups"])
        plan.set_deterministic(True)
        rb = pp.ResidentBatch(shards[r])
        try:
            plan.CallbackResident([rb])
            if merge:
                time.sleep(float(delays[r]))  # a different arrival order every pass
                assert comms[r].allreduce(plan) is True
            return bits(plan.Finish())
        finally:
            plan.Close(); rb.close()
    return run_ranks(world, rank_fn)
a, b = one_pass(1), one_pass(2)
assert all(x == a[0] for x in a), "ranks disagree on the merged sums' bits"
assert a == b, "the merged sums' bits depend on the order the ranks arrived in"
parts = one_pass(3, merge=False)
expect = {}
for k in a[0]:
    acc = None
    for r in range(world):
        v = float(np.array([parts[r].get(k, 0)], dtype=np.uint64).view(np.float64)[0])
        acc = v if acc is None else acc + v
    expect[k] = int(np.array([acc], dtype=np.float64).view(np.uint64)[0])
assert a[0] == expect, "the merge is not the rank-ordered fold of the ranks' partial sums"
report["rank_ordered_merge"] = {"groups": len(expect)}
for c in comms: c.close()
print("REPORT " + json.dumps(report))
'''


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_transport_with_several_ranks_as_threads_init_all(fake_lib, world):
    p = subprocess.run([sys.executable, "-c", THREADS_CHILD % {"root": ROOT}, str(world)], capture_output=True, text=True, timeout=550,
                       env=child_env(fake_lib), cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    rep = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("REPORT ")][-1][7:])
    assert rep["world"] == world and rep["transport_ranks"] == [world] * world
    assert rep["allreduce"] == "ok" and rep["exchange"] == "ok"
    assert rep["rank_ordered_merge"]["groups"] > 30  # float sums bit-identical across ranks, runs and arrival orders (section 5 of the child)
    if world <= 4:
        assert rep["sliced_exchange"]["rows_bytes_per_rank_estimate"] / world > (1 << 20)  # more than one slice per peer
    if world >= 3:
        assert rep["failure_vote"] == "ok"


PROC_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from frostdb_amd import physicalplan as pp, comm as fcomm
from frostdb_amd.logicalplan import Col, Count, DynCol, Max, Min, Or, Sum
from tests.test_gpu_parity import CFG3
from tests.test_gpu_comm import drifting_shard
from tests.util import arrow_to_pydict, make_prometheus_batch

uid, world, rank, out_path = bytes.fromhex(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
comm = fcomm.Comm(uid, world, rank, 0)   # ncclCommInitRank of the bound library: one PROCESS per rank, all on device 0
assert (comm.rank, comm.size, comm.transport_ranks) == (rank, world, world)
res = {}
# aligned: every rank's Finish after the in-place all-reduce
rng = np.random.default_rng(500 + rank)
rec = make_prometheus_batch(rng, 30_000 + 777 * rank, n_path=40, null_frac=0.0)
plan = pp.HashAggregatePlan(CFG3["filter_expr"], CFG3["aggs"], CFG3["groups"])
plan.Callback(rec)
assert comm.allreduce(plan) is True
res["aligned"] = arrow_to_pydict(plan.Finish())
plan.Close()
# unaligned: the exchange's shard of this rank
rec2 = drifting_shard(np.random.default_rng(900 + rank), rank)
filt = Or(Col("labels.code") == "200", Col("labels.code") == "500")
aggs = [Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp")), Sum(Col("value"))]
plan = pp.HashAggregatePlan(filt, aggs, [DynCol("labels")])
plan.Callback(rec2)
assert comm.allreduce(plan) is False
shard = comm.merge_alltoall(plan)
res["shard"] = arrow_to_pydict(shard.Finish())
shard.Close(); plan.Close()
comm.close()
def enc(d): return {k: [x.decode("latin1") if isinstance(x, bytes) else x for x in v] for k, v in d.items()}
json.dump({k: enc(v) for k, v in res.items()}, open(out_path, "w"))
'''


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_rccl_transport_with_one_process_per_rank_init_rank(fake_lib, world, tmp_path):
    import numpy as np
    sys.path.insert(0, ROOT)
    from frostdb_amd import physicalplan as pp  # noqa: F401  (this process only needs a unique id's worth of bytes: any 128 will do)
    from frostdb_amd.logicalplan import Col, Count, DynCol, Max, Min, Or, Sum
    from tests.test_gpu_comm import drifting_shard
    from tests.test_gpu_parity import CFG3, assert_same_result, run_oracle
    from tests.util import make_prometheus_batch
    uid = os.urandom(16).hex() + "00" * 112
    procs = [subprocess.Popen([sys.executable, "-c", PROC_CHILD % {"root": ROOT}, uid, str(world), str(r), str(tmp_path / f"r{r}.json")],
                              env=child_env(fake_lib), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=500) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-4000:])

    def dec(d):
        return {k: [x.encode("latin1") if isinstance(x, str) else x for x in v] for k, v in d.items()}
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(world)]
    recs = [make_prometheus_batch(np.random.default_rng(500 + r), 30_000 + 777 * r, n_path=40, null_frac=0.0) for r in range(world)]
    want = run_oracle(recs, **CFG3)
    cols = ["labels.path"] + [a.Name() for a in CFG3["aggs"]]
    for r in range(world):
        assert_same_result(dec(res[r]["aligned"]), want, cols, float_cols={"sum(value)"})
    recs2 = [drifting_shard(np.random.default_rng(900 + r), r) for r in range(world)]
    filt = Or(Col("labels.code") == "200", Col("labels.code") == "500")
    aggs = [Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp")), Sum(Col("value"))]
    want2 = run_oracle(recs2, filt, aggs, [DynCol("labels")])
    parts = [dec(res[r]["shard"]) for r in range(world)]
    cols2 = sorted({c for p in parts for c in p if c.startswith("labels.")}) + [a.Name() for a in aggs]
    merged = {c: [] for c in cols2}
    for p in parts:
        n = len(next(iter(p.values()))) if p else 0
        for c in cols2:
            merged[c] += p.get(c, [None] * n)
    assert len(merged["count(value)"]) == len(want2["count(value)"])
    assert_same_result(merged, want2, cols2, float_cols={"sum(value)"})


@pytest.mark.timeout(900)
def test_bench_gpus_2_one_process_per_rank_reports_two_ranks_inside_the_communicator(fake_lib):
    """`python bench.py --gpus 2` in the driver's own form (self-launch → torch.distributed.run → one process per rank →
    fdb_comm_init_rank) on this 1-GPU box: the ranks share the device (FDB_BENCH_TEST_SHARE_DEVICE), the control plane runs over
    gloo, the merge over the stand-in library — the line's rccl_ranks_seen must say 2 on both ranks, the merged result is checked
    inside bench.py against the numpy expectation summed over the ranks."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "40000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=850, cwd=ROOT, env=child_env(fake_lib, FDB_BENCH_TEST_SHARE_DEVICE="1"))
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == [2, 2] and line["devices_used"] == 1
    assert "fdb_comm_init_rank" in line["config"]["parallelism"] and "fallback" not in line["config"]["parallelism"]
    assert line["config"]["rows_per_gpu"] == [20_000_000, 20_000_000] and line["checked"]["groups_out"] == 1025
