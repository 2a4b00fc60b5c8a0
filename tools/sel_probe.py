"""Tuning aid: times fdb_plan_filter_batches (value > 500 over 4 × 25 M resident rows) without checking results — for kernel
variants that deliberately break them (FDB_COMPACT_BLOCKS_PER_CU). Prints kernel ms per pass (hipEvents) and wall ms."""
import os, sys, time
sys.path.insert(0, os.environ.get("FDB_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (FDB_PKG_ROOT: A/B against another build of the package)
import torch
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col
which = os.environ.get("FDB_PROBE_FILTER", "value")  # value: `value > 500` (50 %); cfg3: cfg 3's predicate on cfg 3's columns (18.5 %, three dictionary leaves); both: value AND code (35 %)
recs = [pp.ResidentBatch(synth.prometheus_chunk(0, i, 25_000_000, row_base=i * 25_000_000, cfg3=which == "cfg3")) for i in range(4)]
if which == "cfg3":
    import bench
    filt = bench.query(3)[0]
elif which == "both":
    from frostdb_amd.logicalplan import And
    filt = And(Col("value") > 500.0, Col("labels.code") == "200")
else:
    filt = Col("value") > 500.0
def step(timing=False):
    plan = pp.HashAggregatePlan(filt)
    plan.set_timing(timing)
    outs = plan.FilterResidentMany(recs)
    st = plan.stats() if timing else None
    plan.Close()
    for o in outs: o.close()
    return st
step(); step()
N = int(os.environ.get("FDB_PROBE_PASSES", "30"))
torch.cuda.synchronize(); t0 = time.perf_counter(); ks = []
for _ in range(N): ks.append(step(True)["kernel_ms"])
torch.cuda.synchronize()
ks.sort()
print(which, "two-pass" if os.environ.get("FDB_SELECT_TWO_PASS") else "one-pass", "kernel_ms median", round(ks[N // 2], 4), "min", round(ks[0], 4), "wall_ms", round((time.perf_counter() - t0) / N * 1e3, 4))
