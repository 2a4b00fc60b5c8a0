"""Tuning aid: where a small shard's step goes. One cfg 2 step over `rows` resident rows (default 125 M: a rank's shard of cfg 4 at
N = 8) = plan create → CallbackResident (one launch over every record) → Finish → Close; prints the host time of each call
(median over the passes), the kernel's own time (hipEvents, separate passes) and, with FDB_PROFILE=1, the library's per-phase
lines of the last pass.   python tools/step_probe.py [rows] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import to_desc
import bench

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 200
br = 25_000_000
recs = [pp.ResidentBatch(synth.prometheus_chunk(0, i, min(br, rows - i * br), row_base=i * br, cfg3=os.environ.get("CFG", "2") == "3")) for i in range((rows + br - 1) // br)]
cfg = int(os.environ.get("CFG", "2"))
filt, aggs, groups, _ = bench.query(cfg)
desc = to_desc(filt, aggs, groups)


def step(timing=False):
    t = [time.perf_counter()]
    plan = pp.HashAggregatePlan(filt, aggs, groups, desc=desc)
    if timing:
        plan.set_timing(True)
    plan.set_tuning(0, 0)
    t.append(time.perf_counter())
    plan.CallbackResident(recs)
    t.append(time.perf_counter())
    out = plan.Finish()
    t.append(time.perf_counter())
    st = plan.stats() if timing else None
    plan.Close()
    t.append(time.perf_counter())
    del out
    return [(b - a) * 1e6 for a, b in zip(t, t[1:])], st


quiet = os.environ.pop("FDB_PROFILE", None)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
parts = [step()[0] for _ in range(passes)]
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / passes * 1e6
med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
names = ["create", "push (returns after the launches)", "finish (waits for the scan)", "stats + close"]
print(f"rows {rows} records {len(recs)} passes {passes}: wall {wall:.1f} us/step")
for i, n in enumerate(names):
    print(f"  {n:38s} {med([p[i] for p in parts]):8.1f} us")
ks = sorted(step(True)[1]["kernel_ms"] * 1e3 for _ in range(30))
print(f"  kernel (hipEvents, median of 30)       {ks[15]:8.1f} us   → outside the kernel: {wall - ks[15]:.1f} us/step")
if quiet is not None:
    os.environ["FDB_PROFILE"] = quiet
    sys.stderr.flush()
    step()
