"""Host-side cost of one bench step, piece by piece (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col, Sum

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
nrec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rbs = [pp.ResidentBatch(synth.prometheus_chunk(0, i, rows // nrec, row_base=i * (rows // nrec))) for i in range(nrec)]
filt, aggs, groups = Col("labels.code") == "200", [Sum(Col("value"))], [Col("labels.path")]
acc = {}
def t(name, f):
    t0 = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
N = 50
for it in range(N + 5):
    if it == 5: acc.clear()
    plan = t("create", lambda: pp.HashAggregatePlan(filt, aggs, groups))
    t("tuning", lambda: plan.set_tuning(0, 0))
    t("callback", lambda: plan.CallbackResident(rbs))
    t("finish", lambda: plan.Finish())
    t("close", lambda: plan.Close())
os.environ["FDB_PROFILE"] = "1"
plan = pp.HashAggregatePlan(filt, aggs, groups); plan.CallbackResident(rbs); plan.Finish(); plan.Close()
for k, v in acc.items():
    print(f"{k:10s} {v / N * 1e6:9.1f} us")
print("total", sum(acc.values()) / N * 1e6, "us for", rows, "rows")
