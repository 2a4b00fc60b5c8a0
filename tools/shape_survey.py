"""Achieved HBM rate of the run-time specialised scan over a spread of plan shapes (tuning aid): the geometry rule and the
code generator should hold up beyond the two benchmark configurations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import And, Col, Count, DynCol, Max, Min, Or, Sum

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
b = synth.prometheus_chunk(0, 0, rows, cfg3=True)
rb = pp.ResidentBatch(b)
V, T = Col("value"), Col("timestamp")
C, M, I, P = Col("labels.code"), Col("labels.method"), Col("labels.instance"), Col("labels.path")
t0 = synth.T0
tmid = t0 + 15_000 * (rows // synth.SERIES // 2)
cases = {
    "cfg2": (C == "200", [Sum(V)], [P]),
    "cfg3": (And(Or(C == "200", C == "500"), M == "GET", I != None), [Count(V), Min(T), Max(T), Sum(V)], [P]),
    "no filter, sum by path": (None, [Sum(V)], [P]),
    "no filter, count by code": (None, [Count(V)], [C]),
    "count(*) no groups": (None, [Count(V)], []),
    "sum no groups": (None, [Sum(V)], []),
    "time range (50 %) sum by path": (And(T >= t0, T < tmid), [Sum(V)], [P]),
    "time range + code, 4 aggs by path": (And(T >= t0, T < tmid, C == "200"), [Count(V), Min(V), Max(V), Sum(V)], [P]),
    "value > 900 (10 %) sum by path": (V > 900.0, [Sum(V)], [P]),
    "code==503 (2 %) sum by path": (C == "503", [Sum(V)], [P]),
    "sum by (code, method)": (None, [Sum(V)], [C, M]),
    "sum by (path, method) 5k slots": (None, [Sum(V)], [P, M]),
    "sum(value * 2.0) by path": (C == "200", [Sum(V * 2.0)], [P]),
    "min/max(ts - 1000) by path": (C == "200", [Min(T - 1000), Max(T - 1000)], [P]),
    "distinct path": (None, [], [P]),
    "sum by instance (513 slots, 5 % NULL)": (None, [Sum(V)], [I]),
    "sum by (path, instance) 526k slots": (None, [Sum(V)], [P, I]),
    "4 aggs by (path, instance) 526k slots": (C == "200", [Count(V), Min(T), Max(T), Sum(V)], [P, I]),
    "sum by (path, instance, method) hash": (None, [Sum(V)], [P, I, M]),
    "by all labels (hash: 4 cols)": (None, [Sum(V)], [DynCol("labels")]),
}
only = sys.argv[2].split("|") if len(sys.argv) > 2 else None
for name, (f, aggs, G) in cases.items():
    if only is not None and name not in only:
        continue
    tot, nb, n, kern = 0.0, 0, 0, ""
    for it in range(7):
        plan = pp.HashAggregatePlan(f, aggs, G)
        plan.set_timing(True)
        plan.Callback(rb)
        kern = plan.last_kernel()
        out = plan.Finish()
        st = plan.stats()
        plan.Close()
        if it >= 2:
            tot += st["kernel_ms"]; nb += st["algorithmic_bytes"]; n += st["launches"]
    print(f"{name:38s} {kern:18s} {tot / max(n, 1):8.4f} ms/launch  {nb / max(tot, 1e-9) / 1e6:8.1f} GB/s  {nb / 5 / rows:6.2f} B/row  groups {out.num_rows}", flush=True)
