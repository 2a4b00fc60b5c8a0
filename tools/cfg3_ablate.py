"""Which part of cfg 3 costs what (tuning aid): kernel time for variations of the filter / aggregate list."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import And, Col, Count, Max, Min, Or, Sum

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
rpt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
b = synth.prometheus_chunk(0, 0, rows, cfg3=True)
rb = pp.ResidentBatch(b)
F3 = And(Or(Col("labels.code") == "200", Col("labels.code") == "500"), Col("labels.method") == "GET", Col("labels.instance") != None)
F2 = Col("labels.code") == "200"
G = [Col("labels.path")]
V, T = Col("value"), Col("timestamp")
C, M, I = Col("labels.code"), Col("labels.method"), Col("labels.instance")
cases = {
    "cfg3 full": (F3, [Count(V), Min(T), Max(T), Sum(V)]),
    "cfg3 filter, sum": (F3, [Sum(V)]),
    "code==200": (C == "200", [Sum(V)]),
    "code==200|500": (Or(C == "200", C == "500"), [Sum(V)]),
    "code==200 & method==GET": (And(C == "200", M == "GET"), [Sum(V)]),
    "method==GET": (M == "GET", [Sum(V)]),
    "method!=GET": (M != "GET", [Sum(V)]),
    "code==200 & inst!=NULL": (And(C == "200", I != None), [Sum(V)]),
    "inst!=NULL": (I != None, [Sum(V)]),
    "inst==NULL": (I == None, [Sum(V)]),
    "code==503 (2%)": (C == "503", [Sum(V)]),
    "code==404 (10%)": (C == "404", [Sum(V)]),
    "code!=200 (30%)": (C != "200", [Sum(V)]),
    "no filter": (None, [Sum(V)]),
}
grids = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
for name, (f, aggs) in cases.items():
    if only is not None and name not in only:
        continue
    for grid in grids:
        tot, nb, n = 0.0, 0, 0
        for it in range(8):
            plan = pp.HashAggregatePlan(f, aggs, G)
            plan.set_tuning(rpt, (grid & 0xFFFFF) | ((grid >> 20) << 25))
            plan.set_timing(True)
            plan.Callback(rb)
            plan.Finish()
            st = plan.stats()
            plan.Close()
            if it >= 3:
                tot += st["kernel_ms"]; nb += st["algorithmic_bytes"]; n += 1
        print(f"{name:28s} grid={grid & 0xFFFFF:5d} sub={grid >> 20} {tot / n:8.4f} ms  {nb / tot / 1e6:8.1f} GB/s  {nb / n / rows:6.2f} B/row")
