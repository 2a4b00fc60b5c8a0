"""Host cost of fdb_plan_push per record size (tuning aid): the record is exported to the Arrow C Data Interface ONCE and pushed
many times through the C ABI directly, so pyarrow's per-call export cost is not part of the number."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.arrow_c import ExportedBatch
from frostdb_amd.logicalplan import Col, Sum

L = pp.lib()
for rows in (1024, 8192, 65536, 524288):
    b = synth.prometheus_chunk(0, 0, rows)
    n = max(20, min(2000, 40_000_000 // rows))
    plan = pp.HashAggregatePlan(Col("labels.code") == "200", [Sum(Col("value"))], [Col("labels.path")])
    with ExportedBatch(b) as ex:
        for _ in range(5):
            L.fdb_plan_push(plan.handle, ctypes.addressof(ex.array), ctypes.addressof(ex.schema))
        plan.num_groups()
        t = time.perf_counter()
        for _ in range(n):
            rc = L.fdb_plan_push(plan.handle, ctypes.addressof(ex.array), ctypes.addressof(ex.schema))
            assert rc == 0
        plan.num_groups()
        dt = time.perf_counter() - t
    plan.Close()
    print(f"{rows:8d}-row records: {dt / n * 1e6:8.1f} us per push  {rows * n / dt / 1e9:6.3f} G rows/s (one chain)", flush=True)
