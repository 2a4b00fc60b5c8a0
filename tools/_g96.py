import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, pyarrow as pa
from frostdb_amd import physicalplan as pp
from frostdb_amd.logicalplan import Col, Sum
rng = np.random.default_rng(41)
n = 300_000
ts = np.sort(rng.integers(1000, 4_000_000, n)).astype(np.int64)
lab = rng.integers(0, 6, n)
order = np.lexsort((lab, ts // 1000))
ts, lab = ts[order], lab[order]
labels = pa.DictionaryArray.from_arrays(pa.array(np.where(lab == 5, 0, lab).astype(np.uint32), mask=lab == 5), pa.array([b"a", b"b", b"c", b"d", b"e"], type=pa.binary()))
rec = pa.RecordBatch.from_arrays([pa.array(ts), labels, pa.array(rng.integers(0, 1000, n).astype(np.int64))], names=["timestamp", "labels.x", "v"])
bucket = (Col("timestamp") / 1000 * 1000).Alias("bucket")
for ordered in (True, False):
    os.environ["FDB_RUNS_ALWAYS"] = "1"
    plan = pp.HashAggregatePlan(None, [Sum(Col("v"))], [bucket, Col("labels.x")], ordered=ordered, final_stage=False)
    rb = pp.ResidentBatch(rec)
    plan.CallbackResident([rb])
    print(ordered, plan.last_kernel())
    out = plan.Finish()
    print(out.schema.names, out.num_rows, [c.null_count for c in out.columns])
    print(out.slice(0, 5).to_pydict())
    plan.Close(); rb.close()
