#!/usr/bin/env python
"""End-to-end number for SURVEY §8(f).3: Parquet row groups (bytes in host memory) → resident batches decoded on the device →
the cfg 2 query, against the same data imported as Arrow records decoded by pyarrow on the host (what the reference's
ParquetConverter does with parquet-go). Prints one JSON line. Run on the GPU box."""
import io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
from frostdb_amd import build as fb
fb.build()
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col, Sum
from tests.parquet_util import row_group_chunks, write_parquet

rows, rg_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 5_000_000
rec = synth.prometheus_chunk(0, 0, rows)
t = pa.Table.from_batches([rec])
t = t.set_column(0, "labels.code", t.column(0).cast(pa.binary())).set_column(1, "labels.path", t.column(1).cast(pa.binary()))
variant = os.environ.get("PQ_VARIANT", "plain")  # "plain": UNCOMPRESSED, PLAIN int64; "delta_snappy": timestamp DELTA_BINARY_PACKED, pages SNAPPY
kw = {}
if variant == "delta_snappy":
    kw = dict(compression="SNAPPY", column_encoding={"timestamp": "DELTA_BINARY_PACKED"}, use_dictionary=["labels.code", "labels.path"])
data = write_parquet(t, row_group_size=rg_rows, data_page_size=1 << 20, **kw)
n_rg = pq.ParquetFile(io.BytesIO(data)).metadata.num_row_groups
groups = [row_group_chunks(data, g) for g in range(n_rg)]
# the file's bytes in PINNED host memory (what a host that reads parts for the GPU would read into); chunks are (address, length)
import torch
pinned = torch.empty(len(data), dtype=torch.uint8, pin_memory=True)
pinned.numpy()[:] = np.frombuffer(data, dtype=np.uint8)
def _pin(ch):
    out = []
    for nm, ty, opt, u8, b, cd in ch:
        off = data.find(b[:64]) if len(b) >= 64 else data.find(b)
        assert data[off:off + len(b)] == b
        out.append((nm, ty, opt, u8, (pinned.data_ptr() + off, len(b)), cd))
    return out
groups = [(_pin(ch), n) for ch, n in groups]
q = (Col("labels.code") == "200", [Sum(Col("value"))], [Col("labels.path")])

def run_device():
    plan = pp.HashAggregatePlan(*q)
    # every row group in one call (fdb_batches_from_parquet); $PQ_ONE_BY_ONE: one call per row group, one after the other
    keep = [pp.ResidentBatch.from_parquet(ch, n) for ch, n in groups] if os.environ.get("PQ_ONE_BY_ONE") else pp.ResidentBatch.from_parquet_many(groups)
    plan.CallbackResident(keep)
    out = plan.Finish(); plan.Close()
    for k in keep: k.close()
    return out

def run_host_decode():
    plan = pp.HashAggregatePlan(*q)
    pf = pq.ParquetFile(io.BytesIO(data), read_dictionary=["labels.code", "labels.path"])
    for g in range(n_rg):
        for b in pf.read_row_group(g).to_batches():
            plan.Callback(b)
    out = plan.Finish(); plan.Close()
    return out

a, b = run_device(), run_host_decode()
da = dict(zip(a.column(0).to_pylist(), a.column(1).to_pylist())); db = dict(zip(b.column(0).to_pylist(), b.column(1).to_pylist()))
assert da.keys() == db.keys() and all(abs(da[k] - db[k]) <= 1e-9 * abs(db[k]) for k in da)
res = {}
for name, fn in ((("device_decode", run_device),) if os.environ.get("PQ_DEVICE_ONLY") else (("device_decode", run_device), ("host_decode_pyarrow", run_host_decode))):
    fn(); t0 = time.perf_counter(); n = 3
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n
    res[name] = {"s_per_pass": dt, "rows_per_s": rows / dt, "parquet_GB_per_s": len(data) / dt / 1e9}
print(json.dumps({"metric": "rows/sec parquet bytes (host memory) → filter + aggregate result", "variant": variant, "rows": rows, "row_groups": n_rg, "parquet_bytes": len(data), **res}))
