"""The synthetic workloads bench.py times are the ones BASELINE.json's configs describe: schema, value
distributions, bytes per row, number of distinct groups — and any rank can regenerate any chunk."""
import numpy as np
import pyarrow as pa

from frostdb_amd import synth


def _payload_bytes(rec: pa.RecordBatch, cols):
    """Algorithmic bytes the way SURVEY §8d counts them: values / indices buffer + validity bitmap where the column has NULLs."""
    total = 0.0
    for name in cols:
        c = rec.column(name)
        width = 4 if pa.types.is_dictionary(c.type) else 8
        total += rec.num_rows * width + (rec.num_rows / 8.0 if c.null_count > 0 else 0.0)
    return total


def test_cfg2_chunk_matches_the_described_workload():
    n = 400_000
    b = synth.prometheus_chunk(0, 3, n, row_base=3 * n)
    assert b.schema.names == ["labels.code", "labels.path", "timestamp", "value"]
    assert b.schema.field("labels.code").type == pa.dictionary(pa.uint32(), pa.binary())
    assert b.schema.field("value").type == pa.float64() and b.schema.field("timestamp").type == pa.int64()
    assert b.column("value").null_count == 0 and b.column("timestamp").null_count == 0
    codes = b.column("labels.code").dictionary_decode().to_pylist()
    share = codes.count(b"200") / n
    assert abs(share - 0.70) < 0.01  # labels.code == '200' for 70 % of the rows
    paths = b.column("labels.path")
    assert len(paths.dictionary) == 1024 and paths.null_count > 0
    v = b.column("value").to_numpy()
    assert 0.0 <= v.min() and v.max() < 1000.0 and abs(v.mean() - 500.0) < 5.0
    ts = b.column("timestamp").to_numpy()
    assert np.all(np.diff(ts) >= 0) and (ts[0] - synth.T0) % 15_000 == 0
    # 16.25 B/row: code idx 4 + path idx 4 + value 8 + two validity bits (cfg 2)
    assert abs(_payload_bytes(b, ["labels.code", "labels.path", "value"]) / n - 16.25) < 1e-9


def test_cfg3_chunk_adds_method_and_instance():
    n = 200_000
    b = synth.prometheus_chunk(1, 0, n, cfg3=True)
    assert set(b.schema.names) == {"labels.code", "labels.path", "labels.method", "labels.instance", "timestamp", "value"}
    assert len(b.column("labels.method").dictionary) == 4 and len(b.column("labels.instance").dictionary) == 512
    assert abs(b.column("labels.instance").null_count / n - 0.05) < 0.005
    # 32.5 B/row: four dictionary columns + value + timestamp + four validity bits
    cols = ["labels.code", "labels.path", "labels.method", "labels.instance", "value", "timestamp"]
    assert abs(_payload_bytes(b, cols) / n - 32.5) < 1e-9


def test_chunks_are_reproducible_and_distinct():
    a = synth.prometheus_chunk(2, 5, 10_000, row_base=50_000)
    b = synth.prometheus_chunk(2, 5, 10_000, row_base=50_000)
    c = synth.prometheus_chunk(2, 6, 10_000, row_base=60_000)
    d = synth.prometheus_chunk(3, 5, 10_000, row_base=50_000)
    assert a.equals(b) and not a.equals(c) and not a.equals(d)
    total = sum(r.num_rows for r in synth.prometheus_batches(0, 25_000, 10_000))
    assert total == 25_000


def test_cfg5_chunk_has_32_label_columns_and_the_requested_groups():
    n_groups = 3000
    b = synth.cfg5_chunk(0, 0, 60_000, n_groups=n_groups)
    labels = [n for n in b.schema.names if n.startswith("labels.")]
    assert len(labels) == 32 and "value" in b.schema.names
    keys = pa.table([b.column(n).dictionary_decode() for n in labels], names=labels)
    distinct = keys.group_by(labels, use_threads=False).aggregate([]).num_rows
    assert 0.98 * n_groups <= distinct <= n_groups  # every group shows up once rows ≫ groups
    nulls = sum(b.column(n).null_count for n in labels) / (32 * b.num_rows)
    assert 0.01 < nulls < 0.06  # ≈3 % of the label digits are NULL
    assert synth.cfg5_chunk(0, 0, 1000, n_groups=n_groups).equals(synth.cfg5_chunk(0, 0, 1000, n_groups=n_groups))
