"""Parquet column chunks decoded on the device (fdb_batch_from_parquet) against pyarrow's own reader.

pyarrow writes the files (uncompressed, dictionary-encoded label columns, PLAIN numeric columns, data pages V1 and V2, small
pages so that a chunk has hundreds of them) and reads them back as the expected answer — an independent implementation of the
format the reference reads with parquet-go (pqarrow/arrow.go:711-823). The decoded batch must be bit-identical (values and
NULLs), and it must be usable by the aggregate path like any imported batch.
"""
import io

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from frostdb_amd.logicalplan import Col, Count, DynCol, Sum, UInt64
from tests.parquet_util import row_group_chunks, write_parquet
from tests.test_gpu_parity import assert_same_result, run_oracle
from tests.util import arrow_to_pydict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import physicalplan
    assert physicalplan.device_count() >= 1
    return physicalplan


def prometheus_table(rng, n, null_frac=0.05, n_path=700):
    def lab(vals, nf):
        idx = rng.integers(0, len(vals), n)
        mask = rng.random(n) < nf
        return pa.array([None if m else vals[i] for i, m in zip(idx, mask)], type=pa.binary())
    # runs of repeated values (sorted-ish data compresses into RLE runs) next to random stretches (bit-packed runs)
    code_vals = [b"200", b"404", b"500"]
    code = np.repeat(rng.integers(0, 3, n // 50 + 1), 50)[:n]
    code_col = pa.array([None if rng.random() < null_frac / 5 else code_vals[c] for c in code], type=pa.binary())
    return pa.table({
        "labels.code": code_col,
        "labels.path": lab([b"/api/v1/p%04d" % i for i in range(n_path)], null_frac),
        "labels.req": pa.array([b"x"] * n, type=pa.binary()).cast(pa.binary()),
        "timestamp": pa.array(1_700_000_000_000 + np.arange(n, dtype=np.int64) * 15),
        "ivalue": pa.array(rng.integers(-10**12, 10**12, n), mask=rng.random(n) < null_frac),
        "value": pa.array(rng.uniform(0, 1000, n), mask=rng.random(n) < null_frac),
    }, schema=pa.schema([pa.field("labels.code", pa.binary()), pa.field("labels.path", pa.binary()), pa.field("labels.req", pa.binary(), nullable=False),
                         pa.field("timestamp", pa.int64(), nullable=False), pa.field("ivalue", pa.int64()), pa.field("value", pa.float64())]))


def decoded_equals_pyarrow(pp, data, rg=0):
    chunks, rows = row_group_chunks(data, rg)
    rb = pp.ResidentBatch.from_parquet(chunks, rows)
    return batch_equals_pyarrow(rb, data, rg, rows)


def batch_equals_pyarrow(rb, data, rg, rows):
    got = rb.to_arrow()
    want = pq.ParquetFile(io.BytesIO(data)).read_row_group(rg)
    assert got.num_rows == want.num_rows == rows
    assert got.schema.names == want.schema.names
    for name in want.schema.names:
        g, w = got.column(name), want.column(name).combine_chunks()
        if pa.types.is_dictionary(g.type):
            assert g.type.index_type == pa.uint32()
            g = g.dictionary_decode()
        if pa.types.is_dictionary(w.type):
            w = w.dictionary_decode()
        assert g.null_count == w.null_count, name
        if pa.types.is_floating(w.type) or pa.types.is_integer(w.type):
            gv, wv = g.to_numpy(zero_copy_only=False), w.to_numpy(zero_copy_only=False)
            ok = ~np.isnan(wv.astype(np.float64)) if w.null_count else np.ones(len(wv), bool)
            assert np.array_equal(np.asarray(gv)[ok].view(np.int64) if gv.dtype.kind == "f" else np.asarray(gv)[ok],
                                  np.asarray(wv)[ok].view(np.int64) if wv.dtype.kind == "f" else np.asarray(wv)[ok]), name
            assert np.array_equal(np.asarray(g.is_null()), np.asarray(w.is_null())), name
        elif pa.types.is_boolean(w.type):
            assert g.equals(w), name
        else:
            assert g.cast(pa.binary()).equals(w.cast(pa.binary())), name
    return rb, want


@pytest.mark.parametrize("version", ["1.0", "2.0"])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 65_537, 400_003])
def test_decoded_row_group_is_bit_identical_to_pyarrow(pp, n, version):
    rng = np.random.default_rng(n)
    data = write_parquet(prometheus_table(rng, n), data_page_version=version, data_page_size=4096 if n < 100_000 else 64 * 1024)
    rb, _ = decoded_equals_pyarrow(pp, data)
    rb.close()


@pytest.mark.parametrize("version", ["1.0", "2.0"])
@pytest.mark.parametrize("codec", ["SNAPPY", "GZIP", "ZSTD", "LZ4"])
def test_compressed_pages_decode_bit_identically(pp, codec, version):
    """Pages are inflated on the host into one image per chunk (V2: levels stay as they are, values are inflated behind them);
    run and page offsets then point into the image. Small pages → many pages per chunk, and a second row group."""
    rng = np.random.default_rng(len(codec))
    data = write_parquet(prometheus_table(rng, 120_001), compression=codec, data_page_version=version, data_page_size=16 * 1024, row_group_size=70_000)
    for rg in range(2):
        chunks, _ = row_group_chunks(data, rg)
        assert {c[5] for c in chunks} == {codec}
        rb, _ = decoded_equals_pyarrow(pp, data, rg)
        rb.close()


@pytest.mark.parametrize("version", ["1.0", "2.0"])
@pytest.mark.parametrize("n", [1, 2, 33, 129, 1025, 70_001, 300_000])
def test_delta_binary_packed_int64_columns(pp, n, version):
    """DELTA_BINARY_PACKED (the reference's default encoding for int64 fields of struct-tag schemas): regular timestamps (narrow
    deltas), a constant column (width 0), random int64 values (64-bit deltas that wrap), an optional column with NULLs (values are
    packed by rank) and an all-NULL one; several pages and, for the big case, two row groups; SNAPPY on top for one size."""
    rng = np.random.default_rng(n)
    t = pa.table({
        "labels.a": pa.array([b"v%d" % (i % 7) for i in range(n)], type=pa.binary()),
        "ts": pa.array(1_700_000_000_000 + np.cumsum(rng.integers(0, 30, n)).astype(np.int64)),
        "const": pa.array(np.full(n, -42, dtype=np.int64)),
        "wild": pa.array(rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)),
        "opt": pa.array(rng.integers(-10**9, 10**9, n), mask=rng.random(n) < 0.3),
        "none": pa.array([None] * n, type=pa.int64()),
        "value": pa.array(rng.normal(size=n)),
    }, schema=pa.schema([pa.field("labels.a", pa.binary()), pa.field("ts", pa.int64(), nullable=False), pa.field("const", pa.int64(), nullable=False),
                         pa.field("wild", pa.int64(), nullable=False), pa.field("opt", pa.int64()), pa.field("none", pa.int64()), pa.field("value", pa.float64())]))
    enc = {c: "DELTA_BINARY_PACKED" for c in ("ts", "const", "wild", "opt", "none")}
    data = write_parquet(t, use_dictionary=["labels.a"], column_encoding=enc, data_page_version=version, data_page_size=8 * 1024,
                         row_group_size=200_000, compression="SNAPPY" if n == 70_001 else "NONE")
    for rg in range(pq.ParquetFile(io.BytesIO(data)).metadata.num_row_groups):
        rb, _ = decoded_equals_pyarrow(pp, data, rg)
        rb.close()


@pytest.mark.parametrize("mode", ["no_dictionary", "fallback", "fallback_snappy_v2"])
def test_plain_byte_array_pages_are_dictionary_encoded_on_the_host(pp, mode):
    """BYTE_ARRAY pages that are PLAIN — a writer without dictionaries, or one whose dictionary outgrew its limit half way through
    the chunk (RLE_DICTIONARY pages first, PLAIN ones after) — become indices into the chunk's dictionary (its dictionary page plus
    the new values) on the host; the device reads them as a 32-bit-wide run. NULLs, empty strings, several row groups."""
    rng = np.random.default_rng(len(mode))
    n = 90_000
    t = pa.table({
        "labels.many": pa.array([None if i % 11 == 0 else b"" if i % 13 == 0 else b"value-%06d" % v for i, v in enumerate(rng.integers(0, 30_000, n))], type=pa.binary()),
        "labels.few": pa.array([b"k%d" % (i % 5) for i in range(n)], type=pa.binary()),
        "text": pa.array([None if i % 7 == 0 else "s%d" % v for i, v in enumerate(rng.integers(0, 2_000, n))], type=pa.string()),
        "value": pa.array(rng.normal(size=n)),
    })
    kw = dict(row_group_size=50_000, data_page_size=4096)
    if mode == "no_dictionary":
        kw.update(use_dictionary=False)
    else:
        kw.update(use_dictionary=["labels.many", "labels.few", "text"], dictionary_pagesize_limit=16 * 1024)
        if mode == "fallback_snappy_v2":
            kw.update(compression="SNAPPY", data_page_version="2.0")
    data = write_parquet(t, **kw)
    md = pq.ParquetFile(io.BytesIO(data)).metadata
    encs = set(md.row_group(0).column(0).encodings)
    assert "PLAIN" in encs and (mode == "no_dictionary" or "RLE_DICTIONARY" in encs), encs
    for rg in range(md.num_row_groups):
        rb, _ = decoded_equals_pyarrow(pp, data, rg)
        rb.close()


def test_several_row_groups_all_null_columns_and_wide_dictionaries(pp):
    """Row groups are decoded one by one; a column that is entirely NULL, a dictionary of 70 000 entries (17-bit indices) and a
    required column whose pages carry no definition levels."""
    rng = np.random.default_rng(5)
    n = 150_000
    t = pa.table({
        "labels.wide": pa.array([b"k%06d" % i for i in rng.integers(0, 70_000, n)], type=pa.binary()),
        "labels.none": pa.array([None] * n, type=pa.binary()),
        "labels.one": pa.array([b"only"] * n, type=pa.binary()),
        "value": pa.array(rng.normal(size=n)),
    })
    data = write_parquet(t, row_group_size=40_000, data_page_size=8192)
    n_rg = pq.ParquetFile(io.BytesIO(data)).metadata.num_row_groups
    assert n_rg == 4
    for rg in range(n_rg):
        rb, _ = decoded_equals_pyarrow(pp, data, rg)
        rb.close()


def _many_files():
    rng = np.random.default_rng(23)
    prom = prometheus_table(rng, 230_001)
    wide = pa.table({
        "labels.wide": pa.array([b"k%06d" % i for i in rng.integers(0, 70_000, 90_000)], type=pa.binary()),
        "labels.none": pa.array([None] * 90_000, type=pa.binary()),
        "flag": pa.array(rng.random(90_000) < 0.3, mask=rng.random(90_000) < 0.1),
        "value": pa.array(rng.normal(size=90_000)),
    })
    return {
        "plain_v1": lambda: write_parquet(prom, row_group_size=50_000, data_page_version="1.0"),
        "plain_v2_small_pages": lambda: write_parquet(prom, row_group_size=70_000, data_page_size=4096, data_page_version="2.0"),
        "snappy_v1": lambda: write_parquet(prom, row_group_size=50_000, compression="SNAPPY", data_page_version="1.0"),
        "zstd_v2_delta": lambda: write_parquet(prom, row_group_size=64_000, compression="ZSTD", data_page_version="2.0",
                                               column_encoding={"timestamp": "DELTA_BINARY_PACKED", "ivalue": "DELTA_BINARY_PACKED"},
                                               use_dictionary=["labels.code", "labels.path", "labels.req"]),
        "no_dictionary_gzip": lambda: write_parquet(prom.slice(0, 60_000), row_group_size=25_000, compression="GZIP", use_dictionary=False),
        "wide_and_all_null": lambda: write_parquet(wide, row_group_size=40_000, data_page_size=8192),
        "forty_small_row_groups": lambda: write_parquet(prom.slice(0, 40_000), row_group_size=1_000, compression="SNAPPY"),
    }


@pytest.mark.parametrize("which", ["plain_v1", "plain_v2_small_pages", "snappy_v1", "zstd_v2_delta", "no_dictionary_gzip", "wide_and_all_null", "forty_small_row_groups"])
def test_row_groups_decoded_by_one_call_are_bit_identical_to_pyarrow(pp, which):
    """fdb_batches_from_parquet: every row group of a file in ONE call (one copy queue, the host work of all of them side by side, a
    row group's kernels launched while later ones are still being parsed) — each batch bit-identical to pyarrow's reading of its row
    group, and to what the one-row-group call returns; the last row group is shorter than the others."""
    data = _many_files()[which]()
    n_rg = pq.ParquetFile(io.BytesIO(data)).metadata.num_row_groups
    assert n_rg >= 3
    groups = [row_group_chunks(data, rg) for rg in range(n_rg)]
    calls0 = pp.parquet_stats()["calls"]
    rbs = pp.ResidentBatch.from_parquet_many(groups)
    assert pp.parquet_stats()["calls"] == calls0 + 1 and len(rbs) == n_rg
    try:
        for rg, rb in enumerate(rbs):
            batch_equals_pyarrow(rb, data, rg, groups[rg][1])
        # a subset, in another order, is as good a call as the whole file
        some = pp.ResidentBatch.from_parquet_many([groups[n_rg - 1], groups[0]])
        try:
            batch_equals_pyarrow(some[0], data, n_rg - 1, groups[n_rg - 1][1])
            batch_equals_pyarrow(some[1], data, 0, groups[0][1])
        finally:
            for rb in some:
                rb.close()
    finally:
        for rb in rbs:
            rb.close()


def test_one_call_over_row_groups_reports_a_damaged_chunk_and_returns_nothing(pp):
    """A truncated chunk in the LAST row group of a call: the call fails with that chunk's error (the host threads of the other row
    groups are joined, the copies already queued are waited for), no batch comes back, and the next call works."""
    rng = np.random.default_rng(29)
    data = write_parquet(prometheus_table(rng, 120_000), row_group_size=30_000, compression="SNAPPY")
    groups = [row_group_chunks(data, rg) for rg in range(4)]
    chunks3, rows3 = groups[3]
    nm, ty, opt, u8, b, cd = chunks3[1]
    broken = list(chunks3)
    broken[1] = (nm, ty, opt, u8, bytes(b[: len(b) // 2]), cd)
    with pytest.raises(pp.FdbError) as e:
        pp.ResidentBatch.from_parquet_many(groups[:3] + [(broken, rows3)])
    assert e.value.code == pp.FDB_ERR_INVALID and "parquet" in str(e.value)
    # … and a corrupt Snappy stream (the page chain is intact, a page body is not) in the FIRST row group
    nm, ty, opt, u8, b, cd = groups[0][0][4]
    garbled = bytearray(b)
    for k in range(len(garbled) // 2, len(garbled) // 2 + 64):
        garbled[k] ^= 0xFF
    bad0 = list(groups[0][0])
    bad0[4] = (nm, ty, opt, u8, bytes(garbled), cd)
    with pytest.raises(pp.FdbError):
        pp.ResidentBatch.from_parquet_many([(bad0, groups[0][1])] + groups[1:])
    rbs = pp.ResidentBatch.from_parquet_many(groups)
    try:
        for rg, rb in enumerate(rbs):
            batch_equals_pyarrow(rb, data, rg, groups[rg][1])
    finally:
        for rb in rbs:
            rb.close()


def test_concurrent_calls_over_row_groups_do_not_disturb_each_other(pp):
    """Four threads (N scan chains reading parts), each decoding another file's row groups with one call at a time, three times over:
    the calls share the host pool's threads, the pinned pools and the device's streams — every batch still equals pyarrow's reading."""
    from concurrent.futures import ThreadPoolExecutor
    files = _many_files()
    names = ["plain_v1", "snappy_v1", "zstd_v2_delta", "wide_and_all_null"]
    datas = {nm: files[nm]() for nm in names}
    groups = {nm: [row_group_chunks(datas[nm], rg) for rg in range(pq.ParquetFile(io.BytesIO(datas[nm])).metadata.num_row_groups)] for nm in names}

    def work(nm):
        for _ in range(3):
            rbs = pp.ResidentBatch.from_parquet_many(groups[nm])
            try:
                for rg, rb in enumerate(rbs):
                    batch_equals_pyarrow(rb, datas[nm], rg, groups[nm][rg][1])
            finally:
                for rb in rbs:
                    rb.close()
        return nm

    with ThreadPoolExecutor(max_workers=4) as ex:
        assert sorted(ex.map(work, names)) == sorted(names)
    assert pp.live_allocations()["device_blocks"] == 0


def test_decoded_batches_feed_the_aggregate_like_imported_ones(pp):
    """Parquet bytes → resident batch → fused filter + aggregate, against the oracle run on pyarrow's reading of the same file."""
    rng = np.random.default_rng(11)
    data = write_parquet(prometheus_table(rng, 300_000), row_group_size=100_000)
    filt = Col("labels.code") == "200"
    aggs, groups = [Sum(Col("value")), Count(Col("value"))], [Col("labels.path")]
    plan = pp.HashAggregatePlan(filt, aggs, groups)
    keep, recs = [], []
    try:
        keep = pp.ResidentBatch.from_parquet_many([row_group_chunks(data, rg) for rg in range(3)])
        for rg in range(3):
            recs.append(pq.ParquetFile(io.BytesIO(data)).read_row_group(rg).to_batches()[0])
        plan.CallbackResident(keep)
        got = arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
        for k in keep:
            k.close()
    # (pyarrow reads strings as plain binary columns; the aggregate treats both representations alike)
    want = run_oracle(recs, filt, aggs, groups)
    assert_same_result(got, want, ["labels.path", "sum(value)", "count(value)"], float_cols={"sum(value)"})


def storage_layout_table(rng, n):
    """One column of every kind `storageLayoutToParquetNode` (dynparquet/schema.go:508-560) can produce and convert.go maps to Arrow:
    STRING, INT64, UINT64, BOOL, DOUBLE — nullable and required."""
    def strs(prefix, card, nf):
        return pa.array([None if rng.random() < nf else b"%s-%05d" % (prefix, rng.integers(0, card)) for _ in range(n)], type=pa.binary())
    return pa.table({
        "labels.host": strs(b"host", 40, 0.05),                                   # shared prefixes: what DELTA_BYTE_ARRAY is for
        "labels.pod": pa.array([b"pod-%07d" % i for i in rng.integers(0, max(n // 3, 1), n)], type=pa.binary()),  # high cardinality
        "flag": pa.array(rng.random(n) < 0.4, mask=rng.random(n) < 0.1),
        "ok": pa.array(rng.random(n) < 0.9),
        "seq": pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2) + np.uint64(1)),
        "timestamp": pa.array(1_700_000_000_000 + np.cumsum(rng.integers(0, 30, n)).astype(np.int64)),
        "value": pa.array(rng.uniform(0, 100, n), mask=rng.random(n) < 0.05),
    }, schema=pa.schema([pa.field("labels.host", pa.binary()), pa.field("labels.pod", pa.binary(), nullable=False), pa.field("flag", pa.bool_()),
                         pa.field("ok", pa.bool_(), nullable=False), pa.field("seq", pa.uint64(), nullable=False), pa.field("timestamp", pa.int64(), nullable=False),
                         pa.field("value", pa.float64())]))


LAYOUTS = [
    # (string encoding, codec, data page version): the encodings / compressions a FrostDB schema can name (schema.proto:54-86)
    ("DELTA_BYTE_ARRAY", "NONE", "1.0"), ("DELTA_LENGTH_BYTE_ARRAY", "SNAPPY", "2.0"), ("DELTA_BYTE_ARRAY", "BROTLI", "2.0"),
    ("PLAIN", "BROTLI", "1.0"), ("RLE_DICTIONARY", "LZ4", "2.0"),
]


@pytest.mark.parametrize("enc,codec,version", LAYOUTS)
@pytest.mark.parametrize("n", [1, 100, 9_999, 150_001])
def test_every_storage_layout_type_encoding_and_codec(pp, n, enc, codec, version):
    """BOOLEAN (PLAIN in V1 pages, RLE in V2), UINT64 (Int(64, unsigned) → uint64), strings in DELTA_BYTE_ARRAY /
    DELTA_LENGTH_BYTE_ARRAY / PLAIN / dictionary pages, BROTLI next to the other codecs: the decoded row group is bit-identical to
    pyarrow's reader, AND the aggregate over it equals the oracle's over pyarrow's record (the checker this repository pins on the
    reference's own vectors) — COUNT / SUM grouped by a string and a bool key, filtered on the uint64 column."""
    rng = np.random.default_rng(n + len(enc))
    t = storage_layout_table(rng, n)
    kw = dict(compression=codec, data_page_version=version, data_page_size=8 * 1024, row_group_size=100_000)
    if enc == "RLE_DICTIONARY":
        kw["use_dictionary"] = ["labels.host", "labels.pod"]
    else:
        kw.update(use_dictionary=False, column_encoding={"labels.host": enc, "labels.pod": enc, "timestamp": "DELTA_BINARY_PACKED"})
    data = write_parquet(t, **kw)
    n_rg = pq.ParquetFile(io.BytesIO(data)).metadata.num_row_groups
    keep, wants = [], []
    for rg in range(n_rg):
        chunks, _ = row_group_chunks(data, rg)
        assert [c[1] for c in chunks] == [6, 6, 0, 0, 2, 2, 5] and chunks[4][3] and not chunks[5][3]  # (seq carries the unsigned flag)
        rb, want = decoded_equals_pyarrow(pp, data, rg)
        got = rb.to_arrow()
        assert got.schema.field("flag").type == pa.bool_() and got.schema.field("seq").type == pa.uint64()
        assert got.column("flag").equals(want.column("flag").combine_chunks()) and got.column("ok").equals(want.column("ok").combine_chunks())
        keep.append(rb)
        wants += want.to_batches()
    filt = Col("seq") > UInt64(2**62)
    aggs, groups = [Count(Col("value")), Sum(Col("value")), Sum(Col("timestamp"))], [Col("labels.host"), Col("flag")]
    plan = pp.HashAggregatePlan(filt, aggs, groups)
    try:
        plan.CallbackResident(keep)
        got = arrow_to_pydict(plan.Finish())
    finally:
        plan.Close()
        for rb in keep:
            rb.close()
    want = run_oracle(wants, filt, aggs, groups)
    assert_same_result(got, want, ["labels.host", "flag", "count(value)", "sum(value)", "sum(timestamp)"], float_cols={"sum(value)"})


def test_repeated_and_unmapped_columns_are_refused_precisely(pp):
    """What pqarrow/convert/convert.go does not map to Arrow either — INT32 / FLOAT physical types — and repeated (list) columns,
    which it maps to lists that this path does not hold: FDB_ERR_UNSUPPORTED with the reason, nothing half-decoded."""
    n = 1000
    t = pa.table({"i32": pa.array(np.arange(n, dtype=np.int32)), "f32": pa.array(np.arange(n, dtype=np.float32)),
                  "lst": pa.array([[i] for i in range(n)], type=pa.list_(pa.int64()))})
    for name, why in (("i32", "only BOOLEAN, INT64, DOUBLE and BYTE_ARRAY"), ("f32", "only BOOLEAN, INT64, DOUBLE and BYTE_ARRAY"), ("lst", "nested / repeated")):
        chunks, rows = row_group_chunks(write_parquet(t.select([name])), 0)
        with pytest.raises(pp.FdbError) as e:
            pp.ResidentBatch.from_parquet(chunks, rows)
        assert e.value.code == pp.FDB_ERR_UNSUPPORTED and why in str(e.value), str(e.value)
    assert pp.live_allocations()["device_blocks"] == 0


@pytest.mark.parametrize("version", ["1.0", "2.0"])
def test_literal_snappy_pages_are_inflated_on_the_device(pp, version, monkeypatch):
    """SNAPPY pages of PLAIN float64 / int64 values that do not compress (random values: a page of literals, compressed size ≈ plain
    size) of ≥ 32 KiB are inflated by the device's block decoder, the rest (levels, dictionary indices, DELTA pages, small pages)
    on the host's threads as before: 1 MiB pages, optional columns with NULLs (V1: the definition levels sit INSIDE the compressed body:
    the host inflates just them) and required ones, two row groups, next to compressible columns in the same row group. Bit-identical
    to pyarrow's reader, and to what the all-host path (FDB_PARQUET_HOST_INFLATE) decodes."""
    rng = np.random.default_rng(77)
    n = 700_000
    t = pa.table({
        "labels.path": pa.array([b"/p%03d" % i for i in rng.integers(0, 300, n)], type=pa.binary()),
        "timestamp": pa.array(1_700_000_000_000 + np.arange(n, dtype=np.int64) * 15),          # compressible: stays on the host
        "noise_req": pa.array(rng.integers(-2**62, 2**62, n)),                                    # literals, required
        "noise_opt": pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1),          # literals, 10 % NULLs (bit-packed levels)
        "value": pa.array(rng.uniform(0, 1000, n), mask=rng.random(n) < 0.001),                   # literals, rare NULLs (RLE + bit-packed levels)
        "value_req": pa.array(rng.uniform(0, 1000, n)),
    }, schema=pa.schema([pa.field("labels.path", pa.binary()), pa.field("timestamp", pa.int64(), nullable=False), pa.field("noise_req", pa.int64(), nullable=False),
                         pa.field("noise_opt", pa.int64()), pa.field("value", pa.float64()), pa.field("value_req", pa.float64(), nullable=False)]))
    data = write_parquet(t, compression="SNAPPY", data_page_version=version, data_page_size=1 << 20, row_group_size=400_000)
    for rg in range(2):
        chunks, rows = row_group_chunks(data, rg)
        sizes = {c[0]: len(c[4]) for c in chunks}
        assert sizes["value_req"] > 0.9 * rows * 8 and sizes["noise_opt"] > 0.8 * rows * 8 and sizes["timestamp"] < 0.7 * rows * 8  # (the file is what the test thinks it is)
        monkeypatch.delenv("FDB_PARQUET_HOST_INFLATE", raising=False)
        rb, want = decoded_equals_pyarrow(pp, data, rg)
        dev = rb.to_arrow()
        rb.close()
        monkeypatch.setenv("FDB_PARQUET_HOST_INFLATE", "1")
        rb2, _ = decoded_equals_pyarrow(pp, data, rg)
        assert rb2.to_arrow().equals(dev)
        rb2.close()
    # a damaged literal page is still refused (by the device's decoder now): flip the page's length preamble
    monkeypatch.delenv("FDB_PARQUET_HOST_INFLATE", raising=False)
    chunks, rows = row_group_chunks(data, 0)
    bad = []
    for c in chunks:
        if c[0] == "value_req":
            b = bytearray(c[4])
            b[len(b) // 2] ^= 0xFF  # inside some page's literal … harmless for the decoder (a literal byte) but …
            b[40] ^= 0x7F           # … this one sits in the first page's header / preamble region
            c = (c[0], c[1], c[2], c[3], bytes(b), c[5])
        bad.append(c)
    try:
        rb = pp.ResidentBatch.from_parquet(bad, rows)
        got = rb.to_arrow().column("value_req").to_numpy(zero_copy_only=False)
        rb.close()
        # (if the flipped byte happened to be a literal too, the decode succeeds with different values: also fine — it must not crash)
        assert len(got) == rows
    except pp.FdbError:
        pass


def _thrift_page_header_v1(n_values: int, uncompressed: int, compressed: int) -> bytes:
    """PageHeader{1: type = DATA_PAGE, 2: uncompressed_page_size, 3: compressed_page_size, 5: DataPageHeader{1: num_values, 2: PLAIN, 3: RLE,
    4: RLE}} in Thrift's compact protocol (field header = delta << 4 | type; i32 = zigzag varint)."""
    def zz(v):
        v = (v << 1) ^ (v >> 31)
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80); v >>= 7
        out.append(v)
        return bytes(out)
    inner = b"\x15" + zz(n_values) + b"\x15" + zz(0) + b"\x15" + zz(3) + b"\x15" + zz(3) + b"\x00"
    return b"\x15" + zz(0) + b"\x15" + zz(uncompressed) + b"\x15" + zz(compressed) + b"\x2c" + inner + b"\x00"


@pytest.mark.parametrize("how", ["four_byte_offset", "two_byte_offset_past_the_ring"])
def test_snappy_pages_with_copies_from_far_back_take_the_host_path(pp, how, monkeypatch):
    """A legal Snappy page the device's decoder cannot take: a copy that reaches further back than the 64 KiB of output it keeps
    (offset > 65 472) — what a 4-byte-offset element does, or a block longer than 64 KiB as klauspost/compress's Snappy encoder writes
    them for parquet-go (go.mod). `plan_chunk` walks the page's tags and leaves such a page to the host's inflate; the row group
    loads, bit-identical to the values the page was built from (it used to be refused as corrupt)."""
    rng = np.random.default_rng(5)
    n = 60_000                                      # 480 000 bytes of PLAIN doubles: ≥ 256 KiB, and they do not compress
    vals = rng.uniform(0, 1000, n)
    far = 200_000 if how == "four_byte_offset" else 65_500
    raw = bytearray(vals.tobytes())
    at = 300_000
    raw[at:at + 64] = raw[at - far:at - far + 64]   # 64 bytes repeated from `far` bytes back
    want = np.frombuffer(bytes(raw), dtype=np.float64)

    def literal(b):
        out = bytearray()
        for i in range(0, len(b), 65536):
            piece = b[i:i + 65536]
            l = len(piece) - 1
            out += (bytes([l << 2]) if l < 60 else bytes([61 << 2]) + l.to_bytes(2, "little")) + piece
        return bytes(out)
    def varint(v):
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80); v >>= 7
        out.append(v)
        return bytes(out)
    copy = (bytes([(63 << 2) | 3]) + far.to_bytes(4, "little")) if how == "four_byte_offset" else (bytes([(63 << 2) | 2]) + far.to_bytes(2, "little"))
    body = varint(len(raw)) + literal(bytes(raw[:at])) + copy + literal(bytes(raw[at + 64:]))
    import pyarrow as _pa
    assert _pa.decompress(body, decompressed_size=len(raw), codec="snappy").to_pybytes() == bytes(raw)  # (the page is what the test thinks it is)
    chunk = _thrift_page_header_v1(n, len(raw), len(body)) + body
    for host in (False, True):
        if host:
            monkeypatch.setenv("FDB_PARQUET_HOST_INFLATE", "1")
        else:
            monkeypatch.delenv("FDB_PARQUET_HOST_INFLATE", raising=False)
        rb = pp.ResidentBatch.from_parquet([("value", 5, 0, False, chunk, "SNAPPY")], n)
        got = rb.to_arrow().column("value").to_numpy(zero_copy_only=False)
        rb.close()
        assert got.tobytes() == want.tobytes()
    assert pp.live_allocations()["device_blocks"] == 0
