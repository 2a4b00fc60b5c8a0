"""Fuzzes the HOST side of fdb_batch_from_parquet (page headers, run / delta tables, page inflation, PLAIN byte-array pages) with
mutated column chunks under AddressSanitizer + UBSan: python tools/asan_parquet_run.py [mutations per variant] [seed] — started by
tools/asan_parquet.sh, which builds the instrumented library. Every call must come back with an error code (1 invalid, 2 unsupported,
4 = parsed fine, no device); a sanitizer report or a hang is a bug."""
import os, sys, io, random, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
from tests.parquet_util import row_group_chunks, write_parquet
lib = ctypes.CDLL(os.environ['FDB_ASAN_LIB'])
class ParquetChunk(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("physical_type", ctypes.c_int32), ("optional", ctypes.c_int32), ("utf8", ctypes.c_int32),
                ("codec", ctypes.c_int32), ("data", ctypes.c_void_p), ("n_bytes", ctypes.c_int64)]
CODECS = {"UNCOMPRESSED": 0, "SNAPPY": 1, "GZIP": 2, "LZO": 3, "BROTLI": 4, "LZ4": 5, "ZSTD": 6, "LZ4_RAW": 7}
lib.fdb_batch_from_parquet.restype = ctypes.c_int
lib.fdb_last_error.restype = ctypes.c_char_p
def call(chunks, rows):
    arr = (ParquetChunk * len(chunks))(); keep = []
    for i, (nm, ty, opt, u8, data, cd) in enumerate(chunks):
        # exact-size heap copy so that ASan sees reads past the chunk's end
        buf = (ctypes.c_ubyte * len(data)).from_buffer_copy(data) if len(data) else (ctypes.c_ubyte * 1)()
        keep.append(buf); n = nm.encode(); keep.append(n)
        arr[i] = ParquetChunk(n, ty, opt, 1 if u8 else 0, CODECS[cd], ctypes.addressof(buf), len(data))
    out = ctypes.c_void_p()
    return lib.fdb_batch_from_parquet(arr, len(chunks), ctypes.c_int64(rows), 0, ctypes.byref(out))
class ParquetRowGroup(ctypes.Structure):
    _fields_ = [("chunks", ctypes.c_void_p), ("n_chunks", ctypes.c_int32), ("n_rows", ctypes.c_int64)]
lib.fdb_batches_from_parquet.restype = ctypes.c_int
def call_many(groups):
    """fdb_batches_from_parquet: the row groups' chunks are planned, inflated and parsed side by side on the pool's threads"""
    rgs = (ParquetRowGroup * len(groups))(); keep = []
    for g, (chunks, rows) in enumerate(groups):
        arr = (ParquetChunk * len(chunks))()
        for i, (nm, ty, opt, u8, data, cd) in enumerate(chunks):
            buf = (ctypes.c_ubyte * len(data)).from_buffer_copy(data) if len(data) else (ctypes.c_ubyte * 1)()
            keep.append(buf); n = nm.encode(); keep.append(n)
            arr[i] = ParquetChunk(n, ty, opt, 1 if u8 else 0, CODECS[cd], ctypes.addressof(buf), len(data))
        keep.append(arr)
        rgs[g] = ParquetRowGroup(ctypes.addressof(arr), len(chunks), rows)
    outs = (ctypes.c_void_p * len(groups))()
    return lib.fdb_batches_from_parquet(rgs, len(groups), 0, outs)
rng = np.random.default_rng(7); random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
n = 3000
t = pa.table({"labels.a": pa.array([None if i % 9 == 0 else b"v%d" % (i % 13) for i in range(n)], type=pa.binary()),
              "labels.b": pa.array([b"w%d" % v for v in rng.integers(0, 900, n)], type=pa.binary()),
              "ts": pa.array(np.cumsum(rng.integers(0, 50, n)).astype(np.int64)),
              "opt": pa.array(rng.integers(-10**9, 10**9, n), mask=rng.random(n) < 0.2),
              "flag": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1), "seq": pa.array(rng.integers(0, 2**62, n).astype(np.uint64)),
              "value": pa.array(rng.random(n), mask=rng.random(n) < 0.1)})
variants = [dict(), dict(compression="SNAPPY"), dict(compression="GZIP"), dict(compression="ZSTD"), dict(compression="LZ4"),
            dict(use_dictionary=["labels.a", "labels.b"], column_encoding={"ts": "DELTA_BINARY_PACKED", "opt": "DELTA_BINARY_PACKED"}),
            dict(use_dictionary=False), dict(use_dictionary=False, compression="SNAPPY", data_page_version="2.0"),
            dict(data_page_version="2.0", compression="ZSTD"), dict(compression="BROTLI"),
            dict(use_dictionary=False, column_encoding={"labels.a": "DELTA_BYTE_ARRAY", "labels.b": "DELTA_LENGTH_BYTE_ARRAY", "ts": "DELTA_BINARY_PACKED"}),
            dict(use_dictionary=False, column_encoding={"labels.a": "DELTA_LENGTH_BYTE_ARRAY", "labels.b": "DELTA_BYTE_ARRAY"}, compression="SNAPPY", data_page_version="2.0")]
# pages of literals, big enough for the device's Snappy decoder (fdb_parquet.cpp plan_chunk: ≥ 256 KiB, compressed ≥ 0.9 × plain): the
# host prefix-decodes a V1 page's definition levels out of the compressed body (snappy_prefix) and leaves the values to the device
nb = 70_000
t_big = pa.table({"noise": pa.array(rng.integers(-2**62, 2**62, nb), mask=rng.random(nb) < 0.1), "value": pa.array(rng.random(nb), mask=rng.random(nb) < 0.001),
                  "req": pa.array(rng.random(nb))}, schema=pa.schema([pa.field("noise", pa.int64()), pa.field("value", pa.float64()), pa.field("req", pa.float64(), nullable=False)]))
variants += [dict(compression="SNAPPY", big=True), dict(compression="SNAPPY", data_page_version="2.0", big=True)]
codes = {}; total = 0
for kw in variants:
    kw = dict(kw); big = kw.pop("big", False)
    data = write_parquet(t_big if big else t, data_page_size=(512 << 10) if big else 2048, **kw)
    chunks, rows = row_group_chunks(data, 0)
    rc = call(chunks, rows); codes[rc] = codes.get(rc, 0) + 1   # unmutated: parses, then fails at the device
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
        # one chunk at a time, so that most runs get past the other columns
        victim = random.randrange(len(chunks))
        mut = []
        for ci, (nm, ty, opt, u8, b, cd) in enumerate(chunks):
            b = bytearray(b)
            if ci == victim:
                r = random.random()
                if r < 0.4:
                    for _ in range(random.randint(1, 6)):
                        b[random.randrange(len(b))] = random.randrange(256)
                elif r < 0.6:
                    b = b[: random.randrange(1, len(b))]
                elif r < 0.75:
                    k = random.randrange(len(b)); b[k:k] = bytes(random.randrange(256) for _ in range(random.randint(1, 40)))
                elif r < 0.82:
                    k = random.randrange(len(b)); l = random.randint(1, 64); b[k:k + l] = b"\xff" * l
                elif r < 0.9:
                    # a huge varint (2^56 … 2^63, as a run / block / collection header would carry it) written over a random spot
                    k = random.randrange(len(b)); v = (1 << random.randint(56, 63)) | random.getrandbits(8); enc = bytearray()
                    while v >= 0x80: enc.append((v & 0x7F) | 0x80); v >>= 7
                    enc.append(v); b[k:k + len(enc)] = enc
                else:
                    cd = random.choice(["SNAPPY", "UNCOMPRESSED", "ZSTD", "GZIP", "LZ4"]); opt = random.choice([0, 1])
            mut.append((nm, ty, opt, u8, bytes(b), cd))
        import time as _t; _t0 = _t.time()
        n_rows = random.choice([rows, rows, rows, rows - 1, rows + 5])
        if it % 4 == 3:   # the damaged row group between intact ones, one call
            order = [(chunks, rows), (mut, n_rows), (chunks, rows)]
            random.shuffle(order)
            rc = call_many(order)
        else:
            rc = call(mut, n_rows)
        _dt = _t.time() - _t0
        if _dt > 2: print("slow call", round(_dt, 1), "s variant", kw, "victim", chunks[victim][0], "rc", rc, flush=True)
        codes[rc] = codes.get(rc, 0) + 1; total += 1
print("variants", len(variants), "runs", total, "return codes", codes)
