import ctypes, sys
sys.path.insert(0, '/root/repo')
import numpy as np, pyarrow as pa
from frostdb_amd.physicalplan import ExportedBatch, ArrowArray, ArrowSchema, import_batch
L = ctypes.CDLL('/tmp/asan/libfdb_asan.so')
vp = ctypes.c_void_p
L.fdb_arrow_roundtrip.argtypes = [vp, vp, vp, vp]
L.fdb_last_error.restype = ctypes.c_char_p
def rt(rec):
    arr, sch = ArrowArray(), ArrowSchema()
    with ExportedBatch(rec) as ex:
        rc = L.fdb_arrow_roundtrip(ctypes.addressof(ex.array), ctypes.addressof(ex.schema), ctypes.addressof(arr), ctypes.addressof(sch))
    if rc != 0:
        raise RuntimeError(L.fdb_last_error().decode())
    return import_batch(arr, sch)
rng = np.random.default_rng(5)
words = ["", "a", "é", "abc\x00d", "zeta"] + ["w%d" % k for k in range(50)]
for n in (0, 1, 7, 8, 9, 63, 64, 65, 1003, 50_000):
    def strs(typ, nf): return pa.array([words[k] for k in rng.integers(0, len(words), n)], type=pa.string(), mask=rng.random(n) < nf).cast(typ)
    def dic(it, vt, nf):
        idx = pa.array(rng.integers(0, 7, n), type=it, mask=rng.random(n) < nf)
        return pa.DictionaryArray.from_arrays(idx, pa.array(["v%d" % k for k in range(7)], type=pa.string()).cast(vt))
    cols = {"i": pa.array(rng.integers(-9, 9, n), type=pa.int64(), mask=rng.random(n) < 0.1), "u": pa.array(rng.integers(0, 9, n).astype(np.uint64)),
            "f": pa.array(rng.normal(size=n), mask=rng.random(n) < 0.5), "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2),
            "s": strs(pa.string(), 0.1), "z": strs(pa.binary(), 0.0), "S": strs(pa.large_string(), 0.3), "Z": strs(pa.large_binary(), 1.0),
            "d8": dic(pa.int8(), pa.binary(), 0.1), "d16": dic(pa.uint16(), pa.string(), 0.0), "d32": dic(pa.uint32(), pa.large_string(), 0.2), "d64": dic(pa.int64(), pa.large_binary(), 0.05)}
    full = pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))
    for rec in (full, full.slice(0, 0), full.slice(n // 3, n // 2), full.slice(max(n - 2, 0), 2)):
        out = rt(rec)
        for name in rec.schema.names:
            a, b = rec.column(name), out.column(name)
            if pa.types.is_dictionary(a.type):
                a, b = a.dictionary_decode().cast(pa.large_binary()), b.dictionary_decode().cast(pa.large_binary())
            assert a.to_pylist() == b.to_pylist(), (n, name)
        del out
print("asan roundtrip ok")
