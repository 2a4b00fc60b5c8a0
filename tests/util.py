"""Helpers shared by the test-suite: build Arrow records shaped like the reference's test schemas."""
from __future__ import annotations

from typing import Any, Dict, List, Sequence

import numpy as np
import pyarrow as pa

DICT_TYPE = pa.dictionary(pa.uint32(), pa.binary())


def dict_array(values: Sequence[Any], dict_type=DICT_TYPE) -> pa.DictionaryArray:
    """Dictionary-encode `values` (bytes/str or None) with a first-seen dictionary, uint32 indices."""
    entries: Dict[bytes, int] = {}
    idx: List[Any] = []
    for v in values:
        if v is None:
            idx.append(None)
            continue
        b = v.encode() if isinstance(v, str) else bytes(v)
        idx.append(entries.setdefault(b, len(entries)))
    vt = dict_type.value_type
    dvals = list(entries.keys())
    if pa.types.is_string(vt) or pa.types.is_large_string(vt):
        dvals = [d.decode() for d in dvals]
    return pa.DictionaryArray.from_arrays(pa.array(idx, type=dict_type.index_type), pa.array(dvals, type=vt))


def column_type(name: str):
    if name.startswith("labels.") or name in ("stacktrace",):
        return DICT_TYPE
    if name == "floatvalue":
        return pa.float64()
    return pa.int64()


def parse_rows(cols: Sequence[str], text: str) -> List[List[Any]]:
    rows = []
    for line in text.strip().splitlines():
        toks = line.split()
        if not toks:
            continue
        assert len(toks) == len(cols), (cols, toks)
        row = []
        for c, t in zip(cols, toks):
            ty = column_type(c)
            if t == "null":
                row.append(None)
            elif ty == DICT_TYPE:
                row.append(t.encode())
            elif ty == pa.float64():
                row.append(float(t))
            else:
                row.append(int(t))
        rows.append(row)
    return rows


def record_from_rows(cols: Sequence[str], rows: List[List[Any]]) -> pa.RecordBatch:
    arrays = []
    for ci, c in enumerate(cols):
        vals = [r[ci] for r in rows]
        ty = column_type(c)
        arrays.append(dict_array(vals) if ty == DICT_TYPE else pa.array(vals, type=ty))
    return pa.RecordBatch.from_arrays(arrays, names=list(cols))


def bytes_schema_record(table: Dict[str, Any]) -> pa.RecordBatch:
    """A record of logictest's `bytes` schema (logic_test.go:110-146): labels.* dictionary strings, timestamp UINT64, value a plain
    binary column (its DELTA_LENGTH_BYTE_ARRAY storage becomes Arrow binary, pqarrow/convert/convert.go:64-70)."""
    arrays = []
    for ci, c in enumerate(table["cols"]):
        vals = [r[ci] for r in table["rows"]]
        if c.startswith("labels."):
            arrays.append(dict_array(vals))
        elif c == "timestamp":
            arrays.append(pa.array(vals, type=pa.uint64()))
        else:
            arrays.append(pa.array(vals, type=pa.binary()))
    return pa.RecordBatch.from_arrays(arrays, names=list(table["cols"]))


def table_records(table: Dict[str, Any]) -> List[pa.RecordBatch]:
    """One Arrow record per `insert` (an L0 part reaches the scan as a whole record, index/lsm.go:420-427)."""
    return [record_from_rows(table["cols"], parse_rows(table["cols"], text)) for text in table["inserts"]]


def batch_rows(d: Dict[str, List[Any]], out_cols: Sequence[str]) -> List[tuple]:
    n = len(next(iter(d.values()))) if d else 0
    return [tuple(d[c][i] if c in d else None for c in out_cols) for i in range(n)]


def arrow_to_pydict(batch: pa.RecordBatch) -> Dict[str, List[Any]]:
    """RecordBatch → {name: python values} with bytes for string-like values (matches OracleBatch.to_pydict)."""
    out: Dict[str, List[Any]] = {}
    for name, col in zip(batch.schema.names, batch.columns):
        if pa.types.is_dictionary(col.type):
            col = col.dictionary_decode()
        vals = col.to_pylist()
        vals = [v.encode() if isinstance(v, str) else v for v in vals]
        out[name] = vals
    return out


def fmt(v: Any) -> Any:
    """logictest prints floats with %f (logictest/runner.go:438)."""
    return "%f" % v if isinstance(v, float) else v


def sort_key(row: tuple):
    return tuple((x is None, "" if x is None else repr(x)) for x in row)


def make_prometheus_batch(rng: np.random.Generator, n: int, n_path: int = 64, null_frac: float = 0.02,
                          with_method: bool = True) -> pa.RecordBatch:
    """A small synthetic batch in the shape of BASELINE.json's Prometheus schema."""
    codes = [b"200", b"404", b"500", b"301", b"201", b"503"]
    p = np.array([0.70, 0.10, 0.08, 0.06, 0.04, 0.02])
    code_idx = rng.choice(len(codes), size=n, p=p).astype(np.uint32)
    path_idx = rng.integers(0, n_path, size=n).astype(np.uint32)
    value = rng.uniform(0, 1000, size=n)
    ts = (1_700_000_000_000 + 15_000 * (np.arange(n) // 7)).astype(np.int64)

    def darr(idx, names, nf):
        mask = rng.random(n) < nf if nf > 0 else None
        ia = pa.array(idx, type=pa.uint32(), mask=mask)
        return pa.DictionaryArray.from_arrays(ia, pa.array(names, type=pa.binary()))

    arrays = [darr(code_idx, codes, null_frac), darr(path_idx, [b"/api/v1/p%04d" % i for i in range(n_path)], null_frac)]
    names = ["labels.code", "labels.path"]
    if with_method:
        methods = [b"GET", b"POST", b"PUT", b"DELETE"]
        arrays.append(darr(rng.integers(0, 4, size=n).astype(np.uint32), methods, 0.0))
        names.append("labels.method")
        arrays.append(darr(rng.integers(0, 16, size=n).astype(np.uint32), [b"inst-%02d" % i for i in range(16)], 0.05))
        names.append("labels.instance")
    arrays += [pa.array(ts), pa.array(value)]
    names += ["timestamp", "value"]
    return pa.RecordBatch.from_arrays(arrays, names=names)


def make_simple_batches(rng, total_rows=10_000, n_records=3):
    """BASELINE.json config 1: the `examples/simple` schema (examples/simple/simple.go:24-27) — dynamic `names.*` label columns
    (the struct-tag ingest path yields dictionary<uint32, utf8>, internal/records/record_builder.go:510-524) + `value int64`.
    Like the example, only some records carry `names.middle_name` (schema drift between records)."""
    firsts = ["Frederic", "Thor", "Matthias", "Ada", "Grace"]
    surnames = ["Brancz", "Hansen", "Loibl", "Lovelace", "Hopper", "Ritchie", "Pike"]
    middles = ["Oliver Rainer", "B.", "M."]
    utf8_dict = pa.dictionary(pa.uint32(), pa.string())
    out = []
    per = total_rows // n_records
    for r in range(n_records):
        n = per if r < n_records - 1 else total_rows - per * (n_records - 1)
        cols = {
            "names.first_name": dict_array([firsts[i] for i in rng.integers(0, len(firsts), size=n)], utf8_dict),
            "names.surname": dict_array([surnames[i] for i in rng.integers(0, len(surnames), size=n)], utf8_dict),
        }
        if r % 2 == 1:
            cols["names.middle_name"] = dict_array([None if rng.random() < 0.5 else middles[rng.integers(0, len(middles))] for _ in range(n)], utf8_dict)
        cols["value"] = pa.array(rng.integers(90, 110, size=n), type=pa.int64())
        out.append(pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys())))
    return out
