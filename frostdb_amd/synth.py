"""Synthetic "Prometheus" Arrow data in the shape BASELINE.json's configs name.

Schema (README.md:76-82 of the reference): ``labels.* : dictionary<uint32, binary>`` nullable,
``timestamp : int64`` non-null, ``value : float64`` non-null. Counter-based generation: chunk ``c`` of a
shard uses ``numpy.random.Philox(key=seed, counter=c)``, so any rank can regenerate any chunk.

cfg 2/4:  labels.code ∈ {200: 70 %, 404: 10 %, 500: 8 %, 301: 6 %, 201: 4 %, 503: 2 %};
          labels.path: 1 024 values, Zipf s = 1.1; value ~ U[0, 1000); timestamp = t0 + 15 000·⌊i/S⌋.
cfg 3:    + labels.method (4 values), labels.instance (512 values, 5 % NULL).
Label columns carry validity bitmaps (0.1 % NULLs in code/path) so the bitmap bytes of the roofline
(16.25 B/row for cfg 2) are really read.
"""
from __future__ import annotations

from typing import Iterator, List

import numpy as np
import pyarrow as pa

SEED = 0xF057DB
CODES = [b"200", b"404", b"500", b"301", b"201", b"503"]
CODE_P = np.array([0.70, 0.10, 0.08, 0.06, 0.04, 0.02])
N_PATH = 1024
METHODS = [b"GET", b"POST", b"PUT", b"DELETE"]
N_INSTANCE = 512
T0 = 1_700_000_000_000
SERIES = 4096  # rows per scrape interval


def _zipf_cdf(n: int, s: float) -> np.ndarray:
    w = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** s
    c = np.cumsum(w)
    return c / c[-1]


_PATH_CDF = _zipf_cdf(N_PATH, 1.1)
_CODE_CDF = np.cumsum(CODE_P) / CODE_P.sum()
PATHS = [b"/api/v1/path/%04d" % i for i in range(N_PATH)]
INSTANCES = [b"10.0.%d.%d:9100" % (i // 256, i % 256) for i in range(N_INSTANCE)]


def _dict_col(idx: np.ndarray, values: List[bytes], null_mask) -> pa.DictionaryArray:
    ia = pa.array(idx, type=pa.uint32(), mask=null_mask)
    return pa.DictionaryArray.from_arrays(ia, pa.array(values, type=pa.binary()))


def prometheus_chunk(shard: int, chunk: int, rows: int, row_base: int = 0, cfg3: bool = False,
                     label_null_frac: float = 0.001, sorted_by_path: int = 0) -> pa.RecordBatch:
    """`sorted_by_path` = n > 0: chunk `chunk` of n of a table SORTED by labels.path (values ascending, NULLs last): its rows hold the
    chunk-th 1/n quantile slice of the path distribution, in order — the input an OrderedAggregate by labels.path is planned for. (The other
    columns are independent draws, so only the path column has to be put in order.)"""
    rng = np.random.Generator(np.random.Philox(key=SEED + shard, counter=[0, 0, 0, chunk]))
    u = rng.random(rows, dtype=np.float32)
    code = np.searchsorted(_CODE_CDF, u, side="right").astype(np.uint32)
    np.minimum(code, len(CODES) - 1, out=code)
    u = rng.random(rows)
    if sorted_by_path:
        u = (chunk + u) / sorted_by_path
    path = np.searchsorted(_PATH_CDF, u, side="right").astype(np.uint32)
    np.minimum(path, N_PATH - 1, out=path)
    value = rng.random(rows) * 1000.0
    ts = T0 + 15_000 * ((row_base + np.arange(rows, dtype=np.int64)) // SERIES)
    path_null = rng.random(rows, dtype=np.float32) < label_null_frac
    if sorted_by_path:
        path.sort()
        k = int(path_null.sum()) if chunk == sorted_by_path - 1 else 0  # the table's NULL paths are its last rows
        path_null = np.arange(rows) >= rows - k
    arrays = [_dict_col(code, CODES, rng.random(rows, dtype=np.float32) < label_null_frac),
              _dict_col(path, PATHS, path_null)]
    names = ["labels.code", "labels.path"]
    if cfg3:
        arrays.append(_dict_col(rng.integers(0, len(METHODS), size=rows, dtype=np.uint32), METHODS,
                                rng.random(rows, dtype=np.float32) < label_null_frac))
        names.append("labels.method")
        arrays.append(_dict_col(rng.integers(0, N_INSTANCE, size=rows, dtype=np.uint32), INSTANCES,
                                rng.random(rows, dtype=np.float32) < 0.05))
        names.append("labels.instance")
    arrays += [pa.array(ts), pa.array(value)]
    names += ["timestamp", "value"]
    return pa.RecordBatch.from_arrays(arrays, names=names)


def prometheus_batches(shard: int, total_rows: int, batch_rows: int, cfg3: bool = False) -> Iterator[pa.RecordBatch]:
    done, chunk = 0, 0
    while done < total_rows:
        n = min(batch_rows, total_rows - done)
        yield prometheus_chunk(shard, chunk, n, row_base=done, cfg3=cfg3)
        done += n
        chunk += 1


# ---- cfg 5: 32 dynamic label columns, 10 M distinct groups -----------------------------------------------------------
CFG5_COLS = 32
CFG5_CARD = 4


def _cfg5_tables(n_groups: int):
    """Per-group label digits: group g ↦ 32 digits in [0, 4); columns 0-11 are the base-4 digits of g (so distinct g give
    distinct tuples), columns 12-31 a hash of (g, column); ≈3 % of the digits are NULL (a function of (g, column) only)."""
    g = np.arange(n_groups, dtype=np.uint64)
    digits = np.empty((CFG5_COLS, n_groups), dtype=np.uint8)
    nulls = np.empty((CFG5_COLS, n_groups), dtype=bool)
    for c in range(CFG5_COLS):
        h = (g + np.uint64(c + 1)) * np.uint64(0x9E3779B97F4A7C15)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        digits[c] = ((g >> np.uint64(2 * c)) & np.uint64(3)) if c < 12 else (h & np.uint64(3))
        nulls[c] = ((h >> np.uint64(8)) % np.uint64(100)) < 3 if c >= 12 else False  # keep the identifying digits non-NULL
    return digits, nulls


_CFG5_CACHE = {}


def cfg5_group_ids(shard: int, chunk: int, rows: int, n_groups: int = 10_000_000) -> np.ndarray:
    """The group id of every row of `cfg5_chunk(shard, chunk, rows, n_groups)` (its first draw from the chunk's generator): lets a
    test compute the expected per-group answer with numpy, without a group-by over 32 string columns."""
    rng = np.random.Generator(np.random.Philox(key=SEED + 5 + shard, counter=[0, 0, 0, chunk]))
    return rng.integers(0, n_groups, size=rows, dtype=np.int64)


def cfg5_decode_group_ids(batch: pa.RecordBatch) -> np.ndarray:
    """Group id of every row of a record that carries cfg 5's label columns (input or result): columns 0-11 hold the base-4
    digits of the id as the dictionary values b"lCC=D" (wide dictionaries, `card` > 4: b"lCC=EEEEE" with digit = E // (card // 4))."""
    gid = np.zeros(batch.num_rows, dtype=np.int64)
    for c in range(12):
        col = batch.column(batch.schema.get_field_index("labels.l%02d" % c))
        assert col.null_count == 0
        vals = col.dictionary.to_pylist()
        wide = len(vals[0].rsplit(b"=", 1)[1]) > 1
        step = _cfg5_card(c, _CFG5_WIDE_CARDS) // CFG5_CARD if wide else 1
        digit_of_entry = np.array([int(v.rsplit(b"=", 1)[1]) // step for v in vals], dtype=np.int64)
        gid |= digit_of_entry[col.indices.to_numpy(zero_copy_only=False).astype(np.int64)] << (2 * c)
    return gid


# Wide dictionaries for cfg 5 (the table-free OrderedAggregate's wide run records, key ids beyond one byte): column c's dictionary has
# cards[c % len(cards)] entries b"lCC=EEEEE", of which the four entries digit × step + (37 c mod step) are used (step = card // 4) —
# monotone in the digit, so the table's sort order is the same as with the 4-entry dictionaries.
_CFG5_WIDE_CARDS = (512, 1024, 4096, 65532)  # (65 534 is the most a two-byte key id holds: 0 = NULL; a multiple of 4 for the digit step)


def _cfg5_card(c: int, cards) -> int:
    return int(cards[c % len(cards)])


def _rev4_12(x: np.ndarray) -> np.ndarray:
    """The 12 base-4 digits of x in reverse order: group id ↔ its rank in the lexicographic order of (labels.l00, …, labels.l11) —
    column c holds digit c of the id, least significant first, and the dictionary values "lCC=0" < … < "lCC=3" sort like the digits."""
    out = np.zeros_like(x)
    for i in range(12):
        out |= ((x >> (2 * i)) & 3) << (2 * (11 - i))
    return out


_CFG5_DICTS = {}


def cfg5_chunk(shard: int, chunk: int, rows: int, n_groups: int = 10_000_000, sorted_rows: bool = False, of_chunks: int = 0, wide_dicts: bool = False) -> pa.RecordBatch:
    """`sorted_rows`: the record's rows ordered by group id — what a scan of a table SORTED by its label columns (FrostDB's sorting
    columns) hands the aggregate: rows of one group arrive next to each other.
    `of_chunks` > 0 (with sorted_rows): the TABLE is sorted, not just each record — rows are ordered by (labels.l00, labels.l01, …)
    with dictionary values ascending, and chunk i of `of_chunks` holds the i-th slice of the key space, so the records of a scan
    arrive in key order one after the other: the input an OrderedAggregate is planned for (physicalplan.go:525-560)."""
    if n_groups not in _CFG5_CACHE:
        _CFG5_CACHE[n_groups] = _cfg5_tables(n_groups)
    digits, nulls = _CFG5_CACHE[n_groups]
    rng = np.random.Generator(np.random.Philox(key=SEED + 5 + shard, counter=[0, 0, 0, chunk]))
    if sorted_rows and of_chunks > 0:
        assert n_groups <= 4 ** 12
        lo, hi = (4 ** 12) * chunk // of_chunks, (4 ** 12) * (chunk + 1) // of_chunks
        parts, have = [], 0
        while have < rows:  # ranks whose group id is < n_groups (ids are the digit-reversed ranks)
            r = rng.integers(lo, hi, size=max(1024, int((rows - have) * 2.2)), dtype=np.int64)
            r = r[_rev4_12(r) < n_groups]
            parts.append(r)
            have += len(r)
        r = np.concatenate(parts)[:rows]
        r.sort()
        gid = _rev4_12(r)
    else:
        gid = rng.integers(0, n_groups, size=rows, dtype=np.int64)
        if sorted_rows:
            gid.sort()
    arrays, names = [], []
    for c in range(CFG5_COLS):
        d = digits[c][gid].astype(np.uint32)
        if wide_dicts:
            card = _cfg5_card(c, _CFG5_WIDE_CARDS)
            step = card // CFG5_CARD
            d = d * np.uint32(step) + np.uint32((37 * c) % step)
            if (c, card) not in _CFG5_DICTS:
                _CFG5_DICTS[(c, card)] = pa.array([b"l%02d=%05d" % (c, k) for k in range(card)], type=pa.binary())
            values = _CFG5_DICTS[(c, card)]
        else:
            values = pa.array([b"l%02d=%d" % (c, k) for k in range(CFG5_CARD)], type=pa.binary())
        idx = pa.array(d, type=pa.uint32(), mask=nulls[c][gid] if c >= 12 else None)
        arrays.append(pa.DictionaryArray.from_arrays(idx, values))
        names.append("labels.l%02d" % c)
    arrays.append(pa.array(rng.random(rows) * 1000.0))
    names.append("value")
    return pa.RecordBatch.from_arrays(arrays, names=names)
