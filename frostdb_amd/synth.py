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
                     label_null_frac: float = 0.001) -> pa.RecordBatch:
    rng = np.random.Generator(np.random.Philox(key=SEED + shard, counter=[0, 0, 0, chunk]))
    u = rng.random(rows, dtype=np.float32)
    code = np.searchsorted(_CODE_CDF, u, side="right").astype(np.uint32)
    np.minimum(code, len(CODES) - 1, out=code)
    u = rng.random(rows)
    path = np.searchsorted(_PATH_CDF, u, side="right").astype(np.uint32)
    np.minimum(path, N_PATH - 1, out=path)
    value = rng.random(rows) * 1000.0
    ts = T0 + 15_000 * ((row_base + np.arange(rows, dtype=np.int64)) // SERIES)
    arrays = [_dict_col(code, CODES, rng.random(rows, dtype=np.float32) < label_null_frac),
              _dict_col(path, PATHS, rng.random(rows, dtype=np.float32) < label_null_frac)]
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
