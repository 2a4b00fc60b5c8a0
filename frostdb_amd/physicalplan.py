"""ctypes binding of the C ABI (include/frostdb_amd.h), shaped like the reference's push operators.

``HashAggregatePlan`` stands where one chain ``PredicateFilter → HashAggregate(final=false)`` stands in
``physicalplan.Build`` (query/physicalplan/physicalplan.go:417-474) and offers the same five verbs as
``PhysicalPlan`` (physicalplan.go:24-30): ``Callback(record)``, ``Finish()``, ``SetNext(next)``, ``Draw()``,
``Close()``. pyarrow plays the role arrow-go's ``cdata`` package plays in the Go shim.

There is NO CPU fallback: if ``libfrostdb_amd.so`` is missing, or the HIP runtime reports an error, calls raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Callable, List, Optional, Sequence

import pyarrow as pa

from .arrow_c import ArrowArray, ArrowSchema, ExportedBatch, import_batch
from .logicalplan import AggregationFunction, Column, Expr, to_desc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfrostdb_amd.so")

FDB_OK, FDB_ERR_INVALID, FDB_ERR_UNSUPPORTED, FDB_ERR_NOT_FOUND, FDB_ERR_DEVICE, FDB_ERR_OOM, FDB_ERR_STATE = range(7)


class FdbError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"frostdb_amd error {code}: {msg}")
        self.code = code
        self.msg = msg


class UnsupportedError(FdbError):
    """≙ ErrUnsupportedBooleanExpression / ErrUnsupportedBinaryOperation / ErrUnsupportedSumType."""


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Loads libfrostdb_amd.so (built in-tree by frostdb_amd.build / __graft_entry__.build). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("FDB_LIB_PATH") or LIB_PATH  # (FDB_LIB_PATH: an instrumented build of the same library — tools/asan_gpu.sh)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m frostdb_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(path)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    P = ctypes.POINTER
    L.fdb_version.restype = ctypes.c_char_p
    L.fdb_last_error.restype = ctypes.c_char_p
    L.fdb_device_count.argtypes = [P(ctypes.c_int)]
    L.fdb_plan_create.argtypes = [vp, ctypes.c_int, P(vp)]
    L.fdb_plan_push.argtypes = [vp, vp, vp]
    L.fdb_plan_push_batch.argtypes = [vp, vp]
    L.fdb_plan_push_batches.argtypes = [vp, P(vp), i32]
    L.fdb_plan_finish.argtypes = [vp, vp, vp, P(i64)]
    L.fdb_plan_finish_next.argtypes = [vp, vp, vp, P(i64), P(i32)]
    L.fdb_plan_merge.argtypes = [vp, vp]
    L.fdb_plan_filter.argtypes = [vp, vp, vp, vp, vp, P(i64)]
    L.fdb_plan_select.argtypes = [vp, vp, vp, vp, i64, P(i64)]
    L.fdb_plan_draw.restype = ctypes.c_char_p
    L.fdb_plan_draw.argtypes = [vp]
    L.fdb_plan_last_error.restype = ctypes.c_char_p
    L.fdb_plan_last_error.argtypes = [vp]
    L.fdb_plan_close.argtypes = [vp]
    L.fdb_plan_close.restype = None
    L.fdb_plan_num_groups.argtypes = [vp, P(i64)]
    L.fdb_plan_partial_keys.argtypes = [vp, vp, vp]
    L.fdb_plan_partial_state.argtypes = [vp, i32, vp, i64]
    L.fdb_plan_agg_type.argtypes = [vp, i32, ctypes.c_char_p]
    L.fdb_plan_state_signature.argtypes = [vp, P(ctypes.c_uint64), P(i64)]
    L.fdb_plan_state_pointers.argtypes = [vp, P(vp), P(i64), P(i64)]
    L.fdb_plan_state_read.argtypes = [vp, i32, vp, i64]
    L.fdb_plan_state_write.argtypes = [vp, i32, vp, i64]
    L.fdb_batch_import.argtypes = [vp, vp, ctypes.c_int, P(vp)]
    L.fdb_batch_num_rows.restype = i64
    L.fdb_batch_num_rows.argtypes = [vp]
    L.fdb_batch_device_bytes.restype = i64
    L.fdb_batch_device_bytes.argtypes = [vp]
    L.fdb_batch_release.argtypes = [vp]
    L.fdb_batch_release.restype = None
    L.fdb_plan_stats.argtypes = [vp, P(i64), P(ctypes.c_double), P(i64), P(i64)]
    L.fdb_regex_match.argtypes = [ctypes.c_char_p, i64, ctypes.c_char_p, i64, P(i32)]
    L.fdb_parquet_stats.argtypes = [P(i64), P(ctypes.c_double), P(ctypes.c_double), P(i64), P(i64)]
    L.fdb_jit_stats.argtypes = [P(i64), P(ctypes.c_double), P(i64)]
    L.fdb_plan_merge_ms.argtypes = [vp, P(ctypes.c_double)]
    L.fdb_plan_set_timing.argtypes = [vp, i32]
    L.fdb_plan_stream.argtypes = [vp, P(vp)]
    L.fdb_plan_set_tuning.argtypes = [vp, i32, i32]
    L.fdb_plan_set_deterministic.argtypes = [vp, i32]
    L.fdb_snappy_decode_pages.argtypes = [vp, i64, vp, i32, vp, i64, ctypes.c_int, vp, P(ctypes.c_double)]
    L.fdb_plan_state_arrays.argtypes = [vp, P(i32)]
    L.fdb_plan_state_array_op.argtypes = [vp, i32, P(i32)]
    L.fdb_plan_group_schema.argtypes = [vp, vp, vp]
    L.fdb_plan_seed_groups.argtypes = [vp, vp, vp]
    L.fdb_plan_hash_export.argtypes = [vp, vp, i32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(i32)]
    L.fdb_plan_hash_import.argtypes = [vp, vp, ctypes.c_int64]
    L.fdb_read_ceiling.argtypes = [ctypes.c_int, ctypes.c_int64, i32, ctypes.POINTER(ctypes.c_double)]
    L.fdb_plan_last_kernel.argtypes = [vp]
    L.fdb_arrow_roundtrip.argtypes = [vp, vp, vp, vp]
    L.fdb_selftest_widen.argtypes = [vp, i32, vp, i64]
    L.fdb_plan_explain.argtypes = [vp, ctypes.c_char_p, i64, P(i64)]
    L.fdb_plan_last_kernel.restype = ctypes.c_char_p
    L.fdb_comm_unique_id.argtypes = [vp]
    L.fdb_comm_init_rank.argtypes = [vp, i32, i32, ctypes.c_int, P(vp)]
    L.fdb_comm_init_all.argtypes = [P(ctypes.c_int), i32, P(vp)]
    L.fdb_comm_init_local.argtypes = [P(ctypes.c_int), i32, P(vp)]
    L.fdb_comm_rank.argtypes = [vp]
    L.fdb_comm_rank.restype = i32
    L.fdb_comm_size.argtypes = [vp]
    L.fdb_comm_size.restype = i32
    L.fdb_plan_push_many.argtypes = [vp, vp, vp, i32, P(i32)]
    L.fdb_plan_push_many.restype = i32
    L.fdb_comm_transport_ranks.argtypes = [vp]
    L.fdb_comm_transport_ranks.restype = i32
    L.fdb_comm_last_error.argtypes = [vp]
    L.fdb_comm_last_error.restype = ctypes.c_char_p
    L.fdb_comm_destroy.argtypes = [vp]
    L.fdb_comm_destroy.restype = None
    L.fdb_plan_allreduce.argtypes = [vp, vp, P(i32)]
    L.fdb_plan_exchange.argtypes = [vp, vp, P(vp)]
    L.fdb_live_allocations.argtypes = [P(i64), P(i64), P(i64)]
    L.fdb_plan_filter_batch.argtypes = [vp, vp, P(vp), P(i64)]
    L.fdb_plan_finish_batch.argtypes = [vp, P(vp), P(i64)]
    L.fdb_plan_filter_batches.argtypes = [vp, P(vp), i32, P(vp), P(i64)]
    L.fdb_plan_select_batch.argtypes = [vp, vp, vp, i64, P(i64)]
    L.fdb_batch_export.argtypes = [vp, vp, vp]
    L.fdb_batch_from_parquet.argtypes = [vp, i32, i64, ctypes.c_int, P(vp)]
    L.fdb_batches_from_parquet.argtypes = [vp, i32, ctypes.c_int, P(vp)]
    _lib = L
    return L


def jit_stats() -> dict:
    """Kernels compiled with hiprtc by this process, the wall time of those compilations, code objects loaded from the disk cache."""
    n, ms, d = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64()
    lib().fdb_jit_stats(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(d))
    return {"compiled": n.value, "compile_ms": ms.value, "disk_loads": d.value}


def regex_match(pattern, value: bytes) -> bool:
    """The library's built-in RE2-syntax engine (fdb_regex_match): unanchored match like Go's regexp.Regexp.Match. Raises FdbError
    (FDB_ERR_INVALID) for a pattern that does not compile."""
    pat = pattern.encode() if isinstance(pattern, str) else bytes(pattern)
    m = ctypes.c_int32()
    rc = lib().fdb_regex_match(pat, len(pat), value, len(value), ctypes.byref(m))
    if rc != 0:
        _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
    return bool(m.value)


def parquet_stats() -> dict:
    """fdb_batch_from_parquet accumulated over the process: calls, host-part and device-part wall ms, bytes in and out."""
    c, h, d, fb, ob = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
    lib().fdb_parquet_stats(ctypes.byref(c), ctypes.byref(h), ctypes.byref(d), ctypes.byref(fb), ctypes.byref(ob))
    return {"calls": c.value, "host_ms": h.value, "device_ms": d.value, "file_bytes": fb.value, "out_bytes": ob.value}


def read_ceiling(device: int = 0, nbytes: int = 1 << 31, reps: int = 5) -> float:
    """Best GB/s of a load-only streaming kernel over `nbytes` of HBM on this box (measurement aid)."""
    out = ctypes.c_double(0.0)
    rc = lib().fdb_read_ceiling(device, ctypes.c_int64(nbytes), reps, ctypes.byref(out))
    if rc != 0:
        raise FdbError(rc, lib().fdb_last_error().decode("utf-8", "replace"))
    return out.value


def explain(filter_expr: Optional[Expr], aggs: Sequence[AggregationFunction] = (), groups: Sequence[Column] = (), final_stage: bool = False,
            ordered: bool = False) -> str:
    """≙ Draw() of the operators this descriptor builds (`PredicateFilter (…) - HashAggregate (… by …)`), no device needed."""
    desc = to_desc(filter_expr, list(aggs), list(groups), final_stage, ordered=ordered)
    buf = ctypes.create_string_buffer(4096)
    need = ctypes.c_int64()
    rc = lib().fdb_plan_explain(ctypes.addressof(desc.desc), buf, len(buf), ctypes.byref(need))
    if rc != 0:
        _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
    return buf.value.decode()


def arrow_roundtrip(record: pa.RecordBatch) -> pa.RecordBatch:
    """Host-only self-check (fdb_arrow_roundtrip): the record through the library's Arrow import and export code, no device."""
    arr, sch = ArrowArray(), ArrowSchema()
    with ExportedBatch(record) as ex:
        rc = lib().fdb_arrow_roundtrip(ctypes.addressof(ex.array), ctypes.addressof(ex.schema), ctypes.addressof(arr), ctypes.addressof(sch))
    if rc != 0:
        _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
    return import_batch(arr, sch)


def live_allocations() -> dict:
    """Device blocks / bytes and pinned result blocks the library owns right now (0 once everything is closed and released)."""
    a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    lib().fdb_live_allocations(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return {"device_blocks": a.value, "device_bytes": b.value, "pinned_blocks": c.value}


def local_cpus(device: int = 0):
    """The CPUs next to the GPU — sysfs ``local_cpulist`` of its PCI function, i.e. the cores of the NUMA node it hangs off — or None
    when that cannot be told. Threads that push HOST records (one per chain, like the reference's goroutines) belong there: on a
    two-socket MI355X box one chain moves 1.8 G rows/s of 65 536-row records from the GPU's socket and 1.36 from the other, eight chains on
    the other socket stop at half the link (profiles/round5_push_bench_numa_sdma.txt). ``pin_thread_near(device)`` applies it to the
    calling thread."""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        with open("/sys/bus/pci/devices/%s/local_cpulist" % buf.value.decode().lower()) as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        return cpus or None
    except (OSError, ValueError, AttributeError):
        return None


def pin_thread_near(device: int = 0) -> bool:
    """Restricts the CALLING thread to ``local_cpus(device)`` (intersected with what it may run on). False: left as it was."""
    cpus = local_cpus(device)
    if not cpus:
        return False
    try:
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return False
        os.sched_setaffinity(0, allowed)
        return True
    except (OSError, AttributeError):
        return False


def device_count() -> int:
    n = ctypes.c_int(0)
    lib().fdb_device_count(ctypes.byref(n))
    return n.value


def _raise(code: int, msg: str):
    raise (UnsupportedError if code == FDB_ERR_UNSUPPORTED else FdbError)(code, msg)


class ParquetChunk(ctypes.Structure):
    """fdb_parquet_chunk: one column chunk of a row group, bytes as they sit in the file."""
    _fields_ = [("name", ctypes.c_char_p), ("physical_type", ctypes.c_int32), ("optional", ctypes.c_int32), ("utf8", ctypes.c_int32),
                ("codec", ctypes.c_int32), ("data", ctypes.c_void_p), ("n_bytes", ctypes.c_int64)]


PARQUET_INT64, PARQUET_DOUBLE, PARQUET_BYTE_ARRAY = 2, 5, 6
PARQUET_CODECS = {"UNCOMPRESSED": 0, "SNAPPY": 1, "GZIP": 2, "LZO": 3, "BROTLI": 4, "LZ4": 5, "ZSTD": 6, "LZ4_RAW": 7}  # parquet.thrift CompressionCodec


class ParquetRowGroup(ctypes.Structure):
    """fdb_parquet_row_group: the column chunks of one row group + its row count."""
    _fields_ = [("chunks", ctypes.c_void_p), ("n_chunks", ctypes.c_int32), ("n_rows", ctypes.c_int64)]


class ResidentBatch:
    """An Arrow record kept in HBM between queries (``fdb_batch``)."""

    @staticmethod
    def _parquet_chunks(chunks: Sequence[tuple]):
        arr = (ParquetChunk * len(chunks))()
        keep = []
        for i, (name, ptype, optional, utf8, data, *rest) in enumerate(chunks):
            codec = rest[0] if rest else 0
            codec = PARQUET_CODECS[codec.upper()] if isinstance(codec, str) else int(codec)
            if isinstance(data, tuple):      # (address, length): bytes that already sit somewhere stable, e.g. a pinned file buffer
                addr, size = data
            elif isinstance(data, bytes):    # no copy: the bytes object is kept alive for the call
                addr, size = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value, len(data)
            else:
                data = bytes(data)
                addr, size = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value, len(data)
            nm = name.encode()
            keep += [data, nm]
            arr[i] = ParquetChunk(nm, ptype, int(optional), 1 if utf8 else 0, codec, addr, size)  # optional: the column's max definition level (0 / 1; more = nested)
        return arr, keep

    @classmethod
    def from_parquet_many(cls, groups: Sequence[tuple], device: int = 0) -> list:
        """`groups`: (chunks, n_rows) per row group, `chunks` as for from_parquet — decoded by ONE call (fdb_batches_from_parquet):
        one copy queue for all of them, their host work side by side. Returns one ResidentBatch per row group, in order."""
        if not groups:
            return []
        rgs = (ParquetRowGroup * len(groups))()
        keep = []
        for g, (chunks, n_rows) in enumerate(groups):
            arr, k = cls._parquet_chunks(chunks)
            keep += [arr, k]
            rgs[g] = ParquetRowGroup(ctypes.addressof(arr), len(chunks), int(n_rows))
        outs = (ctypes.c_void_p * len(groups))()
        rc = lib().fdb_batches_from_parquet(rgs, len(groups), device, outs)
        if rc != 0:
            _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
        return [cls(None, device=device, _handle=outs[g]) for g in range(len(groups))]

    @classmethod
    def from_parquet(cls, chunks: Sequence[tuple], n_rows: int, device: int = 0) -> "ResidentBatch":
        """`chunks`: (name, physical type, optional, utf8, bytes-like[, codec]) per column of ONE row group — decoded on the device
        (fdb_batch_from_parquet); `codec` is a CompressionCodec number or name (default UNCOMPRESSED). The byte buffers only need
        to stay alive for the duration of the call."""
        arr = (ParquetChunk * len(chunks))()
        keep = []
        for i, (name, ptype, optional, utf8, data, *rest) in enumerate(chunks):
            codec = rest[0] if rest else 0
            codec = PARQUET_CODECS[codec.upper()] if isinstance(codec, str) else int(codec)
            if isinstance(data, tuple):      # (address, length): bytes that already sit somewhere stable, e.g. a pinned file buffer
                addr, size = data
            elif isinstance(data, bytes):    # no copy: the bytes object is kept alive for the call
                addr, size = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value, len(data)
            else:
                data = bytes(data)
                addr, size = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value, len(data)
            nm = name.encode()
            keep += [data, nm]
            arr[i] = ParquetChunk(nm, ptype, int(optional), 1 if utf8 else 0, codec, addr, size)  # optional: the column's max definition level (0 / 1; more = nested)
        out = ctypes.c_void_p()
        rc = lib().fdb_batch_from_parquet(arr, len(chunks), n_rows, device, ctypes.byref(out))
        if rc != 0:
            _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
        return cls(None, device=device, _handle=out.value)

    def __init__(self, batch: Optional[pa.RecordBatch], device: int = 0, _handle=None):
        if _handle is not None:  # a batch the library made itself (fdb_plan_filter_batch)
            self.handle, self.device = _handle, device
            return
        out = ctypes.c_void_p()
        with ExportedBatch(batch) as ex:
            rc = lib().fdb_batch_import(ctypes.addressof(ex.array), ctypes.addressof(ex.schema), device, ctypes.byref(out))
        if rc != 0:
            _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
        self.handle = out.value
        self.device = device

    def to_arrow(self) -> pa.RecordBatch:
        """The resident record copied back to the host as Arrow (fdb_batch_export)."""
        arr, sch = ArrowArray(), ArrowSchema()
        rc = lib().fdb_batch_export(self.handle, ctypes.addressof(arr), ctypes.addressof(sch))
        if rc != 0:
            _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
        return import_batch(arr, sch)

    @property
    def num_rows(self) -> int:
        return lib().fdb_batch_num_rows(self.handle)

    @property
    def device_bytes(self) -> int:
        return lib().fdb_batch_device_bytes(self.handle)

    def close(self) -> None:
        if getattr(self, "handle", None):
            lib().fdb_batch_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreparedRun:
    """The pointer tables of a run of exported host records (fdb_plan_push_many's arguments), built once."""

    def __init__(self, exported: Sequence["ExportedBatch"]):
        self.keep = list(exported)
        self.n = len(self.keep)
        self.arrs = (ctypes.c_void_p * self.n)(*[ctypes.addressof(e.array) for e in self.keep])
        self.schs = (ctypes.c_void_p * self.n)(*[ctypes.addressof(e.schema) for e in self.keep])

    def __len__(self) -> int:
        return self.n


class HashAggregatePlan:
    """One fused ``PredicateFilter → HashAggregate`` chain on one GPU."""

    def __init__(self, filter_expr: Optional[Expr], aggs: Sequence[AggregationFunction] = (),
                 groups: Sequence[Column] = (), device: int = 0, final_stage: bool = False, desc=None, regex=None, ordered: bool = False):
        """`desc`: a descriptor built once with `to_desc(filter_expr, aggs, groups, final_stage)` and shared by every chain /
        execution of the same query (≙ the logical plan being built once and `physicalplan.Build` instantiating N chains).
        `regex`: the host application's regex engine (`logicalplan.regex_matcher`), else std::regex."""
        self._desc = desc if desc is not None else to_desc(filter_expr, list(aggs), list(groups), final_stage, regex=regex, ordered=ordered)
        self.aggs = list(aggs)
        self._ctor = (filter_expr, list(aggs), list(groups), device, final_stage, self._desc)
        out = ctypes.c_void_p()
        rc = lib().fdb_plan_create(ctypes.addressof(self._desc.desc), device, ctypes.byref(out))
        if rc != 0:
            _raise(rc, lib().fdb_last_error().decode("utf-8", "replace"))
        self.handle = out.value
        self.device = device
        self._next: Optional[Callable[[pa.RecordBatch], None]] = None
        self._next_finish: Optional[Callable[[], None]] = None

    @classmethod
    def _adopt(cls, handle: int, proto: "HashAggregatePlan") -> "HashAggregatePlan":
        """Wraps a plan handle the library created itself (the shard of fdb_plan_exchange) with `proto`'s descriptor."""
        self = cls.__new__(cls)
        self._desc = proto._desc
        self.aggs = list(proto.aggs)
        self._ctor = proto._ctor
        self.handle = handle
        self.device = proto.device
        self._next = None
        self._next_finish = None
        return self

    def _check(self, rc: int) -> None:
        if rc != 0:
            _raise(rc, lib().fdb_plan_last_error(self.handle).decode("utf-8", "replace"))

    def clone_empty(self) -> "HashAggregatePlan":
        """A fresh plan with the same descriptor on the same device (no state)."""
        f, a, g, d, fs, desc = self._ctor
        return HashAggregatePlan(f, a, g, device=d, final_stage=fs, desc=desc)

    # ---- PhysicalPlan verbs --------------------------------------------------------------------------
    def Callback(self, record) -> None:
        if isinstance(record, ResidentBatch):
            self._check(lib().fdb_plan_push_batch(self.handle, record.handle))
            return
        with ExportedBatch(record) as ex:
            self._check(lib().fdb_plan_push(self.handle, ctypes.addressof(ex.array), ctypes.addressof(ex.schema)))

    def CallbackExported(self, ex: "ExportedBatch") -> None:
        """Callback for a host record whose C-data export the caller keeps (fdb_plan_push only borrows the structs, so one export
        can be pushed any number of times — measurement loops keep pyarrow's export cost out of the timed region)."""
        self._check(lib().fdb_plan_push(self.handle, ctypes.addressof(ex.array), ctypes.addressof(ex.schema)))

    def CallbackExportedMany(self, exported: Sequence["ExportedBatch"]) -> None:
        """fdb_plan_push_many: the Callbacks of a run of host records in ONE call into the library (a thread that drives a chain this
        way holds the interpreter lock once per run, not once per record)."""
        n = len(exported)
        arrs = (ctypes.c_void_p * n)(*[ctypes.addressof(e.array) for e in exported])
        schs = (ctypes.c_void_p * n)(*[ctypes.addressof(e.schema) for e in exported])
        done = ctypes.c_int32()
        self._check(lib().fdb_plan_push_many(self.handle, arrs, schs, n, ctypes.byref(done)))

    def CallbackPrepared(self, run: "PreparedRun") -> None:
        """fdb_plan_push_many over pointer tables built beforehand (PreparedRun): the call itself is ONE entry into the library and holds the
        interpreter lock for microseconds — what a goroutine of the Go shim does (its C arrays are built by that goroutine, concurrently with
        the others; here building them per call would serialise N chain threads on the interpreter lock: 0.3 ms per chain and 1 024 records)."""
        done = ctypes.c_int32()
        self._check(lib().fdb_plan_push_many(self.handle, run.arrs, run.schs, run.n, ctypes.byref(done)))

    def CallbackResident(self, records: Sequence[ResidentBatch]) -> None:
        """Callback for several HBM-resident records at once: one fused kernel launch over all of them."""
        arr = (ctypes.c_void_p * len(records))(*[r.handle for r in records])
        self._check(lib().fdb_plan_push_batches(self.handle, arr, len(records)))

    def Finish(self) -> pa.RecordBatch:
        arr, sch = ArrowArray(), ArrowSchema()
        n = ctypes.c_int64()
        self._check(lib().fdb_plan_finish(self.handle, ctypes.addressof(arr), ctypes.addressof(sch), ctypes.byref(n)))
        rec = import_batch(arr, sch)
        if self._next is not None:  # ≙ next.Callback(record) for every aggregate, then next.Finish() (aggregate.go:617-626, :540)
            if rec.num_rows:
                self._next(rec)
            while True:
                more = self.FinishNext()
                if more is None:
                    break
                self._next(more)
            if self._next_finish is not None:
                self._next_finish()
        return rec

    def FinishNext(self) -> Optional[pa.RecordBatch]:
        """The next record of a Finish that emitted several (a plain string / binary key column that would pass 2 GiB in one record starts a
        new one, aggregate.go:426-468); None when there is none left."""
        arr, sch = ArrowArray(), ArrowSchema()
        n, emitted = ctypes.c_int64(), ctypes.c_int32()
        self._check(lib().fdb_plan_finish_next(self.handle, ctypes.addressof(arr), ctypes.addressof(sch), ctypes.byref(n), ctypes.byref(emitted)))
        return import_batch(arr, sch) if emitted.value else None

    def FinishAll(self) -> List[pa.RecordBatch]:
        """≙ Finish as the next operator sees it: every record it emits, in order."""
        recs = [self.Finish()] if self._next is None else []
        if self._next is not None:
            raise FdbError(1, "FinishAll: the plan has a next operator (SetNext): Finish hands it the records")
        while True:
            more = self.FinishNext()
            if more is None:
                return recs
            recs.append(more)

    def FinishResident(self) -> "ResidentBatch":
        """≙ Finish for a device-side consumer (fdb_plan_finish_batch): the result record stays in HBM."""
        out, n = ctypes.c_void_p(), ctypes.c_int64()
        self._check(lib().fdb_plan_finish_batch(self.handle, ctypes.byref(out), ctypes.byref(n)))
        return ResidentBatch(None, device=self.device, _handle=out.value)

    def SetNext(self, callback: Callable[[pa.RecordBatch], None], finish: Optional[Callable[[], None]] = None) -> None:
        self._next, self._next_finish = callback, finish

    def Draw(self) -> str:
        return lib().fdb_plan_draw(self.handle).decode()

    def Close(self) -> None:
        if getattr(self, "handle", None):
            lib().fdb_plan_close(self.handle)
            self.handle = None

    # ---- beyond the interface ---------------------------------------------------------------------------
    def Merge(self, other: "HashAggregatePlan") -> None:
        """≙ Synchronizer + final-stage HashAggregate on one device."""
        self._check(lib().fdb_plan_merge(self.handle, other.handle))

    def Select(self, record: pa.RecordBatch):
        import numpy as np
        idx = np.zeros(max(record.num_rows, 1), dtype=np.uint32)
        n = ctypes.c_int64()
        with ExportedBatch(record) as ex:
            self._check(lib().fdb_plan_select(self.handle, ctypes.addressof(ex.array), ctypes.addressof(ex.schema),
                                              idx.ctypes.data, idx.size, ctypes.byref(n)))
        return idx[: n.value].copy()

    def Filter(self, record: pa.RecordBatch) -> Optional[pa.RecordBatch]:
        """≙ filter(): compacted record, or None when no row qualifies (filter.go:264-266)."""
        arr, sch = ArrowArray(), ArrowSchema()
        n = ctypes.c_int64()
        with ExportedBatch(record) as ex:
            self._check(lib().fdb_plan_filter(self.handle, ctypes.addressof(ex.array), ctypes.addressof(ex.schema),
                                              ctypes.addressof(arr), ctypes.addressof(sch), ctypes.byref(n)))
        if n.value == 0:
            return None
        return import_batch(arr, sch)

    def FilterResident(self, record: "ResidentBatch") -> "ResidentBatch":
        """≙ filter() on a record resident in HBM: the compacted record, resident too (zero rows when nothing qualifies)."""
        out, n = ctypes.c_void_p(), ctypes.c_int64()
        self._check(lib().fdb_plan_filter_batch(self.handle, record.handle, ctypes.byref(out), ctypes.byref(n)))
        return ResidentBatch(None, device=self.device, _handle=out.value)

    def FilterResidentMany(self, records: Sequence["ResidentBatch"]) -> List["ResidentBatch"]:
        """≙ filter() over several resident records at once (fdb_plan_filter_batches): one launch sequence for all of them."""
        n = len(records)
        arr = (ctypes.c_void_p * n)(*[r.handle for r in records])
        outs = (ctypes.c_void_p * n)()
        counts = (ctypes.c_int64 * n)()
        self._check(lib().fdb_plan_filter_batches(self.handle, arr, n, outs, counts))
        return [ResidentBatch(None, device=self.device, _handle=outs[i]) for i in range(n)]

    def SelectResident(self, record: "ResidentBatch", dev_ptr: int, capacity: int) -> int:
        """Selection vector of a resident record into a DEVICE buffer (uint32 × capacity ≥ rows); returns the number selected."""
        n = ctypes.c_int64()
        self._check(lib().fdb_plan_select_batch(self.handle, record.handle, ctypes.c_void_p(dev_ptr), capacity, ctypes.byref(n)))
        return n.value

    def num_groups(self) -> int:
        n = ctypes.c_int64()
        self._check(lib().fdb_plan_num_groups(self.handle, ctypes.byref(n)))
        return n.value

    def partial_keys(self) -> pa.RecordBatch:
        arr, sch = ArrowArray(), ArrowSchema()
        self._check(lib().fdb_plan_partial_keys(self.handle, ctypes.addressof(arr), ctypes.addressof(sch)))
        return import_batch(arr, sch)

    # ---- hash-partitioned exchange of high-cardinality partial tables (see include/frostdb_amd.h) -------------------
    def group_schema(self) -> pa.RecordBatch:
        """Zero-row record: dictionary columns carry this plan's distinct key values, plus one column per typed aggregate."""
        arr, sch = ArrowArray(), ArrowSchema()
        self._check(lib().fdb_plan_group_schema(self.handle, ctypes.addressof(arr), ctypes.addressof(sch)))
        return import_batch(arr, sch)

    def seed_groups(self, schema_record: pa.RecordBatch) -> None:
        with ExportedBatch(schema_record) as ex:
            self._check(lib().fdb_plan_seed_groups(self.handle, ctypes.addressof(ex.array), ctypes.addressof(ex.schema)))

    def hash_export(self, layout: "HashAggregatePlan", n_parts: int):
        """(device pointer, rows per partition, bytes per row): this plan's groups re-keyed for `layout`, packed by
        destination partition. The buffer belongs to this plan until its next push or Close."""
        ptr, rw = ctypes.c_void_p(), ctypes.c_int32()
        counts = (ctypes.c_int64 * n_parts)()
        self._check(lib().fdb_plan_hash_export(self.handle, layout.handle, n_parts, ctypes.byref(ptr), counts, ctypes.byref(rw)))
        return ptr.value or 0, list(counts), rw.value * 4

    def hash_import(self, dev_ptr: int, n_rows: int) -> None:
        self._check(lib().fdb_plan_hash_import(self.handle, ctypes.c_void_p(dev_ptr), n_rows))

    def state_array_ops(self) -> List[int]:
        """Merge operation of every table array (0 unused, 1 int sum, 2 float64 sum, 3 int min, 4 int max); array 0 = row counts."""
        n = ctypes.c_int32()
        self._check(lib().fdb_plan_state_arrays(self.handle, ctypes.byref(n)))
        ops = []
        for a in range(n.value):
            op = ctypes.c_int32()
            self._check(lib().fdb_plan_state_array_op(self.handle, a, ctypes.byref(op)))
            ops.append(op.value)
        return ops

    def agg_format(self, agg: int) -> str:
        c = ctypes.create_string_buffer(2)
        self._check(lib().fdb_plan_agg_type(self.handle, agg, c))
        return c.value.decode() or "l"

    def partial_state_into(self, agg: int, dst_ptr: int, capacity_bytes: int) -> None:
        """Copies aggregation `agg`'s partial column (n_groups × 8 B) to a host or device pointer."""
        self._check(lib().fdb_plan_partial_state(self.handle, agg, dst_ptr, capacity_bytes))

    def state_signature(self):
        """(layout signature, n_slots) of the plan's table — see fdb_plan_state_signature."""
        sig, n = ctypes.c_uint64(), ctypes.c_int64()
        self._check(lib().fdb_plan_state_signature(self.handle, ctypes.byref(sig), ctypes.byref(n)))
        return sig.value, n.value

    def state_pointers(self):
        """(device address of array 0, stride between arrays in elements, n_slots) — zero-copy view of the dense table."""
        base, stride, n = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        self._check(lib().fdb_plan_state_pointers(self.handle, ctypes.byref(base), ctypes.byref(stride), ctypes.byref(n)))
        return base.value or 0, stride.value, n.value

    def stream_ptr(self) -> int:
        s = ctypes.c_void_p()
        self._check(lib().fdb_plan_stream(self.handle, ctypes.byref(s)))
        return s.value or 0

    def state_read(self, array: int, dst_ptr: int, capacity_bytes: int) -> None:
        self._check(lib().fdb_plan_state_read(self.handle, array, dst_ptr, capacity_bytes))

    def state_write(self, array: int, src_ptr: int, nbytes: int) -> None:
        self._check(lib().fdb_plan_state_write(self.handle, array, src_ptr, nbytes))

    def set_timing(self, enabled: bool) -> None:
        lib().fdb_plan_set_timing(self.handle, 1 if enabled else 0)

    def set_tuning(self, rows_per_thread: int = 8, grid_blocks: int = 0) -> None:
        lib().fdb_plan_set_tuning(self.handle, rows_per_thread, grid_blocks)

    def set_deterministic(self, enabled: bool = True) -> None:
        """Reproducible float64 sums (fdb_plan_set_deterministic): the same pushes give the same bits on every run."""
        self._check(lib().fdb_plan_set_deterministic(self.handle, 1 if enabled else 0))

    def last_kernel(self) -> str:
        """Name of the scan kernel the latest push launched (``fdb_plan_kernel`` = run-time specialised)."""
        return (lib().fdb_plan_last_kernel(self.handle) or b"").decode()

    def stats(self) -> dict:
        b, ms, n, r = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
        lib().fdb_plan_stats(self.handle, ctypes.byref(b), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(r))
        mm = ctypes.c_double()
        lib().fdb_plan_merge_ms(self.handle, ctypes.byref(mm))
        return {"algorithmic_bytes": b.value, "kernel_ms": ms.value, "launches": n.value, "rows": r.value, "merge_ms": mm.value}

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass


def execute(records: Sequence, filter_expr: Optional[Expr], aggs: Sequence[AggregationFunction],
            groups: Sequence[Column], device: int = 0) -> pa.RecordBatch:
    """The engine-level shape of the path: scan `records` through one GPU chain and return the final record."""
    plan = HashAggregatePlan(filter_expr, aggs, groups, device=device)
    try:
        for r in records:
            plan.Callback(r)
        return plan.Finish()
    finally:
        plan.Close()


def snappy_decode_pages(pages: "list[bytes]", sizes: "list[int]", device: int = 0):
    """Snappy-compressed pages → their bytes, inflated on the device (fdb_snappy_decode_pages; tests and measurement).
    Returns (list of bytes — None for a page the decoder refused —, list of status codes, kernel milliseconds)."""
    import numpy as np
    n = len(pages)
    src = b"".join(pages)
    table = np.zeros((n, 3), dtype=np.uint64)  # src_off, dst_off, (src_len | dst_len << 32)
    so = do = 0
    for i, (p, z) in enumerate(zip(pages, sizes)):
        table[i] = (so, do, len(p) | (int(z) << 32))
        so += len(p); do += int(z)
    dst = np.zeros(max(do, 1), dtype=np.uint8)
    status = np.zeros(max(n, 1), dtype=np.uint32)
    ms = ctypes.c_double(0.0)
    srcb = np.frombuffer(src, dtype=np.uint8) if src else np.zeros(1, dtype=np.uint8)
    rc = lib().fdb_snappy_decode_pages(srcb.ctypes.data, len(src), table.ctypes.data, n, dst.ctypes.data, do, device, status.ctypes.data, ctypes.byref(ms))
    if rc != FDB_OK:
        _raise(rc, (lib().fdb_last_error() or b"").decode("utf-8", "replace"))
    out, at = [], 0
    for i, z in enumerate(sizes):
        out.append(bytes(dst[at:at + int(z)]) if status[i] == 0 else None)
        at += int(z)
    return out, [int(x) for x in status[:n]], ms.value
