"""Cross-GPU merge of per-GPU partial aggregate tables (≙ Synchronizer + HashAggregate(final=true),
query/physicalplan/synchronize.go:31-53, physicalplan.go:438-471) over torch.distributed.

One process per GPU. Parts / row groups are sharded across ranks with NO data-path collective: every rank runs
the fused filter+aggregate kernel over its own resident parts. The only exchange is this merge:

  1. key unification — partial tables have different key sets, so ranks first all-gather their (small) key
     columns and agree on one dense global id per distinct key tuple (rank order, first seen → deterministic);
  2. one all-reduce per aggregation on dense ``[G]`` vectors: SUM for SUM and COUNT, MIN for MIN, MAX for MAX
     (``backend="nccl"`` is RCCL over xGMI on ROCm; ``gloo`` on CPU runs the same code in the tests);
  3. rank ``dst`` materialises the final Arrow record.

Messages are G × 8 bytes per aggregation (8 KiB for the 1 024-path configs): latency-bound, microseconds.
High-cardinality tables (cfg 5: millions of groups) take `merge_plan_alltoall` instead: no rank could hold every other
rank's table, and a ring all-reduce / all-gather would be bound by ONE xGMI link. Groups are hash-partitioned by their
128-bit fingerprint; each rank ships 1/N-th of its table straight to the owner of each partition (`all_to_all_single`:
7 point-to-point xGMI links busy at once), owners merge on the device, and the result stays sharded.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import pyarrow as pa
import torch
import torch.distributed as dist

from .logicalplan import AGG_COUNT, AGG_MAX, AGG_MIN, AGG_SUM, AggregationFunction

I64_MAX = (1 << 63) - 1
I64_MIN = -(1 << 63)


def _keys_to_tuples(keys: pa.RecordBatch) -> Tuple[List[str], List[Tuple]]:
    names = list(keys.schema.names)
    cols = []
    for c in keys.columns:
        if pa.types.is_dictionary(c.type):
            c = c.dictionary_decode()
        cols.append([v.encode() if isinstance(v, str) else v for v in c.to_pylist()])
    rows = list(zip(*cols)) if cols else [()] * keys.num_rows
    return names, rows


def unify_keys(all_names: Sequence[Sequence[str]], all_rows: Sequence[Sequence[Tuple]]):
    """Global column order (first seen, rank order) and global id of every rank's local row.
    A column a rank never saw is NULL for all of its groups (aggregate.go:568-575). An integer key of 0 and a NULL key are
    ONE group, as on a single GPU and in the reference (both hash to 0, dynparquet/hashed.go:254-272); the key printed for it
    is the first one seen in rank order."""
    gnames: List[str] = []
    for names in all_names:
        for n in names:
            if n not in gnames:
                gnames.append(n)
    index: Dict[Tuple, int] = {}
    gkeys: List[Tuple] = []
    perms: List[List[int]] = []
    for names, rows in zip(all_names, all_rows):
        pos = [names.index(n) if n in names else -1 for n in gnames]
        perm = []
        for r in rows:
            k = tuple(r[p] if p >= 0 else None for p in pos)
            ident = tuple(None if (isinstance(v, int) and not isinstance(v, bool) and v == 0) else v for v in k)
            g = index.get(ident)
            if g is None:
                g = index[ident] = len(gkeys)
                gkeys.append(k)
            perm.append(g)
        perms.append(perm)
    return gnames, gkeys, perms


def _identity(func: int, dtype: torch.dtype):
    if func == AGG_MIN:
        return float("inf") if dtype.is_floating_point else I64_MAX
    if func == AGG_MAX:
        return float("-inf") if dtype.is_floating_point else I64_MIN
    return 0


def merge_partials(keys: pa.RecordBatch, partials: Sequence[torch.Tensor], aggs: Sequence[AggregationFunction],
                   key_types: Optional[Dict[str, pa.DataType]] = None, group=None, dst: int = 0) -> Optional[pa.RecordBatch]:
    """All ranks call this with their partial table: `keys` (one row per local group) and one 1-D tensor per
    aggregation (int64 or float64, on the device the process group reduces on). Returns the final record on
    rank `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    names, rows = _keys_to_tuples(keys)
    gathered: List = [None] * world
    dist.all_gather_object(gathered, (names, rows, [str(t.dtype) for t in partials]), group=group)
    gnames, gkeys, perms = unify_keys([g[0] for g in gathered], [g[1] for g in gathered])
    G = len(gkeys)
    out_cols: List[torch.Tensor] = []
    device = partials[0].device if partials else torch.device("cpu")
    perm = torch.tensor(perms[rank], dtype=torch.long, device=device)
    for j, a in enumerate(aggs):
        # a rank whose shard was empty may not know the column type yet: take it from a rank that does
        dt_names = {g[2][j] for g in gathered}
        dtype = torch.float64 if "torch.float64" in dt_names and a.func != AGG_COUNT else torch.int64
        t = torch.full((max(G, 1),), _identity(a.func, dtype), dtype=dtype, device=device)
        if perm.numel():
            t.index_copy_(0, perm, partials[j].to(dtype))
        op = dist.ReduceOp.MIN if a.func == AGG_MIN else dist.ReduceOp.MAX if a.func == AGG_MAX else dist.ReduceOp.SUM
        dist.all_reduce(t, op=op, group=group)
        out_cols.append(t[:G])
    if rank != dst:
        return None
    arrays, out_names = [], []
    for ci, n in enumerate(gnames):
        vals = [k[ci] for k in gkeys]
        ty = (key_types or {}).get(n, pa.dictionary(pa.uint32(), pa.binary()))
        vt = ty.value_type if pa.types.is_dictionary(ty) else ty
        if pa.types.is_string(vt):
            vals = [v.decode() if v is not None else None for v in vals]
        arr = pa.array(vals, type=vt)
        if pa.types.is_dictionary(ty):
            arr = arr.dictionary_encode().cast(ty)
        arrays.append(arr)
        out_names.append(n)
    for a, t in zip(aggs, out_cols):
        arrays.append(pa.array(t.cpu().numpy()))
        out_names.append(a.Name())
    return pa.RecordBatch.from_arrays(arrays, names=out_names)


class _DeviceArray:
    """Zero-copy view of plan-owned device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def layout_probe(plan, device: torch.device, group=None) -> torch.Tensor:
    """Starts the ONE tiny all-reduce that tells whether every rank's table has the same slot layout. Issue it right
    after the scan has been launched: it runs on torch's stream, next to the scan kernel, so its latency is hidden."""
    sig, n_slots = plan.state_signature()
    s = sig & ((1 << 62) - 1)  # keep it a positive int64
    probe = torch.tensor([s, -s, n_slots, -n_slots], dtype=torch.int64, device=device)
    dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=group)
    return probe


def merge_plan_aligned(plan, group=None, dst: int = 0, device: Optional[torch.device] = None, probe: Optional[torch.Tensor] = None):
    """Fast path of the cross-GPU merge. If every rank's table has the same slot layout (`layout_probe`), slot i means
    the same group everywhere and the table arrays are all-reduced IN PLACE, on the plan's own stream, through zero-copy
    tensor views of the plan's device memory: SUM for counts and sums, integer MIN/MAX for MIN/MAX (float64 MIN/MAX live
    as order-preserving int64 keys, so the integer reduction is exact). No key exchange, no copies, no host
    synchronisation besides the one Finish does anyway. Returns (True, record-or-None), or (False, None) when layouts
    differ (the caller falls back to key unification)."""
    if device is None:
        device = torch.device("cuda", plan.device)
    if probe is None:
        probe = layout_probe(plan, device, group)
    mx, nmn, ns, nns = (int(x) for x in probe.tolist())
    if mx != -nmn or ns != -nns or ns == 0:
        return False, None
    base, stride, n_slots = plan.state_pointers()
    ext = torch.cuda.ExternalStream(plan.stream_ptr(), device=device)
    ops = plan.state_array_ops()  # per PHYSICAL table array (UNIQUE owns a MIN and a MAX array, AND a MIN array)
    with torch.cuda.stream(ext):  # ordered after the scan and the fold kernel; Finish then waits for this stream
        for a, op in enumerate(ops):
            if op == 0:
                continue  # COUNT is served by the row-count array
            ptr = base + a * stride * 8
            if op == 2:
                dist.all_reduce(torch.as_tensor(_DeviceArray(ptr, n_slots, "<f8"), device=device), op=dist.ReduceOp.SUM, group=group)
            else:
                red = dist.ReduceOp.MIN if op == 3 else dist.ReduceOp.MAX if op == 4 else dist.ReduceOp.SUM
                dist.all_reduce(torch.as_tensor(_DeviceArray(ptr, n_slots, "<i8"), device=device), op=red, group=group)
    if dist.get_rank(group) != dst:
        ext.synchronize()  # the plan is about to be closed: its memory must outlive the collectives
        return True, None
    return True, plan.Finish()


def merge_plan(plan, group=None, dst: int = 0, device: Optional[torch.device] = None,
               probe: Optional[torch.Tensor] = None) -> Optional[pa.RecordBatch]:
    """Merges a HashAggregatePlan's partial table across the process group: the aligned-layout fast path when all
    ranks agree on the slot layout, otherwise key unification (device-to-device copies through the C ABI:
    fdb_plan_partial_state, then merge_partials)."""
    ok, rec = merge_plan_aligned(plan, group=group, dst=dst, device=device, probe=probe)
    if ok:
        return rec
    if any(a.func not in (AGG_SUM, AGG_MIN, AGG_MAX, AGG_COUNT) for a in plan.aggs):
        raise NotImplementedError("UNIQUE / AND aggregations merge through the table arrays only (aligned layouts or merge_plan_alltoall)")
    keys = plan.partial_keys()
    n = keys.num_rows
    if device is None:
        device = torch.device("cuda", plan.device)
    partials = []
    for j, a in enumerate(plan.aggs):
        fmt = plan.agg_format(j)
        dtype = torch.float64 if fmt == "g" else torch.int64
        t = torch.empty((n,), dtype=dtype, device=device)
        if n:
            plan.partial_state_into(j, t.data_ptr(), n * 8)
        partials.append(t)
    key_types = {f.name: f.type for f in keys.schema}
    return merge_partials(keys, partials, plan.aggs, key_types=key_types, group=group, dst=dst)


# ---- high-cardinality merge: hash-partitioned all-to-all ---------------------------------------------------------------
def _schema_to_obj(schema: pa.RecordBatch):
    """Picklable form of a plan's zero-row group schema: [(name, 'dict' | 'strs', value type, values) | (name, 'plain', type)].
    'strs' is a plain string / binary key column: fdb_plan_group_schema describes it by its value set too, with a SIGNED
    index type as the marker."""
    out = []
    for f, c in zip(schema.schema, schema.columns):
        if pa.types.is_dictionary(f.type):
            kind = "strs" if pa.types.is_signed_integer(f.type.index_type) else "dict"
            out.append((f.name, kind, str(f.type.value_type), c.dictionary.to_pylist()))
        else:
            out.append((f.name, "plain", str(f.type), None))
    return out


def unify_group_schemas(objs: Sequence[Sequence[tuple]]) -> pa.RecordBatch:
    """One global schema from every rank's: columns in first-seen order (rank order, like the Synchronizer's arrival order
    but deterministic); a dictionary column's global dictionary is the union of the ranks' values, first seen first. Key
    ids assigned from it mean the same group on every rank."""
    order: List[str] = []
    cols: Dict[str, list] = {}
    for obj in objs:
        for name, kind, ty, values in obj:
            if name not in cols:
                order.append(name)
                cols[name] = [kind, ty, [], set()]
            e = cols[name]
            if e[0] != kind or e[1] != ty:
                raise ValueError(f"group column {name!r} has different types on different ranks: {e[0]} {e[1]} vs {kind} {ty}")
            if kind in ("dict", "strs"):
                for v in values:
                    if v not in e[3]:
                        e[3].add(v)
                        e[2].append(v)
    arrays, names = [], []
    for name in order:
        kind, ty, values, _ = cols[name]
        if kind in ("dict", "strs"):
            vt = {"string": pa.string(), "utf8": pa.string(), "large_string": pa.large_string(), "large_utf8": pa.large_string(),
                  "large_binary": pa.large_binary()}.get(ty, pa.binary())
            it = pa.uint32() if kind == "dict" else pa.int32()
            arrays.append(pa.DictionaryArray.from_arrays(pa.array([], type=it), pa.array(values, type=vt)))
        else:
            arrays.append(pa.array([], type=pa.float64() if ty == "double" else pa.bool_() if ty == "bool" else pa.uint64() if ty == "uint64" else pa.int64()))
        names.append(name)
    return pa.RecordBatch.from_arrays(arrays, names=names)


# Bytes one rank sends to one peer per all_to_all call. Measured on MI355X / RCCL 2.26: a single all_to_all_single moving
# 1.4 GB silently delivered only the first ≈0.69 GB (tools/dbg history in DESIGN.md §7), so big tables go in slices.
EXCHANGE_CHUNK_BYTES = 128 << 20


def exchange_rows(send: torch.Tensor, counts: Sequence[int], words: int, group=None, chunk_bytes: int = EXCHANGE_CHUNK_BYTES):
    """The exchange step alone: `send` holds the packed rows (int64 words, `words` per row) of N partitions back to back,
    counts[p] rows for rank p. Returns (received rows grouped by source rank, rows received from each rank). Large
    partitions travel in slices of at most `chunk_bytes` per peer and call."""
    world = len(counts)
    counts = [int(c) for c in counts]
    send_counts = torch.tensor(counts, dtype=torch.int64, device=send.device)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    rc = [int(x) for x in recv_counts.tolist()]
    recv = torch.empty((sum(rc) * words,), dtype=torch.int64, device=send.device)
    send_off = [0] * world
    recv_off = [0] * world
    for p in range(1, world):
        send_off[p] = send_off[p - 1] + counts[p - 1]
        recv_off[p] = recv_off[p - 1] + rc[p - 1]
    rows_per_slice = max(1, chunk_bytes // (8 * max(words, 1)))
    # every rank must issue the same number of collectives: agree on the largest slice count
    n_slices_local = max([(c + rows_per_slice - 1) // rows_per_slice for c in counts + rc] + [0])
    t = torch.tensor([n_slices_local], dtype=torch.int64, device=send.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    n_slices = int(t.item())
    if n_slices <= 1:
        if n_slices == 1:
            dist.all_to_all_single(recv, send, output_split_sizes=[c * words for c in rc], input_split_sizes=[c * words for c in counts], group=group)
        return recv, rc
    for k in range(n_slices):
        lo = k * rows_per_slice
        in_rows = [max(0, min(rows_per_slice, c - lo)) for c in counts]
        out_rows = [max(0, min(rows_per_slice, c - lo)) for c in rc]
        inp = torch.cat([send[(send_off[p] + lo) * words:(send_off[p] + lo + in_rows[p]) * words] for p in range(world)]) if sum(in_rows) else send[:0]
        out = torch.empty((sum(out_rows) * words,), dtype=torch.int64, device=send.device)
        dist.all_to_all_single(out, inp, output_split_sizes=[r * words for r in out_rows], input_split_sizes=[r * words for r in in_rows], group=group)
        o = 0
        for p in range(world):
            n = out_rows[p] * words
            if n:
                recv[(recv_off[p] + lo) * words:(recv_off[p] + lo) * words + n] = out[o:o + n]
            o += n
    return recv, rc


def merge_plan_alltoall(plan, group=None, device: Optional[torch.device] = None, chunk_bytes: int = EXCHANGE_CHUNK_BYTES):
    """Merges high-cardinality partial tables across the process group and returns a NEW plan holding this rank's shard of
    the final groups (fingerprint % world == rank); the caller calls Finish() / Close() on it. Works for any table mode
    (a dense table is migrated to a hash table first)."""
    import os, sys, time
    prof = os.environ.get("FDB_PROFILE") is not None
    t = [time.perf_counter()]

    def mark(what):
        if prof:
            t.append(time.perf_counter())
            print(f"[fdb] alltoall {what:18s} {1e3 * (t[-1] - t[-2]):8.2f} ms", file=sys.stderr)

    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", plan.device)
    gathered: List = [None] * world
    dist.all_gather_object(gathered, _schema_to_obj(plan.group_schema()), group=group)
    mark("schema gather")
    shard = plan.clone_empty()
    try:
        shard.seed_groups(unify_group_schemas(gathered))
        mark("seed")
        ptr, counts, row_bytes = plan.hash_export(shard, world)  # synchronised: the rows are complete when this returns
        mark("export")
        words = row_bytes // 8
        n_send = sum(counts)
        send = (torch.as_tensor(_DeviceArray(ptr, n_send * words, "<i8"), device=device) if n_send
                else torch.empty((0,), dtype=torch.int64, device=device))
        recv, rc = exchange_rows(send, counts, words, group=group, chunk_bytes=chunk_bytes)
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()
        mark("exchange")
        shard.hash_import(recv.data_ptr(), sum(rc))
        mark("import")
    except Exception:
        shard.Close()
        raise
    return shard
