"""ctypes binding of the cross-GPU merge behind the C ABI (``fdb_comm_*``, ``fdb_plan_allreduce``, ``fdb_plan_exchange``).

≙ Synchronizer + HashAggregate(final=true) (query/physicalplan/synchronize.go:31-53, physicalplan.go:438-471) when the
chains of a query run on different GPUs. Nothing here touches torch: the communicator is RCCL bound inside
``libfrostdb_amd.so`` (or the in-process peer-to-peer transport), and the only thing the host has to move itself is the
128-byte unique id of ``fdb_comm_unique_id`` — over whatever channel it already has.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import pyarrow as pa

from . import physicalplan as pp

UNIQUE_ID_BYTES = 128


def _lib():
    return pp.lib()


def unique_id() -> bytes:
    """Rank 0 of a one-process-per-GPU job calls this and ships the bytes to the other ranks."""
    buf = (ctypes.c_uint8 * UNIQUE_ID_BYTES)()
    rc = _lib().fdb_comm_unique_id(buf)
    if rc != 0:
        pp._raise(rc, _lib().fdb_last_error().decode())
    return bytes(buf)


class Comm:
    """One rank's endpoint. ``Comm(id, n_ranks, rank, device)`` joins an RCCL communicator (one process per GPU);
    ``Comm.init_all(devices)`` / ``Comm.init_local(devices)`` return every rank's endpoint of a one-process communicator."""

    def __init__(self, uid: Optional[bytes] = None, n_ranks: int = 1, rank: int = 0, device: int = 0, _handle=None):
        if _handle is not None:
            self.handle = _handle
        else:
            if uid is None or len(uid) != UNIQUE_ID_BYTES:
                raise ValueError("a 128-byte unique id is required")
            out = ctypes.c_void_p()
            buf = (ctypes.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(uid)
            rc = _lib().fdb_comm_init_rank(buf, n_ranks, rank, device, ctypes.byref(out))
            if rc != 0:
                pp._raise(rc, _lib().fdb_last_error().decode())
            self.handle = out.value

    @staticmethod
    def _many(fn, devices: Sequence[int]) -> List["Comm"]:
        n = len(devices)
        devs = (ctypes.c_int * n)(*devices)
        outs = (ctypes.c_void_p * n)()
        rc = fn(devs, n, outs)
        if rc != 0:
            pp._raise(rc, _lib().fdb_last_error().decode())
        return [Comm(_handle=outs[i]) for i in range(n)]

    @staticmethod
    def init_all(devices: Sequence[int]) -> List["Comm"]:
        """ncclCommInitAll: one process drives len(devices) GPUs (the reference's N chains in one process)."""
        return Comm._many(_lib().fdb_comm_init_all, devices)

    @staticmethod
    def init_local(devices: Sequence[int]) -> List["Comm"]:
        """In-process peer-to-peer transport; ranks may share a device. Each endpoint is driven by its own thread."""
        return Comm._many(_lib().fdb_comm_init_local, devices)

    @property
    def rank(self) -> int:
        return _lib().fdb_comm_rank(self.handle)

    @property
    def size(self) -> int:
        return _lib().fdb_comm_size(self.handle)

    @property
    def transport_ranks(self) -> int:
        """What the transport itself says the communicator's size is (RCCL: ncclCommCount); -1 if it cannot say."""
        return _lib().fdb_comm_transport_ranks(self.handle)

    # ---- plan-level merges (collective: every rank calls) --------------------------------------------------------------
    def allreduce(self, plan: "pp.HashAggregatePlan") -> bool:
        """In-place all-reduce of the plan's dense table when every rank has the same slot layout. False: nothing changed."""
        aligned = ctypes.c_int32()
        plan._check(_lib().fdb_plan_allreduce(plan.handle, self.handle, ctypes.byref(aligned)))
        return bool(aligned.value)

    def merge_alltoall(self, plan: "pp.HashAggregatePlan") -> "pp.HashAggregatePlan":
        """Schema agreement + hash-partitioned exchange; returns this rank's SHARD of the merged groups as a new plan
        (Finish() / Close() it)."""
        out = ctypes.c_void_p()
        plan._check(_lib().fdb_plan_exchange(plan.handle, self.handle, ctypes.byref(out)))
        return pp.HashAggregatePlan._adopt(out.value, plan)

    def merge(self, plan: "pp.HashAggregatePlan", dst: int = 0) -> Optional[pa.RecordBatch]:
        """The final record of a low-cardinality query on rank `dst` (None elsewhere): the aligned in-place all-reduce when
        layouts agree, otherwise the exchange — whose result is sharded, so then EVERY rank returns its shard's record
        (the reference's OutputPlan callback accepts several records)."""
        if self.allreduce(plan):
            return plan.Finish() if self.rank == dst else None
        shard = self.merge_alltoall(plan)
        try:
            return shard.Finish()
        finally:
            shard.Close()

    def close(self) -> None:
        if getattr(self, "handle", None):
            _lib().fdb_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
