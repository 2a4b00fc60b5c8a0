"""frostdb_amd — MI355X-native TableScan → PredicateFilter → HashAggregate path for FrostDB.

The product is ``libfrostdb_amd.so`` (hand-written HIP kernels for gfx950 behind the C ABI of
``include/frostdb_amd.h``); this package is the thin ctypes binding plus the torch.distributed merge
of per-GPU partial tables. Nothing here imports ``oracle/`` and there is no CPU fallback.
"""
from .logicalplan import And, Col, Count, DynCol, Max, Min, Or, Sum  # noqa: F401

__all__ = ["And", "Col", "Count", "DynCol", "Max", "Min", "Or", "Sum"]
