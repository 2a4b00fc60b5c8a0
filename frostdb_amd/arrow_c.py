"""ctypes view of the Arrow C Data Interface (include/arrow_c_data.h) + pyarrow export/import helpers.

pyarrow stands in for arrow-go's ``arrow/cdata`` package here: ``RecordBatch._export_to_c`` produces
exactly the ``struct ArrowArray`` / ``struct ArrowSchema`` pair the Go shim hands to the C ABI.
"""
from __future__ import annotations

import ctypes

import pyarrow as pa


class ArrowSchema(ctypes.Structure):
    pass


class ArrowArray(ctypes.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", ctypes.c_char_p), ("name", ctypes.c_char_p), ("metadata", ctypes.c_char_p),
    ("flags", ctypes.c_int64), ("n_children", ctypes.c_int64),
    ("children", ctypes.POINTER(ctypes.POINTER(ArrowSchema))), ("dictionary", ctypes.POINTER(ArrowSchema)),
    ("release", ctypes.c_void_p), ("private_data", ctypes.c_void_p),
]
ArrowArray._fields_ = [
    ("length", ctypes.c_int64), ("null_count", ctypes.c_int64), ("offset", ctypes.c_int64),
    ("n_buffers", ctypes.c_int64), ("n_children", ctypes.c_int64),
    ("buffers", ctypes.POINTER(ctypes.c_void_p)), ("children", ctypes.POINTER(ctypes.POINTER(ArrowArray))),
    ("dictionary", ctypes.POINTER(ArrowArray)), ("release", ctypes.c_void_p), ("private_data", ctypes.c_void_p),
]

_RELEASE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class ExportedBatch:
    """A pyarrow RecordBatch exported to C structs; releases them (if the consumer did not) on close()."""

    def __init__(self, batch: pa.RecordBatch):
        self.array = ArrowArray()
        self.schema = ArrowSchema()
        batch._export_to_c(ctypes.addressof(self.array), ctypes.addressof(self.schema))

    @property
    def array_ptr(self):
        return ctypes.byref(self.array)

    @property
    def schema_ptr(self):
        return ctypes.byref(self.schema)

    def close(self) -> None:
        for s in (self.array, self.schema):
            if s.release:
                _RELEASE_FN(s.release)(ctypes.addressof(s))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def import_batch(array: ArrowArray, schema: ArrowSchema) -> pa.RecordBatch:
    """Takes ownership of a C record (as produced by fdb_plan_finish) and returns a pyarrow RecordBatch."""
    return pa.RecordBatch._import_from_c(ctypes.addressof(array), ctypes.addressof(schema))
