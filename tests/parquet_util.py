"""Test-side Parquet plumbing: pyarrow writes the files and reads their METADATA; the column chunks' bytes go to the library as
they are (what a Go host gets from parquet-go's file metadata: offsets and sizes of every column chunk)."""
import io

import pyarrow as pa
import pyarrow.parquet as pq

PHYSICAL = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6}


def write_parquet(table: pa.Table, **kw) -> bytes:
    buf = io.BytesIO()
    opts = dict(compression="NONE", use_dictionary=[n for n in table.schema.names if pa.types.is_binary(table.schema.field(n).type) or pa.types.is_string(table.schema.field(n).type) or pa.types.is_dictionary(table.schema.field(n).type)],
                write_statistics=True, data_page_size=64 * 1024, column_encoding=None)
    opts.update(kw)
    pq.write_table(table, buf, **opts)
    return buf.getvalue()


def row_group_chunks(data: bytes, rg: int):
    """[(name, physical type, optional, utf8, chunk bytes, codec name)] and the row count of row group `rg`."""
    pf = pq.ParquetFile(io.BytesIO(data))
    md = pf.metadata.row_group(rg)
    out = []
    for j in range(md.num_columns):
        col = md.column(j)
        sc = pf.schema.column(j)
        offs = [o for o in (col.dictionary_page_offset, col.data_page_offset) if o]
        start = min(offs)
        chunk = data[start:start + col.total_compressed_size]
        lt = str(sc.logical_type).lower()
        utf8 = lt.startswith("string") or (col.physical_type == "INT64" and "int" in lt and "issigned=false" in lt.replace(" ", ""))  # (INT64: the unsigned flag)
        out.append((col.path_in_schema, PHYSICAL.get(col.physical_type, -1), sc.max_definition_level, utf8, chunk, col.compression))
    return out, md.num_rows
