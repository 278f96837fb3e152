"""BED / delimited-text ingestion into Arrow tables for ``execute()``.

The reference leaves ingestion to the engine (``read_csv`` in DuckDB, ``scan_csv`` in
Polars, ``oxbow`` for NGS formats: ``docs/transpilation/performance.rst:19-109``); the
hip path executes on Arrow tables, so this is the matching convenience: a BED file (or
any tab-separated interval file) becomes a ``pyarrow.Table`` with the column names and
types the default :class:`giql_amd.table.Table` expects -- ``chrom`` string, ``start`` /
``end`` int32 -- read in streaming record batches by ``pyarrow.csv``.
"""

from __future__ import annotations

BED_COLUMNS = ["chrom", "start", "end", "name", "score", "strand", "thick_start", "thick_end",
               "item_rgb", "block_count", "block_sizes", "block_starts"]


def read_bed(path, columns=None, *, block_size: int = 64 << 20):
    """Read a BED3..BED12 file (no header; ``#`` / ``track`` / ``browser`` lines skipped).

    ``columns`` overrides the column names (default: the first N standard BED names,
    N = the file's field count).  ``start`` / ``end`` (and the other integer BED fields)
    are read as int32, ``score`` as float64 when it does not parse as an integer.
    """
    import pyarrow as pa
    import pyarrow.csv as pacsv

    with open(path, "rb") as f:
        n_fields = 0
        skip = 0
        for line in f:
            text = line.decode("utf-8", "replace").rstrip("\r\n")
            if not text or text.startswith(("#", "track", "browser")):
                skip += 1
                continue
            n_fields = len(text.split("\t"))
            break
    if n_fields == 0:
        names = list(columns) if columns else BED_COLUMNS[:3]
        types = {"start": pa.int32(), "end": pa.int32()}
        return pa.table({n: pa.array([], types.get(n, pa.string())) for n in names})
    names = list(columns) if columns else BED_COLUMNS[:n_fields]
    if len(names) != n_fields:
        raise ValueError(f"{path}: {n_fields} fields per line but {len(names)} column names")
    int_cols = {"start", "end", "thick_start", "thick_end", "block_count"}
    col_types = {n: pa.int32() for n in names if n in int_cols}
    col_types.update({n: pa.string() for n in names if n in ("chrom", "name", "strand", "item_rgb",
                                                               "block_sizes", "block_starts")})
    reader = pacsv.open_csv(
        path,
        read_options=pacsv.ReadOptions(column_names=names, skip_rows=skip, block_size=block_size),
        parse_options=pacsv.ParseOptions(delimiter="\t", quote_char=False),
        convert_options=pacsv.ConvertOptions(column_types=col_types, strings_can_be_null=False))
    batches = [b for b in reader]
    if not batches:
        return pa.table({n: pa.array([], col_types.get(n, pa.string())) for n in names})
    return pa.Table.from_batches(batches)
