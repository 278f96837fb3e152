"""oracle/pyoracle.py -- Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; nothing under ``giql_amd/`` does.

Two independent restatements of the reference semantics live here:

* ``np_*``  -- pure numpy / Python loops, literal predicate, for small cases
  (follows ``tests/test_duckdb_iejoin.py:34-81`` and
  ``src/giql/expanders/_distance.py:67-87`` of the reference);
* ``c_*``   -- ctypes calls into ``oracle/libgiql_oracle.so`` (``giql_oracle.c``),
  brute force and per-chromosome sweep, used for larger cases and as the timed
  CPU baseline.

Citations are ``path:line`` under ``/root/reference/``.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgiql_oracle.so")

#: (coordinate_system, interval_type) -> (start_off, end_off); restates
#: src/giql/canonical.py:16-52.
ENCODING_OFFSETS = {
    ("0based", "half_open"): (0, 0),
    ("0based", "closed"): (0, +1),
    ("1based", "half_open"): (-1, -1),
    ("1based", "closed"): (-1, 0),
}


@dataclass
class Side:
    """One join side: int32 chrom ids / start / end plus canonical offsets."""

    chrom: np.ndarray
    start: np.ndarray
    end: np.ndarray
    start_off: int = 0
    end_off: int = 0

    def __post_init__(self) -> None:
        self.chrom = np.ascontiguousarray(self.chrom, dtype=np.int32)
        self.start = np.ascontiguousarray(self.start, dtype=np.int32)
        self.end = np.ascontiguousarray(self.end, dtype=np.int32)
        assert self.chrom.shape == self.start.shape == self.end.shape

    @property
    def n(self) -> int:
        return int(self.chrom.shape[0])

    @property
    def cs(self) -> np.ndarray:
        return self.start.astype(np.int64) + self.start_off

    @property
    def ce(self) -> np.ndarray:
        return self.end.astype(np.int64) + self.end_off


def make_side(rows, encoding=("0based", "half_open"), chrom_ids=None) -> Side:
    """Build a Side from ``[(chrom, start, end), ...]`` with str or int chroms."""
    so, eo = ENCODING_OFFSETS[tuple(encoding)]
    if chrom_ids is None:
        chrom_ids = {}
    ch = []
    for r in rows:
        c = r[0]
        if not isinstance(c, (int, np.integer)):
            c = chrom_ids.setdefault(c, len(chrom_ids))
        ch.append(int(c))
    st = [int(r[1]) for r in rows]
    en = [int(r[2]) for r in rows]
    return Side(np.array(ch, np.int32), np.array(st, np.int32), np.array(en, np.int32), so, eo)


# ----------------------------------------------------------------- numpy / python
def np_inner(a: Side, b: Side) -> np.ndarray:
    """All (row_a, row_b) with the literal predicate; sorted; shape (P, 2)."""
    out = []
    bcs, bce, bch = b.cs, b.ce, b.chrom
    acs, ace = a.cs, a.ce
    for i in range(a.n):
        m = (bch == a.chrom[i]) & (acs[i] < bce) & (ace[i] > bcs)
        js = np.nonzero(m)[0]
        if js.size:
            out.append(np.stack([np.full(js.size, i, np.int64), js.astype(np.int64)], 1))
    if not out:
        return np.zeros((0, 2), np.int64)
    return np.concatenate(out, 0)


def np_count(a: Side, b: Side) -> np.ndarray:
    p = np_inner(a, b)
    return np.bincount(p[:, 0], minlength=a.n).astype(np.int64)


def np_semi_anti(a: Side, b: Side, anti: bool) -> np.ndarray:
    c = np_count(a, b)
    keep = (c == 0) if anti else (c > 0)
    return np.nonzero(keep)[0].astype(np.int64)


def py_distance(as_, ae, bs, be, signed=False) -> int:
    """The unstranded distance CASE, src/giql/expanders/_distance.py:67-87."""
    if as_ < be and ae > bs:
        return 0
    if ae <= bs:
        return bs - ae + 1
    d = as_ - be + 1
    return -d if signed else d


def py_nearest_k1(a: Side, b: Side, signed=False, max_distance=None):
    """Per A row ``(idx_b, distance)``; idx_b == -1 when no candidate."""
    idx = np.full(a.n, -1, np.int64)
    dist = np.zeros(a.n, np.int64)
    acs, ace, bcs, bce = a.cs, a.ce, b.cs, b.ce
    for i in range(a.n):
        best = None
        for j in range(b.n):
            if b.chrom[j] != a.chrom[i]:
                continue
            d = py_distance(int(acs[i]), int(ace[i]), int(bcs[j]), int(bce[j]), signed)
            if max_distance is not None and abs(d) > max_distance:
                continue
            key = (abs(d), int(bcs[j]), int(bce[j]), j)
            if best is None or key < best[0]:
                best = (key, d)
        if best is not None:
            idx[i] = best[0][3]
            dist[i] = best[1]
    return idx, dist


# ------------------------------------------------------------------------ ctypes
class _CSide(ctypes.Structure):
    _fields_ = [
        ("chrom", ctypes.c_void_p),
        ("start", ctypes.c_void_p),
        ("end", ctypes.c_void_p),
        ("n", ctypes.c_int64),
        ("start_off", ctypes.c_int32),
        ("end_off", ctypes.c_int32),
    ]


_lib = None


def build_lib(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build_lib()
        L = ctypes.CDLL(_LIB_PATH)
        P = ctypes.POINTER
        L.ora_inner_brute.argtypes = [P(_CSide), P(_CSide), P(ctypes.c_int64),
                                      P(ctypes.c_void_p), P(ctypes.c_void_p)]
        L.ora_inner_sweep.argtypes = [P(_CSide), P(_CSide), ctypes.c_int, P(ctypes.c_int64),
                                      P(ctypes.c_void_p), P(ctypes.c_void_p)]
        L.ora_count_brute.argtypes = [P(_CSide), P(_CSide), ctypes.c_void_p]
        L.ora_count_sweep.argtypes = [P(_CSide), P(_CSide), ctypes.c_int, ctypes.c_void_p]
        L.ora_semi_anti.argtypes = [P(_CSide), P(_CSide), ctypes.c_int, ctypes.c_int,
                                    P(ctypes.c_int64), P(ctypes.c_void_p)]
        L.ora_nearest_k1_brute.argtypes = [P(_CSide), P(_CSide), ctypes.c_int, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_void_p]
        L.ora_nearest_k_brute.argtypes = [P(_CSide), P(_CSide), ctypes.c_int32, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ora_nearest_k1_sweep.argtypes = [P(_CSide), P(_CSide), ctypes.c_int, ctypes.c_int64,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ora_cluster.argtypes = [P(_CSide), ctypes.c_int64, ctypes.c_void_p]
        L.ora_merge.argtypes = [P(_CSide), ctypes.c_int64, P(ctypes.c_int64), P(ctypes.c_void_p),
                                P(ctypes.c_void_p), P(ctypes.c_void_p), P(ctypes.c_void_p)]
        L.ora_pairs_checksum.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.ora_pairs_checksum.restype = ctypes.c_uint64
        L.ora_free.argtypes = [ctypes.c_void_p]
        L.ora_free.restype = None
        L.ora_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _cside(s: Side) -> _CSide:
    return _CSide(s.chrom.ctypes.data, s.start.ctypes.data, s.end.ctypes.data,
                  s.n, s.start_off, s.end_off)


def _take(ptr: ctypes.c_void_p, n: int) -> np.ndarray:
    if n == 0:
        out = np.zeros(0, np.int32)
    else:
        out = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int32)),
                                    shape=(n,)).copy()
    lib().ora_free(ptr)
    return out


def max_threads() -> int:
    return int(lib().ora_max_threads())


def c_inner(a: Side, b: Side, method: str = "sweep", threads: int | None = None):
    """Returns (row_a, row_b) int32 arrays (unsorted)."""
    L = lib()
    n = ctypes.c_int64(0)
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    ca, cb = _cside(a), _cside(b)
    if method == "brute":
        rc = L.ora_inner_brute(ca, cb, n, pa, pb)
    else:
        rc = L.ora_inner_sweep(ca, cb, threads or max_threads(), n, pa, pb)
    if rc:
        raise RuntimeError(f"oracle inner failed rc={rc}")
    return _take(pa, n.value), _take(pb, n.value)


def c_count(a: Side, b: Side, method: str = "sweep", threads: int | None = None) -> np.ndarray:
    L = lib()
    out = np.zeros(a.n, np.int64)
    ca, cb = _cside(a), _cside(b)
    if method == "brute":
        rc = L.ora_count_brute(ca, cb, out.ctypes.data)
    else:
        rc = L.ora_count_sweep(ca, cb, threads or max_threads(), out.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle count failed rc={rc}")
    return out


def c_semi_anti(a: Side, b: Side, anti: bool, threads: int | None = None) -> np.ndarray:
    L = lib()
    n = ctypes.c_int64(0)
    p = ctypes.c_void_p()
    rc = L.ora_semi_anti(_cside(a), _cside(b), int(bool(anti)), threads or max_threads(), n, p)
    if rc:
        raise RuntimeError(f"oracle semi/anti failed rc={rc}")
    return _take(p, n.value)


def c_nearest_k1(a: Side, b: Side, signed=False, max_distance=None, method="sweep",
                 threads: int | None = None):
    L = lib()
    idx = np.full(a.n, -1, np.int32)
    dist = np.zeros(a.n, np.int64)
    md = -1 if max_distance is None else int(max_distance)
    ca, cb = _cside(a), _cside(b)
    if method == "brute":
        rc = L.ora_nearest_k1_brute(ca, cb, int(bool(signed)), md, idx.ctypes.data, dist.ctypes.data)
    else:
        rc = L.ora_nearest_k1_sweep(ca, cb, int(bool(signed)), md, threads or max_threads(),
                                    idx.ctypes.data, dist.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle nearest failed rc={rc}")
    return idx, dist


def py_nearest_k(a: Side, b: Side, k: int, signed=False, max_distance=None):
    """NEAREST k >= 1 in pure Python: per A row the list of (idx_b, distance) under
    ``ORDER BY ABS(distance), start, end LIMIT k`` (nearest.py:336-397)."""
    out = []
    acs, ace, bcs, bce = a.cs, a.ce, b.cs, b.ce
    for i in range(a.n):
        cand = []
        for j in range(b.n):
            if b.chrom[j] != a.chrom[i]:
                continue
            d = py_distance(int(acs[i]), int(ace[i]), int(bcs[j]), int(bce[j]), signed)
            if max_distance is not None and abs(d) > max_distance:
                continue
            cand.append((abs(d), int(bcs[j]), int(bce[j]), j, d))
        cand.sort()
        out.append([(j, d) for _, _, _, j, d in cand[:k]])
    return out


def c_nearest_k(a: Side, b: Side, k: int, signed=False, max_distance=None, threads: int | None = None):
    """C brute force of NEAREST k: ``(idx_b [n_a, k] int32, distance [n_a, k] int64)``, -1 / 0 = no row."""
    L = lib()
    idx = np.full((a.n, k), -1, np.int32)
    dist = np.zeros((a.n, k), np.int64)
    md = -1 if max_distance is None else int(max_distance)
    rc = L.ora_nearest_k_brute(_cside(a), _cside(b), int(k), int(bool(signed)), md, threads or max_threads(),
                               idx.ctypes.data, dist.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle nearest k failed rc={rc}")
    return idx, dist


def py_cluster(s: Side, distance: int = 0) -> np.ndarray:
    """CLUSTER ids per row: the window SQL of src/giql/expanders/cluster.py:210-300 in
    plain Python (raw coordinates; stable order by (chrom, start); peers share the SUM)."""
    order = sorted(range(s.n), key=lambda i: (int(s.chrom[i]), int(s.start[i]), i))
    ids = np.zeros(s.n, np.int64)
    flags = {}
    run_max = None
    prev_c = None
    for i in order:
        c = int(s.chrom[i])
        if c != prev_c:
            flags[i] = 1
            run_max = int(s.end[i])
        else:
            flags[i] = 0 if run_max + distance >= int(s.start[i]) else 1
            run_max = max(run_max, int(s.end[i]))
        prev_c = c
    for i in order:  # SUM(...) OVER (PARTITION BY chrom ORDER BY start): RANGE frame, peers included
        ids[i] = sum(flags[j] for j in order
                     if int(s.chrom[j]) == int(s.chrom[i]) and int(s.start[j]) <= int(s.start[i]))
    return ids


def py_merge(s: Side, distance: int = 0):
    """MERGE rows ``(chrom, start, end, count)`` ordered by (chrom, start)
    (src/giql/expanders/merge.py:186-330 over py_cluster's ids)."""
    ids = py_cluster(s, distance)
    groups = {}
    for i in range(s.n):
        k = (int(s.chrom[i]), int(ids[i]))
        g = groups.setdefault(k, [int(s.start[i]), int(s.end[i]), 0])
        g[0] = min(g[0], int(s.start[i]))
        g[1] = max(g[1], int(s.end[i]))
        g[2] += 1
    return sorted((c, g[0], g[1], g[2]) for (c, _), g in groups.items())


def c_cluster(s: Side, distance: int = 0) -> np.ndarray:
    ids = np.zeros(s.n, np.int64)
    rc = lib().ora_cluster(ctypes.byref(_cside(s)), int(distance), ids.ctypes.data)
    if rc != 0:
        raise MemoryError("ora_cluster")
    return ids


def py_cluster_predicate(part, start, end, distance, holds) -> np.ndarray:
    """CLUSTER with ``predicate := ...`` restated (src/giql/expanders/cluster.py:210-300, 281-296): per partition
    (``part``: any hashable per row), rows in start order (equal starts: input order -- upstream leaves that to the
    engine), a row opens a new cluster unless the running MAX(end) of the partition's preceding rows + distance
    reaches its start AND ``holds(row, predecessor_row)`` is true; ids count from 1 inside a partition.  ``holds``
    gets row indices and must return False where the SQL predicate is NULL.  Pure Python: small cases only."""
    n = len(start)
    ids = np.zeros(n, np.int64)
    groups: dict = {}
    for i in range(n):
        groups.setdefault(part[i], []).append(i)
    for rows in groups.values():
        rows.sort(key=lambda i: (start[i], i))
        cid, run_max, prev = 0, None, None
        for i in rows:
            keep = prev is not None and run_max + max(int(distance), 0) >= start[i] and holds(i, prev)
            if not keep:
                cid += 1
            ids[i] = cid
            run_max = end[i] if run_max is None else max(run_max, end[i])
            prev = i
    return ids


def c_merge(s: Side, distance: int = 0):
    n = ctypes.c_int64(0)
    pc, ps, pe, pn = (ctypes.c_void_p() for _ in range(4))
    rc = lib().ora_merge(ctypes.byref(_cside(s)), int(distance), ctypes.byref(n), ctypes.byref(pc),
                         ctypes.byref(ps), ctypes.byref(pe), ctypes.byref(pn))
    if rc != 0:
        raise MemoryError("ora_merge")
    m = int(n.value)
    cnt = (np.ctypeslib.as_array(ctypes.cast(pn, ctypes.POINTER(ctypes.c_int64)), shape=(m,)).copy()
           if m else np.zeros(0, np.int64))
    lib().ora_free(pn)
    return _take(pc, m), _take(ps, m), _take(pe, m), cnt


def c_pairs_checksum(row_a: np.ndarray, row_b: np.ndarray) -> int:
    ra = np.ascontiguousarray(row_a, np.int32)
    rb = np.ascontiguousarray(row_b, np.int32)
    return int(lib().ora_pairs_checksum(ra.ctypes.data, rb.ctypes.data, ra.shape[0]))


def sort_pairs(row_a, row_b) -> np.ndarray:
    """Canonical sorted (P, 2) int64 form of a pair multiset."""
    p = np.stack([np.asarray(row_a, np.int64), np.asarray(row_b, np.int64)], 1)
    if p.shape[0] == 0:
        return p
    order = np.lexsort((p[:, 1], p[:, 0]))
    return p[order]
