"""Device-resident engine: torch tensors in, torch tensors out, arithmetic in HIP.

torch is used only as plumbing (device memory, the current stream); every result
comes from ``libgiql_hip.so`` through the C ABI of ``include/giql_hip.h``.

Reference interface mirrored (path:line under /root/reference/):
the per-chromosome INNER / SEMI / ANTI plans of
``src/giql/expanders/intersects_duckdb.py:1254-1330``, count_overlaps
(``:806-854``) and NEAREST k=1 (``src/giql/expanders/nearest.py:336-397``), with
each table's coordinate encoding applied as in ``src/giql/canonical.py:16-52``.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass

from . import _lib

#: (coordinate_system, interval_type) -> (start_off, end_off)
#: src/giql/canonical.py:16-52
ENCODING_OFFSETS = {
    ("0based", "half_open"): (0, 0),
    ("0based", "closed"): (0, +1),
    ("1based", "half_open"): (-1, -1),
    ("1based", "closed"): (-1, 0),
}


def _torch():
    import torch

    return torch


@dataclass
class DeviceSide:
    """One join side resident in HBM: int32 chrom ids / start / end tensors."""

    chrom: "object"
    start: "object"
    end: "object"
    start_off: int = 0
    end_off: int = 0

    def __post_init__(self) -> None:
        torch = _torch()
        for name in ("chrom", "start", "end"):
            t = getattr(self, name)
            if not isinstance(t, torch.Tensor) or t.dtype != torch.int32 or t.dim() != 1:
                raise ValueError(f"{name} must be a 1-D torch.int32 tensor")
            if not t.is_cuda:
                raise ValueError(f"{name} must live on the GPU (cuda/hip device)")
            if not t.is_contiguous():
                raise ValueError(f"{name} must be contiguous")
        if not (self.chrom.shape == self.start.shape == self.end.shape):
            raise ValueError("chrom/start/end lengths differ")

    @property
    def n(self) -> int:
        return int(self.chrom.shape[0])

    @property
    def device(self):
        return self.chrom.device

    @classmethod
    def from_numpy(cls, chrom, start, end, encoding=("0based", "half_open"), device="cuda:0"):
        import numpy as np

        torch = _torch()
        so, eo = ENCODING_OFFSETS[tuple(encoding)]

        def up(x):
            x = np.ascontiguousarray(x, dtype=np.int32)
            if x.flags.writeable:
                return torch.from_numpy(x).to(device)
            # read-only memory (cached chromosome ids, read-only Arrow buffers): torch warns about aliasing it; the
            # host tensor here is only ever READ, by the upload on the next line, and dropped
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                return torch.from_numpy(x).to(device)

        return cls(up(chrom), up(start), up(end), so, eo)

    def c_struct(self) -> _lib.CSide:
        n = self.n
        return _lib.CSide(
            self.chrom.data_ptr() if n else None,
            self.start.data_ptr() if n else None,
            self.end.data_ptr() if n else None,
            n, self.start_off, self.end_off)


class DeviceIndex:
    """A table index in HBM (``giql_hip_index``): what the reference's users get from ``CREATE INDEX ... (chrom,
    start, "end")`` (``docs/transpilation/performance.rst:111-130``).  Built by :meth:`HipEngine.index_create`,
    used by :meth:`HipEngine.inner_join_indexed`; released by :meth:`close` / garbage collection."""

    def __init__(self, engine: "HipEngine", handle, n_chrom: int):
        self.engine, self._h, self.n_chrom, self.last_pairs = engine, handle, n_chrom, 0
        n, b, g, sp = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int64(0)
        _lib.check(engine._L.giql_hip_index_info(handle, ctypes.byref(n), ctypes.byref(b), ctypes.byref(g), ctypes.byref(sp)))
        self.n, self.nbytes, self.general, self.span = int(n.value), int(b.value), bool(g.value), int(sp.value)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.engine._L.giql_hip_index_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


class HipEngine:
    """One context (device arena + bookkeeping) on one GPU.  Not thread-safe."""

    def __init__(self, device: int = 0, profiling: bool = False):
        self._L = _lib.load()
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.GiqlHipUnavailable("no HIP device visible to torch; there is no CPU fallback")
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = ctypes.c_void_p()
        _lib.check(self._L.giql_hip_create(self.device_index, ctypes.byref(h)))
        self._h = h
        self.set_profiling(profiling)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.giql_hip_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------- utilities
    def _stream(self):
        torch = _torch()
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_profiling(self, enabled) -> None:
        """``False`` / ``True`` (hipEvent pairs around every phase) / ``2`` (only around the sort passes) /
        a phase name of ``_lib.PHASES`` (only around that phase)."""
        if isinstance(enabled, str):
            enabled = 16 + _lib.PHASES.index(enabled)
        _lib.check(self._L.giql_hip_set_profiling(self._h, int(enabled)))

    def reserve(self, nbytes: int) -> None:
        _lib.check(self._L.giql_hip_reserve(self._h, int(nbytes)))

    def stats(self) -> dict:
        st = _lib.CStats()
        _lib.check(self._L.giql_hip_get_stats(self._h, ctypes.byref(st)))
        out = {k: int(getattr(st, k)) for k in
               ("n_a", "n_b", "n_out", "n_irregular_a", "n_irregular_b", "workspace_bytes", "span",
                "profiled")}
        out["join_form"] = {0: "general", 1: "uniform_b", 2: "uniform_a"}.get(int(st.reserved) & 0x0F, "?")
        out["span_hist"] = bool(int(st.reserved) & 0x10)  # fixed-length side sorted from its raw columns
        out["coarse_b"] = out["span_hist"]  # (SEMI / ANTI / COUNT: the same bit says B was sorted without its lowest digit)
        out["sort_local"] = bool(int(st.reserved) & 0x20)   # a side was sorted in three stages (bucket_sort.hip.h)
        out["sort_resorted"] = bool(int(st.reserved) & 0x40)  # the context fell back to the four-pass sort
        out["swapped"] = bool(int(st.reserved) & 0x80)  # the INNER plan ran with the sides exchanged (larger side as B)
        out["sort_tile_order"] = (int(st.reserved) >> 8) & 0x7F
        out["count_fused"] = bool(int(st.reserved) & 0x8000)  # the bucket sort answered the range bounds (no count kernel)
        out["sort_order_fallbacks"] = (int(st.reserved) >> 16) & 0x7FF
        out["bucket_bits"] = 16 - ((int(st.reserved) >> 27) & 3)  # key bits of a bucket of the last three-stage sort
        out["bucket_join"] = bool((int(st.reserved) >> 29) & 1)  # the bucket stage wrote the pairs itself (one-call form)
        out["fused_fill"] = bool((int(st.reserved) >> 30) & 1)  # the last plan launched its own fill
        out["presorted"] = bool(int(st.reserved) & 0x80000000)  # a side arrived sorted and skipped its sort
        out["total_ms"] = float(st.total_ms)
        out["phase_ms"] = {name: float(st.phase_ms[i]) for i, name in enumerate(_lib.PHASES)}
        out["phase_launches"] = {name: int(st.phase_launches[i]) for i, name in enumerate(_lib.PHASES)}
        out["phase_bytes"] = {name: int(st.phase_bytes[i]) for i, name in enumerate(_lib.PHASES)}
        return out

    def _dev_ptr(self, t, what: str, dtype=None):
        """Device pointer of a tensor argument, or None for an empty one.  Anything that does not live
        on THIS engine's device is refused here: a host pointer handed to a kernel is a GPU memory
        fault, not an exception."""
        if t is None or int(t.numel()) == 0:
            return None
        if not t.is_cuda or t.device != self.device:
            raise ValueError(f"{what} lives on {t.device}, the engine on {self.device}")
        if not t.is_contiguous():
            raise ValueError(f"{what} must be contiguous")
        if dtype is not None and t.dtype != dtype:
            raise ValueError(f"{what} must be {dtype}, got {t.dtype}")
        return t.data_ptr()

    def _check_sides(self, a: DeviceSide, b: DeviceSide) -> None:
        for s in (a, b):
            if s.n and s.device != self.device:
                raise ValueError(f"side lives on {s.device}, engine on {self.device}")

    # ------------------------------------------------------------------ INNER
    def inner_plan(self, a: DeviceSide, b: DeviceSide, n_chrom: int) -> int:
        self._check_sides(a, b)
        n = ctypes.c_int64(0)
        ca, cb = a.c_struct(), b.c_struct()
        _lib.check(self._L.giql_hip_inner_plan_dev(self._h, ca, cb, int(n_chrom), self._stream(),
                                                   ctypes.byref(n)))
        self._keepalive = (a, b)
        return int(n.value)

    def inner_fill(self, row_a, row_b) -> None:
        torch = _torch()
        cap = min(int(row_a.shape[0]), int(row_b.shape[0]))
        _lib.check(self._L.giql_hip_inner_fill_dev(
            self._h, self._dev_ptr(row_a, "row_a", torch.int32), self._dev_ptr(row_b, "row_b", torch.int32), cap,
            self._stream()))

    def inner_join_into(self, a: DeviceSide, b: DeviceSide, n_chrom: int, row_a, row_b) -> int:
        """Plan + fill into caller-owned int32 tensors in ONE C-ABI call (no stream sync between
        the two when the context's guesses hold); returns the pair count.  Raises
        :class:`GiqlHipError` with ``GIQL_ERR_CAPACITY`` when the buffers are too small (the
        plan stays valid for :meth:`inner_fill`)."""
        self._check_sides(a, b)
        cap = min(int(row_a.shape[0]), int(row_b.shape[0]))
        n = ctypes.c_int64(0)
        self.last_pairs = None
        torch = _torch()
        rc = self._L.giql_hip_inner_join_dev(
            self._h, a.c_struct(), b.c_struct(), int(n_chrom), self._dev_ptr(row_a, "row_a", torch.int32),
            self._dev_ptr(row_b, "row_b", torch.int32), cap, self._stream(), ctypes.byref(n))
        self.last_pairs = int(n.value)
        _lib.check(rc)
        return int(n.value)

    def inner_join(self, a: DeviceSide, b: DeviceSide, n_chrom: int, out=None):
        """All ``(row_a, row_b)`` with ``a INTERSECTS b``; two int32 device tensors."""
        torch = _torch()
        # A context that has joined before gets ONE call (giql_hip_inner_join_dev): buffers sized from its previous
        # result -- for large tables the pairs are then written by the sort's last stage itself, with no count /
        # scan / fill kernels and no read-back in between.  A result that does not fit comes back as
        # GIQL_ERR_CAPACITY with the exact count and a plan behind it: the ordinary fill follows.
        # The guess belongs to the INPUTS it was made on: for tables of another size it is scaled by the product of
        # the row counts (the same density assumed) and never exceeds n_a * n_b or a third of the free HBM -- a
        # guess left by a 404M-pair join must not size the buffers of a 1,000-row sub-join.
        guess = getattr(self, "_pairs_guess", 0)
        g_na, g_nb = getattr(self, "_pairs_guess_rows", (a.n, b.n))
        if guess > 0 and a.n and b.n and (g_na, g_nb) != (a.n, b.n):
            guess = int(guess * (a.n / max(g_na, 1)) * (b.n / max(g_nb, 1))) + 1
        guess = min(guess, a.n * b.n)
        row_a = row_b = None
        if out is None and guess > 0 and a.n and b.n:
            cap = min(int(guess * 1.05) + 4096, a.n * b.n)
            try:
                free = int(torch.cuda.mem_get_info(self.device)[0])
                cap = min(cap, max(free // 24, 4096))     # two int32 arrays within a third of what is free
            except Exception:  # pragma: no cover
                pass
            try:   # (a guess left over from a much larger join must not be what runs the device out of memory)
                row_a = torch.empty(cap, dtype=torch.int32, device=self.device)
                row_b = torch.empty(cap, dtype=torch.int32, device=self.device)
            except RuntimeError:
                row_a = row_b = None
                self._pairs_guess = 0
        if row_a is not None and row_b is not None:
            try:
                n = self.inner_join_into(a, b, n_chrom, row_a, row_b)
                self._pairs_guess, self._pairs_guess_rows = n, (a.n, b.n)
                return row_a[:n], row_b[:n]
            except _lib.GiqlHipError as exc:
                if exc.code == _lib.GIQL_ERR_SPAN:
                    return self._inner_by_groups(a, b, n_chrom)
                if exc.code != _lib.GIQL_ERR_CAPACITY:
                    raise
                n = int(self.last_pairs)
                del row_a, row_b
                row_a = torch.empty(n, dtype=torch.int32, device=self.device)
                row_b = torch.empty(n, dtype=torch.int32, device=self.device)
                self.inner_fill(row_a, row_b)
                self._pairs_guess, self._pairs_guess_rows = n, (a.n, b.n)
                return row_a, row_b
        try:
            n = self.inner_plan(a, b, n_chrom)
        except _lib.GiqlHipError as exc:
            if exc.code != _lib.GIQL_ERR_SPAN:
                raise
            return self._inner_by_groups(a, b, n_chrom)
        self._pairs_guess, self._pairs_guess_rows = n, (a.n, b.n)
        if out is not None and out[0].shape[0] >= n:
            row_a, row_b = out[0][:n], out[1][:n]
        else:
            row_a = torch.empty(n, dtype=torch.int32, device=self.device)
            row_b = torch.empty(n, dtype=torch.int32, device=self.device)
        self.inner_fill(row_a, row_b)
        return row_a, row_b

    # ------------------------------------------------------------- table index
    def index_create(self, side: DeviceSide, n_chrom: int) -> "DeviceIndex":
        """Build a :class:`DeviceIndex` over a table (``giql_hip_index_create_dev``): its rows keyed on the linear
        axis, grouped by 65,536-key bucket and sorted, kept in HBM (8 B per row; 12 for variable lengths).  Raises
        ``GiqlHipError`` with ``GIQL_ERR_STATE`` when the table does not take the indexed form."""
        h = ctypes.c_void_p()
        cs = side.c_struct()
        _lib.check(self._L.giql_hip_index_create_dev(self._h, ctypes.byref(cs), int(n_chrom), self._stream(),
                                                     ctypes.byref(h)))
        return DeviceIndex(self, h, int(n_chrom))

    def inner_join_indexed(self, a: DeviceSide, index: "DeviceIndex", cap: int | None = None):
        """``(row_a, row_of_indexed_table)`` int32 device tensors of ``a INTERSECTS indexed table``
        (``giql_hip_inner_join_indexed_dev``).  ``a.chrom`` speaks the indexed table's dictionary.  The buffers
        are sized from the index's previous result (``cap`` overrides); a short buffer costs one more call."""
        torch = _torch()
        if index.engine is not self:
            raise ValueError("the index was built on another engine")
        if a.n and a.device != self.device:
            raise ValueError(f"side lives on {a.device}, engine on {self.device}")
        cap = int(cap) if cap is not None else (int(index.last_pairs * 1.05) + 4096 if index.last_pairs else max(a.n, 1 << 20))
        for _attempt in range(3):
            row_a = torch.empty(cap, dtype=torch.int32, device=self.device)
            row_b = torch.empty(cap, dtype=torch.int32, device=self.device)
            try:
                n = self.inner_join_indexed_into(a, index, row_a, row_b)
            except _lib.GiqlHipError as exc:
                if exc.code != _lib.GIQL_ERR_CAPACITY:
                    raise
                del row_a, row_b
                cap = int(self.last_pairs) + 4096
                continue
            return row_a[:n], row_b[:n]
        raise _lib.GiqlHipError(_lib.GIQL_ERR_CAPACITY, "the pair count kept changing between calls")

    def inner_join_indexed_into(self, a: DeviceSide, index: "DeviceIndex", row_a, row_b) -> int:
        """The same into caller-owned int32 tensors; returns the pair count (``GIQL_ERR_CAPACITY`` with
        ``self.last_pairs`` set when they are too small)."""
        torch = _torch()
        cap = min(int(row_a.shape[0]), int(row_b.shape[0]))
        n = ctypes.c_int64(0)
        ca = a.c_struct()
        rc = self._L.giql_hip_inner_join_indexed_dev(
            self._h, index._h, ctypes.byref(ca), self._dev_ptr(row_a, "row_a", torch.int32),
            self._dev_ptr(row_b, "row_b", torch.int32), cap, self._stream(), ctypes.byref(n))
        self.last_pairs = int(n.value)
        _lib.check(rc)
        index.last_pairs = int(n.value)
        return int(n.value)

    # ------------------------------------------------ genomes longer than 2^32
    def chrom_spans(self, a: DeviceSide, b: DeviceSide, n_chrom: int):
        """Per-chromosome coordinate span (host list of ints)."""
        import numpy as np

        self._check_sides(a, b)
        spans = np.zeros(int(n_chrom), np.int64)
        _lib.check(self._L.giql_hip_chrom_spans_dev(
            self._h, a.c_struct(), b.c_struct(), int(n_chrom), spans.ctypes.data, self._stream()))
        return spans.tolist()

    def _groups(self, a: DeviceSide, b: DeviceSide, n_chrom: int):
        """Yield ``(sub_a, rows_a, sub_b, rows_b)`` per 32-bit chromosome group.

        The kernels place all chromosomes on one u32 axis; a genome whose spans sum
        past 2^32 is joined group by group (chromosomes are independent units).
        Row selection / id mapping here is data plumbing (torch indexing)."""
        torch = _torch()
        from .shard import span_groups

        groups = span_groups(self.chrom_spans(a, b, n_chrom))
        for g in groups:
            lut = torch.zeros(int(n_chrom), dtype=torch.bool, device=self.device)
            lut[torch.tensor(g, dtype=torch.long, device=self.device)] = True
            ra = torch.nonzero(lut[a.chrom.long()], as_tuple=False).flatten()
            rb = torch.nonzero(lut[b.chrom.long()], as_tuple=False).flatten()
            sa = DeviceSide(a.chrom[ra].contiguous(), a.start[ra].contiguous(), a.end[ra].contiguous(),
                            a.start_off, a.end_off)
            sb = DeviceSide(b.chrom[rb].contiguous(), b.start[rb].contiguous(), b.end[rb].contiguous(),
                            b.start_off, b.end_off)
            yield sa, ra, sb, rb

    def _inner_by_groups(self, a, b, n_chrom):
        torch = _torch()
        outs_a, outs_b = [], []
        for sa, ra, sb, rb in self._groups(a, b, n_chrom):
            n = self.inner_plan(sa, sb, n_chrom)
            la = torch.empty(n, dtype=torch.int32, device=self.device)
            lb = torch.empty(n, dtype=torch.int32, device=self.device)
            self.inner_fill(la, lb)
            outs_a.append(ra[la.long()].to(torch.int32))
            outs_b.append(rb[lb.long()].to(torch.int32))
        if not outs_a:
            z = torch.empty(0, dtype=torch.int32, device=self.device)
            return z, z.clone()
        return torch.cat(outs_a), torch.cat(outs_b)

    def _retry_by_groups(self, fn, a, b, n_chrom):
        """Run ``fn(sub_a, sub_b)`` per group after a GIQL_ERR_SPAN."""
        return [(ra, rb, fn(sa, sb)) for sa, ra, sb, rb in self._groups(a, b, n_chrom)]

    # -------------------------------------------------------------- SEMI/ANTI
    def semi_anti(self, a: DeviceSide, b: DeviceSide, n_chrom: int, anti: bool):
        """Ascending A row ids with (SEMI) / without (ANTI) an overlapping B row."""
        torch = _torch()
        self._check_sides(a, b)
        try:
            return self._semi_anti_once(a, b, n_chrom, anti)
        except _lib.GiqlHipError as exc:
            if exc.code != _lib.GIQL_ERR_SPAN:
                raise
        parts = [ra[rows.long()] for ra, _rb, rows in
                 self._retry_by_groups(lambda sa, sb: self._semi_anti_once(sa, sb, n_chrom, anti), a, b, n_chrom)]
        if not parts:
            return torch.empty(0, dtype=torch.int32, device=self.device)
        return torch.sort(torch.cat(parts)).values.to(torch.int32)

    def _semi_anti_once(self, a: DeviceSide, b: DeviceSide, n_chrom: int, anti: bool):
        torch = _torch()
        rows = torch.empty(a.n, dtype=torch.int32, device=self.device)
        n = ctypes.c_int64(0)
        _lib.check(self._L.giql_hip_semi_anti_dev(
            self._h, a.c_struct(), b.c_struct(), int(n_chrom), int(bool(anti)),
            rows.data_ptr() if a.n else None, ctypes.byref(n), self._stream()))
        return rows[: int(n.value)]

    def semi_join(self, a, b, n_chrom):
        return self.semi_anti(a, b, n_chrom, False)

    def anti_join(self, a, b, n_chrom):
        return self.semi_anti(a, b, n_chrom, True)

    # ------------------------------------------------------------------ COUNT
    def count_overlaps(self, a: DeviceSide, b: DeviceSide, n_chrom: int):
        """int64 tensor: number of overlapping B rows per A row (original order)."""
        torch = _torch()
        self._check_sides(a, b)
        try:
            return self._count_once(a, b, n_chrom)
        except _lib.GiqlHipError as exc:
            if exc.code != _lib.GIQL_ERR_SPAN:
                raise
        counts = torch.zeros(a.n, dtype=torch.int64, device=self.device)
        for ra, _rb, c in self._retry_by_groups(lambda sa, sb: self._count_once(sa, sb, n_chrom), a, b, n_chrom):
            counts[ra] = c
        return counts

    def _count_once(self, a: DeviceSide, b: DeviceSide, n_chrom: int):
        torch = _torch()
        counts = torch.zeros(a.n, dtype=torch.int64, device=self.device)
        _lib.check(self._L.giql_hip_count_dev(
            self._h, a.c_struct(), b.c_struct(), int(n_chrom),
            counts.data_ptr() if a.n else None, self._stream()))
        return counts

    # ---------------------------------------------------------------- NEAREST
    def nearest(self, a: DeviceSide, b: DeviceSide, n_chrom: int, signed: bool = False,
                max_distance=None):
        """NEAREST k=1: ``(idx_b int32, distance int64)`` per A row; idx_b=-1 = none."""
        torch = _torch()
        self._check_sides(a, b)
        try:
            return self._nearest_once(a, b, n_chrom, signed, max_distance)
        except _lib.GiqlHipError as exc:
            if exc.code != _lib.GIQL_ERR_SPAN:
                raise
        idx = torch.full((a.n,), -1, dtype=torch.int32, device=self.device)
        dist = torch.zeros(a.n, dtype=torch.int64, device=self.device)
        for ra, rb, (gi, gd) in self._retry_by_groups(
                lambda sa, sb: self._nearest_once(sa, sb, n_chrom, signed, max_distance), a, b, n_chrom):
            hit = gi >= 0
            mapped = torch.full_like(gi, -1)
            if rb.numel():
                mapped[hit] = rb[gi[hit].long()].to(torch.int32)
            idx[ra] = mapped
            dist[ra] = gd
        return idx, dist

    def _nearest_once(self, a: DeviceSide, b: DeviceSide, n_chrom: int, signed: bool = False,
                      max_distance=None):
        torch = _torch()
        # (no pre-fill: the library writes every slot -- k_nearest_unpack, or its own memsets for an empty B)
        idx = torch.empty((a.n,), dtype=torch.int32, device=self.device)
        dist = torch.empty(a.n, dtype=torch.int64, device=self.device)
        md = -1 if max_distance is None else int(max_distance)
        _lib.check(self._L.giql_hip_nearest_dev(
            self._h, a.c_struct(), b.c_struct(), int(n_chrom), int(bool(signed)), md,
            idx.data_ptr() if a.n else None, dist.data_ptr() if a.n else None, self._stream()))
        return idx, dist

    def nearest32(self, a: DeviceSide, b: DeviceSide, n_chrom: int, signed: bool = False, max_distance=None):
        """NEAREST k=1 with the 8-byte-per-row output: an ``[n_a, 2]`` int32 tensor of ``{idx_b, distance}``
        records (``giql_hip_nearest32_dev``; idx_b = -1: none).  Raises ``GiqlHipError`` (GIQL_ERR_INVALID) when a
        distance does not fit int32 -- :meth:`nearest` is the entry for such data; a genome wider than 32 bits
        (GIQL_ERR_SPAN) also belongs there."""
        torch = _torch()
        self._check_sides(a, b)
        out = torch.empty((a.n, 2), dtype=torch.int32, device=self.device)
        md = -1 if max_distance is None else int(max_distance)
        _lib.check(self._L.giql_hip_nearest32_dev(
            self._h, a.c_struct(), b.c_struct(), int(n_chrom), int(bool(signed)), md,
            out.data_ptr() if a.n else None, self._stream()))
        return out

    def nearest_k(self, a: DeviceSide, b: DeviceSide, n_chrom: int, k: int, signed: bool = False, max_distance=None):
        """NEAREST k >= 1: ``(idx_b [n_a, k] int32, distance [n_a, k] int64)`` per A row in the reference's
        order ABS(distance), start, end; unused slots idx_b = -1 (``giql_hip_nearest_k_dev``)."""
        torch = _torch()
        self._check_sides(a, b)
        k = int(k)
        # (no pre-fill: the library writes every slot, unused ones as idx -1 / distance 0)
        idx = torch.empty((a.n, k), dtype=torch.int32, device=self.device)
        dist = torch.empty((a.n, k), dtype=torch.int64, device=self.device)
        md = -1 if max_distance is None else int(max_distance)
        _lib.check(self._L.giql_hip_nearest_k_dev(
            self._h, a.c_struct(), b.c_struct(), int(n_chrom), k, int(bool(signed)), md,
            idx.data_ptr() if a.n else None, dist.data_ptr() if a.n else None, self._stream()))
        return idx, dist

    # ------------------------------------------------ GROUP BY interval + SUM
    def group_rows(self, s: DeviceSide, n_chrom: int):
        """Rows with identical (chrom, raw start, raw end) share a group: returns
        ``(group_of_row int32[n], rep_row int32[n_groups])`` -- the GROUP BY half of
        count_overlaps (``intersects_duckdb.py:806-854``)."""
        torch = _torch()
        gid = torch.empty(s.n, dtype=torch.int32, device=self.device)
        rep = torch.empty(s.n, dtype=torch.int32, device=self.device)
        g = ctypes.c_int64(0)
        _lib.check(self._L.giql_hip_group_rows_dev(
            self._h, s.c_struct(), int(n_chrom), gid.data_ptr() if s.n else None,
            rep.data_ptr() if s.n else None, ctypes.byref(g), self._stream()))
        return gid, rep[: int(g.value)]

    def segment_sum(self, values, group_of_row, n_groups: int):
        """``sums[g] = sum(values[i] for rows i of group g)`` (int64)."""
        torch = _torch()
        if values.dtype != torch.int64 or group_of_row.dtype != torch.int32:
            raise ValueError("values must be int64 and group_of_row int32")
        n = int(values.shape[0])
        sums = torch.empty(int(n_groups), dtype=torch.int64, device=self.device)
        _lib.check(self._L.giql_hip_segment_sum_dev(
            self._h, values.data_ptr() if n else None, group_of_row.data_ptr() if n else None, n,
            sums.data_ptr() if n_groups else None, int(n_groups), self._stream()))
        return sums

    # --------------------------------------------------------- CLUSTER / MERGE
    def _empty_side(self, like: DeviceSide) -> DeviceSide:
        torch = _torch()
        z = torch.empty(0, dtype=torch.int32, device=self.device)
        return DeviceSide(z, z.clone(), z.clone(), 0, 0)

    def cluster(self, s: DeviceSide, n_chrom: int, distance: int = 0, preds=None):
        """CLUSTER ids per row (int64, 1-based within each partition ``s.chrom``):
        ``src/giql/expanders/cluster.py:210-300``.  Raw coordinates (offsets must be 0).
        ``preds`` (``predicate := ... PREV(col)``, cluster.py:281-296): ``[(lhs, op, rhs)]`` as for
        :meth:`select`, operand ``("a", column)`` = the current row's value, ``("b", column)`` = its sorted
        predecessor's; columns are device tensors of ``s.n`` rows."""
        torch = _torch()
        if preds:
            return self._cluster_pred(s, n_chrom, distance, preds)
        try:
            return self._cluster_once(s, n_chrom, distance)
        except _lib.GiqlHipError as exc:
            if exc.code != _lib.GIQL_ERR_SPAN:
                raise
        ids = torch.zeros(s.n, dtype=torch.int64, device=self.device)
        for sub, rows, _sb, _rb in self._groups(s, self._empty_side(s), n_chrom):
            ids[rows] = self._cluster_once(sub, n_chrom, distance)
        return ids

    def _cluster_pred(self, s: DeviceSide, n_chrom: int, distance: int, preds):
        torch = _torch()
        for p in preds:
            for o in (p[0], p[2]):
                if o[0] in ("a", "b") and int(o[1].shape[0]) != s.n:
                    raise ValueError("a predicate column must have one value per row of the table")
        c_preds, k, _keep_alive, _nodes, n_nodes = self._c_preds(preds)
        if n_nodes:
            raise ValueError("arithmetic in a CLUSTER / MERGE predicate is not supported")
        ids = torch.empty(s.n, dtype=torch.int64, device=self.device)
        _lib.check(self._L.giql_hip_cluster_pred_dev(self._h, s.c_struct(), int(n_chrom), int(distance), c_preds, k,
                                                     ids.data_ptr() if s.n else None, self._stream()))
        return ids

    def _cluster_once(self, s: DeviceSide, n_chrom: int, distance: int):
        torch = _torch()
        ids = torch.empty(s.n, dtype=torch.int64, device=self.device)
        _lib.check(self._L.giql_hip_cluster_dev(self._h, s.c_struct(), int(n_chrom), int(distance),
                                                ids.data_ptr() if s.n else None, self._stream()))
        return ids

    def merge(self, s: DeviceSide, n_chrom: int, distance: int = 0, preds=None):
        """MERGE: ``(chrom, start, end, count)`` tensors of the merged regions ordered by
        (chrom, start) (``src/giql/expanders/merge.py:186-330``).  ``preds``: the ``predicate :=``
        argument, as for :meth:`cluster` (merge.py:201-210 hands it to the CLUSTER underneath)."""
        torch = _torch()
        if preds:
            for p in preds:
                for o in (p[0], p[2]):
                    if o[0] in ("a", "b") and int(o[1].shape[0]) != s.n:
                        raise ValueError("a predicate column must have one value per row of the table")
            return self._merge_once(s, n_chrom, distance, preds)
        try:
            return self._merge_once(s, n_chrom, distance)
        except _lib.GiqlHipError as exc:
            if exc.code != _lib.GIQL_ERR_SPAN:
                raise
        parts = [self._merge_once(sub, n_chrom, distance)
                 for sub, _rows, _sb, _rb in self._groups(s, self._empty_side(s), n_chrom)]
        if not parts:
            return self._merge_once(self._empty_side(s), n_chrom, distance)
        c, st, en, cnt = (torch.cat([p[k] for p in parts]) for k in range(4))
        order = torch.argsort(c.long() * (1 << 32) + (st.long() + (1 << 31)), stable=True)
        return c[order], st[order], en[order], cnt[order]

    def _merge_once(self, s: DeviceSide, n_chrom: int, distance: int, preds=None):
        torch = _torch()
        n = s.n
        c = torch.empty(n, dtype=torch.int32, device=self.device)
        st = torch.empty(n, dtype=torch.int32, device=self.device)
        en = torch.empty(n, dtype=torch.int32, device=self.device)
        cnt = torch.empty(n, dtype=torch.int64, device=self.device)
        m = ctypes.c_int64(0)
        outs = (c.data_ptr() if n else None, st.data_ptr() if n else None, en.data_ptr() if n else None,
                cnt.data_ptr() if n else None, n, ctypes.byref(m), self._stream())
        if preds:
            c_preds, k, _keep_alive, _nodes, n_nodes = self._c_preds(preds)
            if n_nodes:
                raise ValueError("arithmetic in a CLUSTER / MERGE predicate is not supported")
            _lib.check(self._L.giql_hip_merge_pred_dev(self._h, s.c_struct(), int(n_chrom), int(distance), c_preds, k,
                                                       *outs))
        else:
            _lib.check(self._L.giql_hip_merge_dev(self._h, s.c_struct(), int(n_chrom), int(distance), *outs))
        k = int(m.value)
        return c[:k], st[:k], en[:k], cnt[:k]

    # ------------------------------------------------------------- projection
    def take(self, cols, idx, outs=None):
        """Arrow ``take`` of fixed-width device columns by int32 row ids ``idx``
        (the reference's outer SELECT, ``intersects_duckdb.py:1402-1644``); one
        fused launch per 8 columns.  ``idx < 0`` yields zeros.  Returns new tensors,
        or fills the caller's ``outs`` (contiguous, same dtype, ``len(idx)`` rows)."""
        torch = _torch()
        cols = list(cols)
        if idx.dtype != torch.int32 or not idx.is_contiguous():
            raise ValueError("idx must be a contiguous int32 tensor")
        n = int(idx.shape[0])
        n_rows = int(cols[0].shape[0]) if cols else 0
        given = None if outs is None else list(outs)
        if given is not None and len(given) != len(cols):
            raise ValueError("outs must match cols")
        outs = []
        for k, c in enumerate(cols):
            if c.dim() != 1 or not c.is_contiguous() or int(c.shape[0]) != n_rows:
                raise ValueError("columns must be contiguous 1-D tensors of equal length")
            if c.device != idx.device:
                raise ValueError("columns and idx must live on the same device")
            if given is None:
                outs.append(torch.empty(n, dtype=c.dtype, device=c.device))
            else:
                o = given[k]
                if (o.dtype != c.dtype or o.device != c.device or o.dim() != 1 or int(o.shape[0]) != n
                        or not o.is_contiguous()):
                    raise ValueError("outs[k] must be a contiguous 1-D tensor of len(idx) rows and the column's dtype")
                outs.append(o)
        k = len(cols)
        if k == 0 or n == 0:
            return outs
        vp = ctypes.c_void_p
        c_cols = (vp * k)(*[c.data_ptr() if n_rows else None for c in cols])
        c_outs = (vp * k)(*[o.data_ptr() for o in outs])
        c_elem = (ctypes.c_int32 * k)(*[c.element_size() for c in cols])
        _lib.check(self._L.giql_hip_take_dev(self._h, c_cols, c_elem, k, n_rows, idx.data_ptr(), n,
                                             c_outs, self._stream()))
        return outs

    def take_utf8(self, offsets, data, idx):
        """``take`` of an Arrow utf8/binary column held as device tensors: int32
        ``offsets[n_rows + 1]`` and uint8 ``data``.  Returns ``(out_offsets, out_data)``."""
        torch = _torch()
        if offsets.dtype != torch.int32 or data.dtype != torch.uint8 or idx.dtype != torch.int32:
            raise ValueError("offsets/idx must be int32 and data uint8")
        n = int(idx.shape[0])
        n_rows = int(offsets.shape[0]) - 1
        if n_rows < 0:
            raise ValueError("offsets must hold n_rows + 1 entries")
        out_off = torch.empty(n + 1, dtype=torch.int32, device=idx.device)
        nbytes = ctypes.c_int64(0)
        _lib.check(self._L.giql_hip_take_utf8_plan_dev(
            self._h, offsets.data_ptr(), n_rows, idx.data_ptr() if n else None, n, out_off.data_ptr(),
            ctypes.byref(nbytes), self._stream()))
        out = torch.empty(int(nbytes.value), dtype=torch.uint8, device=idx.device)
        if nbytes.value:
            _lib.check(self._L.giql_hip_take_utf8_fill_dev(
                self._h, offsets.data_ptr(), data.data_ptr() if data.numel() else None, n_rows,
                idx.data_ptr(), n, out_off.data_ptr(), out.data_ptr(), self._stream()))
        return out_off, out

    # ----------------------------------------------------- residual predicates
    _XOPS = {"+": 16, "-": 17, "*": 18, "/": 19, "neg": 20, "abs": 21, "least": 22, "greatest": 23,
             # boolean nodes (three-valued): a whole condition as one program, GIQL_X_EQ .. GIQL_X_NOT
             "=": 24, "!=": 25, "<": 26, "<=": 27, ">": 28, ">=": 29, "isnull": 30, "notnull": 31,
             "and": 32, "or": 33, "not": 34}
    _UNARY = ("neg", "abs", "isnull", "notnull", "not")
    _NARY = ("least", "greatest", "and", "or")

    def _flatten_expr(self, tree, nodes, keep_alive) -> None:
        """Append the postfix form of ``tree`` -- ``("a" | "b", column[, valid])``, ``("lit", v)`` or
        ``(op, child, ...)`` with op one of ``+ - * / neg abs least greatest`` -- to ``nodes``."""
        kind = tree[0]
        if kind in ("a", "b", "lit"):
            o, ka = self._c_operand(tree)
            nodes.append(o)
            keep_alive.append(ka)
            return
        if kind not in self._XOPS:
            raise ValueError(f"expression operator {kind!r}")
        args = tree[1:]
        arity_ok = (len(args) == 1) if kind in self._UNARY else (len(args) >= 1 if kind in self._NARY
                                                                  else len(args) == 2)
        if not arity_ok:
            raise ValueError(f"{kind!r} with {len(args)} argument(s)")
        self._flatten_expr(args[0], nodes, keep_alive)
        op = _lib.COperand()
        op.side = self._XOPS[kind]
        if kind in self._UNARY:
            nodes.append(op)
            return
        for child in args[1:]:
            self._flatten_expr(child, nodes, keep_alive)
            nd = _lib.COperand()
            nd.side = op.side
            nodes.append(nd)

    def _c_preds(self, preds):
        """``[(lhs, op, rhs[, group])]`` -> (``giql_pred`` array, its length, the tensors to keep alive, the
        expression nodes as a ``giql_operand`` array, their number).  Predicates are AND-ed; adjacent ones
        sharing a non-zero ``group`` are OR-ed (one CNF clause).  An operand ``("expr", tree)`` is arithmetic
        over columns and literals (``giql_hip_select_expr_dev``)."""
        k = len(preds)
        if k > 16:
            raise ValueError("at most 16 predicates per call")
        c_preds = (_lib.CPred * max(k, 1))()
        keep_alive, nodes = [], []

        def operand(spec):
            if spec[0] != "expr":
                return self._c_operand(spec)
            first = len(nodes)
            self._flatten_expr(spec[1], nodes, keep_alive)
            o = _lib.COperand()
            o.side, o.lit_i, o.type = _lib.SIDE_EXPR, first, len(nodes) - first
            return o, ()

        for j, p in enumerate(preds):
            lhs, op, rhs = p[0], p[1], p[2]
            if op not in _lib.OPS:
                raise ValueError(f"operator {op!r}")
            c_preds[j].lhs, ka = operand(lhs)
            keep_alive.append(ka)
            c_preds[j].rhs, ka = operand(rhs if op not in ("isnull", "notnull", "istrue") else ("lit", 0))
            keep_alive.append(ka)
            c_preds[j].op = _lib.OPS[op]
            c_preds[j].group = int(p[3]) if len(p) > 3 else 0
        if len(nodes) > 256:
            raise ValueError("at most 256 expression nodes per call")
        c_nodes = (_lib.COperand * max(len(nodes), 1))(*nodes)
        return c_preds, k, keep_alive, c_nodes, len(nodes)

    def _c_operand(self, spec):
        """``("a" | "b", tensor[, valid_u8_tensor])`` or ``("lit", int | float)`` -> COperand."""
        torch = _torch()
        o = _lib.COperand()
        if spec[0] == "lit":
            o.side = _lib.SIDE_LIT
            v = spec[1]
            if isinstance(v, bool):
                v = int(v)
            if isinstance(v, int):
                if not -(2**63) <= v < 2**63:
                    raise ValueError("integer literal does not fit int64")
                o.lit_i, o.lit_is_float = v, 0
            elif isinstance(v, float):
                o.lit_f, o.lit_is_float = v, 1
            else:
                raise ValueError(f"unsupported literal {v!r}")
            return o, ()
        if spec[0] not in ("a", "b"):
            raise ValueError(f"operand side {spec[0]!r}")
        t = spec[1]
        types = {torch.int32: _lib.T_I32, torch.int64: _lib.T_I64, torch.float32: _lib.T_F32,
                 torch.float64: _lib.T_F64, torch.uint8: _lib.T_U8, torch.bool: _lib.T_U8}
        if t.dtype not in types or t.dim() != 1 or not t.is_contiguous():
            raise ValueError("predicate columns must be contiguous 1-D int32/int64/float32/float64/uint8 tensors")
        o.side = _lib.SIDE_A if spec[0] == "a" else _lib.SIDE_B
        o.type = types[t.dtype]
        o.data = t.data_ptr() if t.numel() else 1  # never dereferenced when empty
        valid = spec[2] if len(spec) > 2 else None
        if valid is not None:
            if valid.dtype not in (torch.uint8, torch.bool) or valid.shape != t.shape or not valid.is_contiguous():
                raise ValueError("validity must be a contiguous uint8/bool tensor of the column's length")
            o.valid = valid.data_ptr() if valid.numel() else None
        return o, (t, valid)

    def select(self, preds, idx_a=None, idx_b=None, n=None, n_rows_a=0, n_rows_b=0, want=("a", "b")):
        """Stable filter by residual predicates (the extra ON / WHERE conditions the
        reference inlines beside the INTERSECTS, ``intersects_duckdb.py:1164-1177,
        1239-1243``), given in conjunctive normal form.

        ``preds`` = ``[(lhs, op, rhs[, group])]`` with operands ``("a" | "b", column[, valid])``
        or ``("lit", value)`` and ``op`` one of ``= != <> < <= > >= isnull notnull``; the
        predicates are AND-ed, adjacent ones sharing a non-zero ``group`` are OR-ed.  Candidates are the
        pairs ``(idx_a[i], idx_b[i])``; a missing id array means "the candidate index".
        Returns the kept ``(ids_a, ids_b)`` (``None`` for a side not in ``want``)."""
        torch = _torch()
        if n is None:
            n = int((idx_a if idx_a is not None else idx_b).shape[0])
        c_preds, k, _keep_alive, c_nodes, n_nodes = self._c_preds(preds)
        for t in (idx_a, idx_b):
            if t is not None and (t.dtype != torch.int32 or not t.is_contiguous() or int(t.shape[0]) != n):
                raise ValueError("id arrays must be contiguous int32 tensors of n rows")
        out_a = torch.empty(n, dtype=torch.int32, device=self.device) if "a" in want else None
        out_b = torch.empty(n, dtype=torch.int32, device=self.device) if "b" in want else None
        kept = ctypes.c_int64(0)
        _lib.check(self._L.giql_hip_select_expr_dev(
            self._h, c_preds, k, c_nodes if n_nodes else None, n_nodes,
            idx_a.data_ptr() if idx_a is not None and n else None, int(n_rows_a),
            idx_b.data_ptr() if idx_b is not None and n else None, int(n_rows_b), int(n),
            out_a.data_ptr() if out_a is not None and n else None,
            out_b.data_ptr() if out_b is not None and n else None, ctypes.byref(kept), self._stream()))
        m = int(kept.value)
        return (out_a[:m] if out_a is not None else None, out_b[:m] if out_b is not None else None)

    def mark(self, idx, n_rows: int):
        """``flags[idx] = 1`` over ``n_rows`` zero-initialised bytes (which left rows keep a pair)."""
        torch = _torch()
        flags = torch.zeros(int(n_rows), dtype=torch.uint8, device=self.device)
        n = int(idx.shape[0])
        if n:
            _lib.check(self._L.giql_hip_mark_dev(self._h, idx.data_ptr(), n, flags.data_ptr(), int(n_rows),
                                                 self._stream()))
        return flags

    # --------------------------------------------------------------- checksum
    def pairs_checksum(self, row_a, row_b) -> int:
        torch = _torch()
        h = ctypes.c_uint64(0)
        n = min(int(row_a.shape[0]), int(row_b.shape[0]))
        _lib.check(self._L.giql_hip_pairs_checksum_dev(
            self._h, self._dev_ptr(row_a, "row_a", torch.int32), self._dev_ptr(row_b, "row_b", torch.int32), n,
            self._stream(), ctypes.byref(h)))
        return int(h.value)

    # ------------------------------------------ compact plan (multi-GPU exchange)
    def plan_export(self, q_rid, lo, cnt, s_rid, rid_add_a: int = 0, rid_add_b: int = 0):
        """Copy the last plan's compact form into caller tensors (``q_rid`` / ``lo`` / ``cnt`` of
        at least n_q int32 elements, ``s_rid`` of at least n_s), adding the two offsets to the A /
        B row ids; returns ``(query_is_a, n_q, n_s)``.  ``GIQL_ERR_STATE`` when the plan is not in
        the single-range form (exchange the pairs instead)."""
        qa, nq, ns = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
        qcap = min(int(q_rid.shape[0]), int(lo.shape[0]), int(cnt.shape[0]))
        scap = int(s_rid.shape[0])
        torch = _torch()
        _lib.check(self._L.giql_hip_inner_plan_export_dev(
            self._h, self._dev_ptr(q_rid, "q_rid", torch.int32), self._dev_ptr(lo, "lo", torch.int32),
            self._dev_ptr(cnt, "cnt", torch.int32), self._dev_ptr(s_rid, "s_rid", torch.int32), qcap, scap,
            int(rid_add_a), int(rid_add_b), ctypes.byref(qa), ctypes.byref(nq), ctypes.byref(ns), self._stream()))
        return bool(qa.value), int(nq.value), int(ns.value)

    def plan_sizes(self):
        """``(query_is_a, n_q, n_s)`` of the last plan's compact form without copying anything
        (the export call with empty buffers reports the sizes with ``GIQL_ERR_CAPACITY``)."""
        qa, nq, ns = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
        rc = self._L.giql_hip_inner_plan_export_dev(self._h, None, None, None, None, 0, 0, 0, 0, ctypes.byref(qa),
                                                    ctypes.byref(nq), ctypes.byref(ns), self._stream())
        if rc not in (_lib.GIQL_OK, _lib.GIQL_ERR_CAPACITY):
            _lib.check(rc)
        return bool(qa.value), int(nq.value), int(ns.value)

    def fill_from_plan(self, q_rid, lo, cnt, s_rid, row_q, row_s, n_pairs_expected: int = -1) -> int:
        """Expand a compact plan (this device's or another's) into ``row_q`` / ``row_s``; returns
        the pair count."""
        n = ctypes.c_int64(0)
        nq, ns = int(q_rid.shape[0]), int(s_rid.shape[0])
        cap = min(int(row_q.shape[0]), int(row_s.shape[0]))
        torch = _torch()
        i32 = torch.int32
        _lib.check(self._L.giql_hip_fill_from_plan_dev(
            self._h, self._dev_ptr(q_rid, "q_rid", i32), self._dev_ptr(lo, "lo", i32), self._dev_ptr(cnt, "cnt", i32),
            nq, self._dev_ptr(s_rid, "s_rid", i32), ns, self._dev_ptr(row_q, "row_q", i32),
            self._dev_ptr(row_s, "row_s", i32), cap, int(n_pairs_expected), self._stream(), ctypes.byref(n)))
        return int(n.value)

    # --------------------------------------------------------- host-buffer join
    def inner_join_host(self, a_cols, b_cols, n_chrom: int, offs_a=(0, 0), offs_b=(0, 0)):
        """``giql_hip_inner``: host (numpy int32) columns in, host index pairs out -- the
        PCIe-inclusive path (H2D of the six columns, join, D2H of the pairs into pinned memory).
        Returns two numpy arrays (copies; the library's buffers are released)."""
        import numpy as np

        def cside(cols, offs):
            c, s, e = (np.ascontiguousarray(x, np.int32) for x in cols)
            cs = _lib.CSide(c.ctypes.data, s.ctypes.data, e.ctypes.data, int(c.shape[0]), int(offs[0]), int(offs[1]))
            return cs, (c, s, e)

        ca, keep_a = cside(a_cols, offs_a)
        cb, keep_b = cside(b_cols, offs_b)
        n = ctypes.c_int64(0)
        pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(self._L.giql_hip_inner(self._h, ctypes.byref(ca), ctypes.byref(cb), int(n_chrom),
                                          ctypes.byref(n), ctypes.byref(pa), ctypes.byref(pb)))
        try:
            m = int(n.value)
            ra = np.ctypeslib.as_array(ctypes.cast(pa, ctypes.POINTER(ctypes.c_int32)), shape=(max(m, 1),))[:m].copy()
            rb = np.ctypeslib.as_array(ctypes.cast(pb, ctypes.POINTER(ctypes.c_int32)), shape=(max(m, 1),))[:m].copy()
        finally:
            self._L.giql_hip_free_host(pa)
            self._L.giql_hip_free_host(pb)
        del keep_a, keep_b
        return ra, rb

    def inner_join_host_timed(self, a_cols, b_cols, n_chrom: int, inspect=None):
        """Wall time (ms) and pair count of one ``giql_hip_inner`` call; the pinned results are
        released without being copied (bench.py's t_e2e).  ``inspect(row_a, row_b)``: called with numpy views
        of the library-owned arrays before they are released (parity checks)."""
        import time

        import numpy as np

        def cside(cols):
            c, s, e = (np.ascontiguousarray(x, np.int32) for x in cols)
            return _lib.CSide(c.ctypes.data, s.ctypes.data, e.ctypes.data, int(c.shape[0]), 0, 0), (c, s, e)

        ca, keep_a = cside(a_cols)
        cb, keep_b = cside(b_cols)
        n = ctypes.c_int64(0)
        pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = self._L.giql_hip_inner(self._h, ctypes.byref(ca), ctypes.byref(cb), int(n_chrom), ctypes.byref(n),
                                    ctypes.byref(pa), ctypes.byref(pb))
        ms = (time.perf_counter() - t0) * 1e3
        try:
            if rc == 0 and inspect is not None and n.value > 0:
                va = np.ctypeslib.as_array(ctypes.cast(pa, ctypes.POINTER(ctypes.c_int32)), shape=(int(n.value),))
                vb = np.ctypeslib.as_array(ctypes.cast(pb, ctypes.POINTER(ctypes.c_int32)), shape=(int(n.value),))
                inspect(va, vb)
                del va, vb
        finally:
            self._L.giql_hip_free_host(pa)
            self._L.giql_hip_free_host(pb)
        _lib.check(rc)
        del keep_a, keep_b
        return ms, int(n.value)

    # ---------------------------------------------------------------- copy probe
    def copy_probe(self, nbytes: int = 1600 << 20, reps: int = 5) -> float:
        """GB/s (read + written) of a 16-byte-per-lane copy of ``nbytes`` on this device."""
        torch = _torch()
        src = torch.empty(nbytes // 4, dtype=torch.int32, device=self.device)
        dst = torch.empty_like(src)
        src.fill_(1)
        g = ctypes.c_double(0.0)
        _lib.check(self._L.giql_hip_copy_probe_dev(self._h, src.data_ptr(), dst.data_ptr(), int(src.numel()) * 4,
                                                   int(reps), self._stream(), ctypes.byref(g)))
        return float(g.value)

    def stream_probe(self, nbytes: int = 1600 << 20, reps: int = 5, shapes=None) -> dict:
        """What this device reads / writes / copies per second (GB/s, every byte moved counted once), by access
        shape: ``{"read": best, "write": best, "copy": best, "memcpy_d2d": x, "shapes": {label: GB/s}}`` --
        16 B per lane with 1 / 4 / 8 accesses in flight, default and non-temporal cache policy, grids of
        ``n_cu x {4, 8, 16}`` blocks (``giql_hip_stream_probe_dev``).  The ceiling the kernels are held against."""
        torch = _torch()
        src = torch.empty(nbytes // 4, dtype=torch.int32, device=self.device)
        dst = torch.empty_like(src)
        src.fill_(1)
        dst.fill_(2)
        g = ctypes.c_double(0.0)

        def run(mode, in_flight, nt, per_cu):
            _lib.check(self._L.giql_hip_stream_probe_dev(
                self._h, src.data_ptr(), dst.data_ptr(), int(src.numel()) * 4, mode, in_flight, int(nt), per_cu,
                int(reps), self._stream(), ctypes.byref(g)))
            return float(g.value)

        if shapes is None:
            shapes = [(u, nt, per_cu) for u in (1, 4, 8) for nt in (0, 1) for per_cu in (4, 8, 16)]
        out = {"shapes": {}}
        for name, mode in (("read", 0), ("write", 1), ("copy", 2)):
            best = 0.0
            for u, nt, per_cu in shapes:
                v = run(mode, u, nt, per_cu)
                out["shapes"][f"{name}/x{u}/{'nt' if nt else 'dflt'}/{per_cu}perCU"] = round(v, 1)
                best = max(best, v)
            out[name] = round(best, 1)
        out["memcpy_d2d"] = round(run(3, 1, 0, 8), 1)
        return out

    @staticmethod
    def host_pool_trim(keep_bytes: int = 0) -> int:
        """Return idle page-locked output buffers of the host-buffer entry points to the OS; bytes released."""
        freed = ctypes.c_int64(0)
        _lib.check(_lib.load().giql_hip_host_pool_trim(int(keep_bytes), ctypes.byref(freed)))
        return int(freed.value)
