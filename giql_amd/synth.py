"""Deterministic synthetic interval tables (SURVEY.md §8(d) / BASELINE.md §3).

24 chromosomes with hg38 lengths, rows per chromosome ~ multinomial(N, G_c/sum G),
length drawn first, then a uniform start in [0, G_c - len), end = start + len,
int32, rows shuffled (unsorted input), no nulls.  ``peaks`` have len ~ U[200,2000),
``reads`` have len = 150.

Generation is per chromosome with its own PCG64 stream so a rank that holds only
some chromosomes (multi-GPU sharding) produces exactly the rows the single-GPU
run has for them.
"""

from __future__ import annotations

import numpy as np

HG38_LENGTHS = np.array([
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
    145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
    101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
    156040895, 57227415], dtype=np.int64)
HG38_NAMES = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]


def rows_per_chrom(n: int, seed: int, lengths=HG38_LENGTHS) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.multinomial(n, lengths / lengths.sum()).astype(np.int64)


def _draw_lengths(rng, n, kind):
    if kind == "peaks":
        return rng.integers(200, 2000, size=n, dtype=np.int64)
    if kind == "reads":
        return np.full(n, 150, dtype=np.int64)
    raise ValueError(f"unknown kind {kind!r}")


def make_table(n: int, seed: int, kind: str = "peaks", lengths=HG38_LENGTHS, chroms=None,
               shuffle: bool = True):
    """Return ``(chrom, start, end)`` int32 arrays.

    ``chroms``: optional iterable of chromosome ids to generate (a shard); ids in
    the output are still the global ids.  Row order: per-chromosome blocks, then a
    seeded shuffle of the generated rows when ``shuffle``.
    """
    lengths = np.asarray(lengths, dtype=np.int64)
    counts = rows_per_chrom(n, seed, lengths)
    sel = range(len(lengths)) if chroms is None else sorted(int(c) for c in chroms)
    cs, ss, es = [], [], []
    for c in sel:
        m = int(counts[c])
        if m == 0:
            continue
        rng = np.random.Generator(np.random.PCG64([seed, c]))
        ln = _draw_lengths(rng, m, kind)
        st = rng.integers(0, lengths[c] - ln, dtype=np.int64)
        cs.append(np.full(m, c, dtype=np.int32))
        ss.append(st.astype(np.int32))
        es.append((st + ln).astype(np.int32))
    if not cs:
        z = np.zeros(0, np.int32)
        return z, z.copy(), z.copy()
    chrom = np.concatenate(cs)
    start = np.concatenate(ss)
    end = np.concatenate(es)
    if shuffle:
        key = (seed * 1_000_003 + (0 if chroms is None else 1 + sum((i + 1) * c for i, c in enumerate(sel)))) % (2**63)
        perm = np.random.Generator(np.random.PCG64(key)).permutation(chrom.shape[0])
        chrom, start, end = chrom[perm], start[perm], end[perm]
    return chrom, start, end


def make_single_chrom(n: int, seed: int, kind: str = "peaks", genome_len: int = 248956422):
    """BASELINE config 2: one chromosome (id 0) of length ``genome_len``."""
    return make_table(n, seed, kind, lengths=np.array([genome_len], dtype=np.int64))


def expected_pairs(n_a, n_b, mean_len_a, mean_len_b, lengths=HG38_LENGTHS) -> float:
    """E[P] ~ N_A N_B (L_A + L_B) sum_c (G_c/G)^2 / G_c = N_A N_B (L_A+L_B) / G."""
    g = float(np.asarray(lengths, dtype=np.float64).sum())
    return n_a * n_b * (mean_len_a + mean_len_b) / g
