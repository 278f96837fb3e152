#!/usr/bin/env python3
"""Mint golden vectors for NEAREST k > 1 and ``stranded := true`` (tests/golden/nearest_k.json).

Run in the BUILD container only (the reference cannot travel to the GPU box; the vectors can).
Expected rows come from ``sqlite3`` executing the reference's own NEAREST SQL: the distance CASE is
produced by the reference's ``generate_distance_case`` -- src/giql/expanders/_distance.py:22-117,
loaded by file path (it imports nothing) -- inside the wrapper of src/giql/expanders/nearest.py:313-333,
387-396 (``WHERE ref.chrom = t.chrom [AND ref.strand = t.strand] [AND ABS(d) <= max_distance]
ORDER BY ABS(distance), start, end LIMIT k``), once per reference row.

Known answers transcribed from tests/integration/datafusion/test_cross_target_oracle.py (k = 2
:293-324, stranded :398-424, :482-520) head the file and double as the self-check of this script.
"""

import json
import os
import random
import sqlite3
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as M  # noqa: E402  (the shared sqlite helpers)

REF = "/root/reference"


def run(distance_mod, a_rows, b_rows, k, signed, max_distance, stranded):
    conn = sqlite3.connect(":memory:")
    conn.execute('CREATE TABLE genes (rid INTEGER, chrom TEXT, "start" INTEGER, "end" INTEGER, strand TEXT)')
    conn.executemany("INSERT INTO genes VALUES (?, ?, ?, ?, ?)",
                     [(i, r[0], r[1], r[2], r[3] if len(r) > 3 else None) for i, r in enumerate(b_rows)])
    out = []
    for row in a_rows:
        ac, as_, ae = row[:3]
        ref_chrom = "'" + ac + "'"
        ref_strand = ("'" + row[3] + "'") if (stranded and len(row) > 3 and row[3] is not None) else None
        case = distance_mod.generate_distance_case(
            ref_chrom, str(as_), str(ae), ref_strand, 'genes."chrom"', 'genes."start"', 'genes."end"',
            'genes."strand"' if stranded else None, stranded=stranded, signed=signed)
        where = [f'{ref_chrom} = genes."chrom"']
        if stranded and ref_strand:
            where.append(f'{ref_strand} = genes."strand"')
        if max_distance is not None:
            where.append(f"(ABS({case})) <= {max_distance}")
        inner = f'SELECT genes.*, {case} AS distance FROM genes WHERE {" AND ".join(where)}'
        sql = (f'SELECT x.rid, x."start", x."end", x."distance" FROM ({inner}) AS x '
               f'ORDER BY ABS(x."distance"), x."start", x."end" LIMIT {k}')
        out.append([list(r) for r in conn.execute(sql).fetchall()])
    conn.close()
    return out


def main():
    dm = M._load_by_path("_ref_distance", os.path.join(REF, "src/giql/expanders/_distance.py"))
    X = "tests/integration/datafusion/test_cross_target_oracle.py"
    cases = []

    def known(name, src, a, b, k, stranded, want_b_starts):
        got = run(dm, a, b, k, False, None, stranded)
        assert [[r[1] for r in rows] for rows in got] == want_b_starts, (name, got)
        cases.append({"name": name, "source": src, "a": a, "b": b, "k": k, "signed": False, "max_distance": None,
                      "stranded": stranded, "expected": got})

    known("k2_two_nearest", X + ":293-324", [["chr1", 200, 300]],
          [["chr1", 1000, 1100], ["chr1", 50, 60], ["chr1", 280, 290], ["chr1", 310, 320]], 2, False, [[280, 310]])
    known("stranded_matches_strand", X + ":398-424", [["chr1", 200, 300, "+"]],
          [["chr1", 280, 290, "+"], ["chr1", 250, 260, "-"]], 1, True, [[280]])
    known("stranded_opposite_strands_same_position", X + ":482-520", [["chr1", 200, 300, "+"], ["chr1", 200, 300, "-"]],
          [["chr1", 280, 290, "+"], ["chr1", 250, 260, "-"]], 1, True, [[280], [250]])

    rng = random.Random(20261004)
    idx = 0
    for k in (2, 3, 5):
        for signed in (False, True):
            for md in (None, 60):
                for stranded in (False, True):
                    for (na, nb, ms, ml) in [(10, 14, 400, 40), (12, 40, 300, 120), (6, 30, 40, 8)]:
                        def rows(n, chroms, min_len):
                            out = []
                            for _ in range(n):
                                c = rng.choice(chroms)
                                s = rng.randint(0, ms)
                                out.append([c, s, s + rng.randint(min_len, ml), rng.choice("+-")])
                            return out
                        a = rows(rng.randint(1, na), ["chr1", "chr2", "chr3"], 0)
                        b = rows(rng.randint(0, nb), ["chr1", "chr2"], 0)
                        if not stranded:
                            a = [r[:3] for r in a]
                            b = [r[:3] for r in b]
                        cases.append({"name": f"fuzz_nearest_k_{idx}", "source": "sqlite3 over the reference's distance CASE",
                                      "a": a, "b": b, "k": k, "signed": signed, "max_distance": md, "stranded": stranded,
                                      "expected": run(dm, a, b, k, signed, md, stranded)})
                        idx += 1
    # k beyond 64 (round 3: the kernel's walk does not depend on k; the cap is n_a * k < 2^31)
    rng2 = random.Random(20261007)
    for k, signed, md, stranded in [(70, False, None, False), (100, True, None, False), (150, False, 900, True)]:
        def rows2(n, chroms):
            out = []
            for _ in range(n):
                s = rng2.randint(0, 3000)
                out.append([rng2.choice(chroms), s, s + rng2.randint(0, 90), rng2.choice("+-")])
            return out
        a, b = rows2(12, ["chr1", "chr2", "chr3"]), rows2(260, ["chr1", "chr2"])
        if not stranded:
            a, b = [r[:3] for r in a], [r[:3] for r in b]
        cases.append({"name": f"large_k_{k}", "source": "sqlite3 over the reference's distance CASE",
                      "a": a, "b": b, "k": k, "signed": signed, "max_distance": md, "stranded": stranded,
                      "expected": run(dm, a, b, k, signed, md, stranded)})
    with open(os.path.join(HERE, "nearest_k.json"), "w") as f:
        json.dump({"_source": __doc__.strip().splitlines()[0] + " -- see tests/golden/make_nearest_k.py", "cases": cases}, f)
    print(f"wrote nearest_k.json: {len(cases)} cases")

    # stranded := true with '.' / '?' strands (round 3): the strand filter pairs a '.' reference row with the '.'
    # targets of its chromosome, the distance CASE yields NULL for them (_distance.py:88-117), so the k rows are the
    # first k by (start, end) with a NULL distance -- and none at all under max_distance (NULL <= d is not true)
    rng = random.Random(20261005)
    dot = []
    for idx, (k, signed, md) in enumerate([(1, False, None), (2, True, None), (3, False, None), (2, False, 500),
                                           (5, True, None), (4, False, None)]):
        def rows(n, chroms):
            out = []
            for _ in range(n):
                s = rng.randint(0, 600)
                out.append([rng.choice(chroms), s, s + rng.randint(0, 60), rng.choice("+-.?")])
            return out
        a = rows(rng.randint(6, 14), ["chr1", "chr2", "chr3"])
        b = rows(rng.randint(10, 40), ["chr1", "chr2"])
        dot.append({"name": f"dot_strands_{idx}", "source": "sqlite3 over the reference's distance CASE",
                    "a": a, "b": b, "k": k, "signed": signed, "max_distance": md, "stranded": True,
                    "expected": run(dm, a, b, k, signed, md, True)})
    assert any(r[3] is None for c in dot for rows_ in c["expected"] for r in rows_)
    with open(os.path.join(HERE, "nearest_dot_strands.json"), "w") as f:
        json.dump({"_source": "stranded NEAREST with '.' / '?' strands -- see tests/golden/make_nearest_k.py", "cases": dot}, f)
    print(f"wrote nearest_dot_strands.json: {len(dot)} cases")


if __name__ == "__main__":
    main()
