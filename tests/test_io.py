"""BED ingestion (pyarrow.csv) -- the tables execute() takes."""

import pytest

pa = pytest.importorskip("pyarrow")

from giql_amd.io import read_bed  # noqa: E402


def test_read_bed6_with_track_lines(tmp_path):
    p = tmp_path / "peaks.bed"
    p.write_text("track name=peaks\n# comment\nchr1\t100\t200\tp1\t10\t+\nchr2\t5\t9\tp2\t0\t-\n")
    t = read_bed(str(p))
    assert t.column_names == ["chrom", "start", "end", "name", "score", "strand"]
    assert t.schema.field("start").type == pa.int32() and t.schema.field("end").type == pa.int32()
    assert t.to_pylist() == [
        {"chrom": "chr1", "start": 100, "end": 200, "name": "p1", "score": 10, "strand": "+"},
        {"chrom": "chr2", "start": 5, "end": 9, "name": "p2", "score": 0, "strand": "-"}]


def test_read_bed3_custom_names_and_empty(tmp_path):
    p = tmp_path / "x.bed"
    p.write_text("chrX\t1\t2\nchrY\t3\t4\n")
    t = read_bed(str(p), columns=["contig", "lo", "hi"])
    assert t.column_names == ["contig", "lo", "hi"] and t.num_rows == 2
    with pytest.raises(ValueError):
        read_bed(str(p), columns=["a", "b"])
    e = tmp_path / "empty.bed"
    e.write_text("# nothing\n")
    assert read_bed(str(e)).num_rows == 0


def test_encode_chroms_rejects_a_null_anywhere_and_remaps_sparse_ids():
    import numpy as np
    import pytest

    from giql_amd.execute import encode_chroms

    a = np.array(["chr1", "chr2", "chr1"], dtype=object)
    ia, ib, d = encode_chroms(a, np.array(["chr2", "chrX"], dtype=object))
    assert d == ["chr1", "chr2", "chrX"] and ia.tolist() == [0, 1, 0] and ib.tolist() == [1, 2]
    for bad in (None, float("nan")):
        b = np.array(["chr2", "chrX", bad, "chr1"], dtype=object)   # NOT the first element
        with pytest.raises(ValueError, match="NULL"):
            encode_chroms(a, b)
        with pytest.raises(ValueError, match="NULL"):
            encode_chroms(b, a)
    # dense integer ids pass through; one huge id is remapped instead of sizing the per-chromosome arrays
    ia, ib, d = encode_chroms(np.array([0, 3, 1]), np.array([2, 3]))
    assert ia.tolist() == [0, 3, 1] and len(d) == 4
    ia, ib, d = encode_chroms(np.array([5, 2_000_000_000, 5]), np.array([7, 5]))
    assert d == [5, 7, 2_000_000_000] and ia.tolist() == [0, 2, 0] and ib.tolist() == [1, 0]
    with pytest.raises(ValueError, match="non-negative"):
        encode_chroms(np.array([-1, 2]), np.array([1]))
