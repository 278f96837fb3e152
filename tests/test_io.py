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
