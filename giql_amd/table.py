"""Table schema configuration -- same fields, defaults and validation as the
reference's ``giql.Table`` (``src/giql/table.py:16-136``,
``src/giql/constants.py:7-11``), so a user's ``tables=[...]`` argument carries over
unchanged.  sqlglot-free.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

DEFAULT_CHROM_COL = "chrom"
DEFAULT_START_COL = "start"
DEFAULT_END_COL = "end"
DEFAULT_STRAND_COL = "strand"
DEFAULT_GENOMIC_COL = "interval"


@dataclass
class Table:
    """Genomic table configuration (mirror of ``giql.table.Table``)."""

    name: str
    genomic_col: str = DEFAULT_GENOMIC_COL
    chrom_col: str = DEFAULT_CHROM_COL
    start_col: str = DEFAULT_START_COL
    end_col: str = DEFAULT_END_COL
    strand_col: str | None = DEFAULT_STRAND_COL
    coordinate_system: Literal["0based", "1based"] = "0based"
    interval_type: Literal["half_open", "closed"] = "half_open"

    def __post_init__(self) -> None:
        if self.coordinate_system not in ("0based", "1based"):
            raise ValueError(
                f"coordinate_system must be '0based' or '1based', "
                f"got {self.coordinate_system!r}")
        if self.interval_type not in ("half_open", "closed"):
            raise ValueError(
                f"interval_type must be 'half_open' or 'closed', "
                f"got {self.interval_type!r}")

    @property
    def encoding(self) -> tuple[str, str]:
        return (self.coordinate_system, self.interval_type)


def encoding_of(table) -> tuple[str, str]:
    """``(coordinate_system, interval_type)`` of a Table -- this module's or the reference's own
    ``giql.table.Table`` (same two fields, ``src/giql/table.py:16-136``).  A helper instead of a property
    patched onto the upstream class: importing this package changes nothing in ``giql``."""
    return (table.coordinate_system, table.interval_type)


try:  # where the reference package is importable its own Table IS the schema type (same fields and
    # validation: src/giql/table.py:16-136); the mirror above serves boxes without giql / sqlglot
    from giql.table import Table as _GiqlTable

    if all(hasattr(_GiqlTable("t"), f) for f in ("genomic_col", "chrom_col", "start_col", "end_col", "strand_col",
                                                  "coordinate_system", "interval_type")):
        Table = _GiqlTable  # noqa: F811
except ImportError:  # no giql (or giql failing to import without sqlglot): the mirror above
    pass


class Tables:
    """Container for Table configurations (mirror of ``giql.table.Tables``)."""

    def __init__(self) -> None:
        self._tables: dict[str, Table] = {}

    def register(self, name: str, table: Table) -> None:
        self._tables[name] = table

    def get(self, name: str) -> Table | None:
        return self._tables.get(name)

    def __contains__(self, name: str) -> bool:
        return name in self._tables

    def __iter__(self):
        return iter(self._tables.values())


def build_tables(tables) -> Tables:
    """``transpile()``'s ``tables`` argument -> Tables (``src/giql/transpile.py:217-242``)."""
    container = Tables()
    for item in tables or []:
        if isinstance(item, str):
            container.register(item, Table(item))
        elif isinstance(item, Table) or (hasattr(item, "name") and hasattr(item, "chrom_col")):
            container.register(item.name, item)  # a real giql.Table works too
        else:
            raise ValueError(f"tables entries must be str or Table, got {type(item).__name__}")
    return container
