"""The ``dialect="hip"`` lowering target: a serialisable join plan.

Where the reference's DuckDB override emits a verbatim SQL payload through
``exp.Command`` (``src/giql/expanders/intersects_duckdb.py:1713``), the hip target
emits this plan's string form; :func:`giql_amd.execute.execute` turns plan +
Arrow tables into one C-ABI call.  sqlglot-free.
"""

from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field

PLAN_PREFIX = "GIQL-HIP-PLAN/1 "

KINDS = ("INNER", "SEMI", "ANTI", "NEAREST", "COUNT", "CLUSTER", "MERGE", "FILTER")


@dataclass(frozen=True)
class PlanSide:
    """One operand: the table's physical columns and its coordinate encoding
    (what ``_build_sql`` reads from ``self.tables``, intersects_duckdb.py:1179-1188)."""

    table: str
    alias: str
    chrom_col: str = "chrom"
    start_col: str = "start"
    end_col: str = "end"
    coordinate_system: str = "0based"
    interval_type: str = "half_open"

    @property
    def encoding(self) -> tuple[str, str]:
        return (self.coordinate_system, self.interval_type)


@dataclass(frozen=True)
class Projection:
    side: str      # "l", "r", "distance" (NEAREST), "count" (COUNT aggregate); CLUSTER / MERGE:
                   # "star" (every table column), "cluster" (the id), "count" (COUNT(*))
    column: str
    name: str      # output column name


@dataclass(frozen=True)
class Operand:
    """One side of a residual comparison: a column of the left / right table
    (``kind`` "l" / "r", ``value`` = column name) or a literal ("int", "float", "str")."""

    kind: str
    value: object


@dataclass(frozen=True)
class Residual:
    """An extra conjunct beside the INTERSECTS: ``lhs op rhs`` from the ON or the
    WHERE clause (the reference inlines these into the per-chromosome join,
    intersects_duckdb.py:1157-1177, 1239-1243)."""

    clause: str    # "on" | "where"
    lhs: Operand
    op: str        # = != < <= > >= isnull notnull (the last two read lhs only)
    rhs: Operand
    group: int = 0  # residuals are AND-ed; those sharing a non-zero group are OR-ed (one CNF clause)


@dataclass(frozen=True)
class Aggregate:
    """A plain aggregate of the outer SELECT (``FUNC([DISTINCT] <col>)`` or ``COUNT(*)``), computed
    over the join's rows per ``group_by`` key (the reference rebuilds these over its wrapper
    relation, intersects_duckdb.py:1402-1644)."""

    func: str      # COUNT SUM MIN MAX AVG
    side: str      # "l" / "r", or "*" for COUNT(*)
    column: str
    name: str      # output column name
    distinct: bool = False


@dataclass(frozen=True)
class Having:
    """One conjunct of the HAVING clause over the grouped result: ``lhs op rhs`` where an operand is an
    output column of the grouping (``kind`` "name": a key, an aggregate of the SELECT list, or a hidden
    ``__giql_h<n>`` aggregate computed for this clause only) or a literal ("int", "float", "str")."""

    lhs: Operand
    op: str        # = != < <= > >= isnull notnull
    rhs: Operand
    group: int = 0  # conjuncts sharing a non-zero group are OR-ed (one clause of the normal form)


@dataclass(frozen=True)
class JoinPlan:
    kind: str
    left: PlanSide
    right: PlanSide | None      # None for the single-table operators (CLUSTER / MERGE / FILTER)
    projection: tuple[Projection, ...] = field(default_factory=tuple)
    distinct: bool = False
    # NEAREST only (src/giql/expanders/nearest.py:240-252)
    k: int = 1
    max_distance: int | None = None
    signed: bool = False
    residuals: tuple[Residual, ...] = field(default_factory=tuple)
    # CLUSTER / MERGE only (src/giql/expanders/cluster.py:222-243)
    distance: int = 0
    stranded: bool = False
    strand_col: str | None = None
    # CLUSTER(..., predicate := <comparison> [AND ...]) (src/giql/expanders/cluster.py:281-296): operand kind "l" = the
    # current row's column, "r" = PREV(column) -- the sorted predecessor's -- or a literal; clause "predicate"
    cluster_predicate: tuple[Residual, ...] = field(default_factory=tuple)
    # the clauses that ride on the reference's outer SELECT wrapper (intersects_duckdb.py:1336-1400),
    # finished on the projected table: projection columns named "__giql_*" are carried for them only
    aggregates: tuple[Aggregate, ...] = field(default_factory=tuple)
    group_by: tuple[str, ...] = field(default_factory=tuple)       # output names of the key columns
    having: tuple[Having, ...] = field(default_factory=tuple)      # conjuncts over the grouped result
    # (output name, descending, NULLs first): the placement is explicit per key -- where the query does not
    # write NULLS FIRST / LAST it is giql's dialect default, "NULLs are small" (first when ascending, last
    # when descending), which the reference's emitted DuckDB SQL spells out the same way
    order_by: tuple[tuple[str, bool, bool], ...] = field(default_factory=tuple)
    limit: int | None = None
    offset: int | None = None
    output: tuple[str, ...] = field(default_factory=tuple)         # final column names in SELECT order (grouped plans)

    def __post_init__(self) -> None:
        if self.kind not in KINDS:
            raise ValueError(f"unknown plan kind {self.kind!r}")

    def to_string(self) -> str:
        d = asdict(self)
        d["projection"] = [asdict(p) for p in self.projection]
        d["residuals"] = [asdict(r) for r in self.residuals]
        d["cluster_predicate"] = [asdict(r) for r in self.cluster_predicate]
        d["aggregates"] = [asdict(a) for a in self.aggregates]
        d["having"] = [asdict(h) for h in self.having]
        d["order_by"] = [list(o) for o in self.order_by]
        return PLAN_PREFIX + json.dumps(d, sort_keys=True, separators=(",", ":"))

    @classmethod
    def from_string(cls, text: str) -> "JoinPlan":
        if not isinstance(text, str) or not text.startswith(PLAN_PREFIX):
            raise ValueError("not a GIQL hip plan string")
        d = json.loads(text[len(PLAN_PREFIX):])
        return cls(
            kind=d["kind"], left=PlanSide(**d["left"]),
            right=PlanSide(**d["right"]) if d.get("right") else None,
            projection=tuple(Projection(**p) for p in d["projection"]),
            distinct=d.get("distinct", False), k=d.get("k", 1),
            max_distance=d.get("max_distance"), signed=d.get("signed", False),
            residuals=tuple(Residual(r["clause"], Operand(**r["lhs"]), r["op"], Operand(**r["rhs"]), r.get("group", 0))
                            for r in d.get("residuals", ())),
            distance=d.get("distance", 0), stranded=d.get("stranded", False), strand_col=d.get("strand_col"),
            cluster_predicate=tuple(Residual(r["clause"], Operand(**r["lhs"]), r["op"], Operand(**r["rhs"]), r.get("group", 0))
                                    for r in d.get("cluster_predicate", ())),
            aggregates=tuple(Aggregate(**a) for a in d.get("aggregates", ())),
            having=tuple(Having(Operand(**h["lhs"]), h["op"], Operand(**h["rhs"]), h.get("group", 0))
                         for h in d.get("having", ())),
            group_by=tuple(d.get("group_by", ())),
            order_by=tuple((o[0], bool(o[1]), bool(o[2]) if len(o) > 2 else not bool(o[1])) for o in d.get("order_by", ())),
            limit=d.get("limit"), offset=d.get("offset"), output=tuple(d.get("output", ())))


def is_plan_string(text) -> bool:
    return isinstance(text, str) and text.startswith(PLAN_PREFIX)
