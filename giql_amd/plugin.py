"""giql plugin: ``dialect="hip"`` registered through giql's OWN extension hook.

Import-guarded: needs the reference package (``giql`` + ``sqlglot``).  Importing
this module registers ``(HipTarget, Intersects)`` on giql's process-wide registry
(``src/giql/expander.py:499-546``), which also declares the target name so
``giql.transpile(..., dialect="hip")`` resolves it (``expander.py:336-362``;
``src/giql/targets.py:219-226``).

The expander mirrors ``expand_intersects_duckdb``
(``src/giql/expanders/intersects_duckdb.py:1674-1715``): for a column-to-column
INTERSECTS join whose whole-query shape the path supports it installs a statement
finalizer that replaces the root with ``exp.Command(this=<plan string>)`` -- the
verbatim-payload precedent of ``:1713`` -- so ``transpile()`` still returns ``str``;
every other shape defers to ``_expand_spatial_op`` (``:1715``), never errors.

Shape acceptance re-uses the sqlglot-free lowering of :mod:`giql_amd.transpile` on
the statement's own SQL text, so the plugin and the standalone mirror accept and
decline exactly the same queries.
"""

from __future__ import annotations

from dataclasses import dataclass

try:  # pragma: no cover - exercised only where giql + sqlglot are installed
    from giql.expander import ExpansionContext, register
    from giql.expanders.intersects import _expand_spatial_op
    from giql.expanders.intersects_duckdb import (
        _has_sibling_spatial_predicate,
        _is_column_intersects,
    )
    from giql.expressions import Intersects
    from giql.targets import Capabilities, Target
    from sqlglot import exp

    HAVE_GIQL = True
except Exception:  # ImportError, or giql failing to import without sqlglot
    HAVE_GIQL = False

from .transpile import HipDeclined, build_plan

if HAVE_GIQL:  # pragma: no cover

    @dataclass(frozen=True)
    class HipTarget(Target):
        """MI355X HIP execution target (declared like ``targets.py:57-73``)."""

        name: str = "hip"
        sqlglot_dialect: str | None = None
        capabilities: Capabilities = Capabilities(
            supports_lateral=True, supports_star_replace=False, supports_qualify=False)

    @register(HipTarget, Intersects)
    def expand_intersects_hip(node: "exp.Expression", ctx: "ExpansionContext") -> "exp.Expression":
        if isinstance(node, Intersects) and _is_column_intersects(node):
            root = node.root()
            if isinstance(root, exp.Select) and not _has_sibling_spatial_predicate(node, root):
                from giql.dialect import GIQLDialect

                try:
                    plan = build_plan(root.sql(dialect=GIQLDialect), ctx.tables)
                except HipDeclined:
                    plan = None
                if plan is not None:
                    payload = plan.to_string()
                    ctx.add_statement_finalizer(lambda _root: exp.Command(this=payload))
                    return node
        return _expand_spatial_op(node, ctx, "intersects")
