"""giql plugin: ``dialect="hip"`` registered through giql's OWN extension hook.

Importing this module where the reference package (``giql`` + ``sqlglot``) is importable
registers ``(HipTarget, Intersects)`` on giql's process-wide registry
(``src/giql/expander.py:499-546``), which also declares the target name so
``giql.transpile(..., dialect="hip")`` resolves it (``expander.py:336-362``;
``src/giql/targets.py:219-226``).

The expander mirrors ``expand_intersects_duckdb``
(``src/giql/expanders/intersects_duckdb.py:1674-1715``): for a column-to-column INTERSECTS join
whose whole-query shape the path supports it installs a statement finalizer that replaces the root
with ``exp.Command(this=<plan string>)`` -- the verbatim-payload precedent of ``:1713`` -- so
``transpile()`` still returns ``str``; every other VALID shape defers to ``_expand_spatial_op``
(``:1715``); user mistakes raise ``ValueError`` as they do upstream (see "Errors" below).

It lowers FROM THE NODE AND THE CONTEXT, not from re-serialised SQL text:

* the two operands come from ``ctx.resolution.column("this" / "expression")``
  (``src/giql/resolver.py:256-299, 334-389``): alias-qualified physical columns.  By pass 3 the
  canonicalizer has already rewritten the fragments of a non-canonical table to ``(a."start" - 1)``
  / ``(a."end" + 1)`` and blanked ``table`` (``canonicalizer.py:320-378``), so the canonical offsets
  are READ OFF those fragments -- never re-derived from ``ctx.tables`` on top of them (that would
  apply a ``- 1`` twice).  An operand the resolver left out falls back to ``ctx.tables`` by table
  name, exactly as ``_build_sql`` does (``intersects_duckdb.py:1179-1225``);
* the statement's clauses are read from the AST by node ``key`` and ``args`` only (no sqlglot class
  is needed to READ a tree), into the neutral :class:`giql_amd.shape.JoinShape`;
* ONE gate, :func:`giql_amd.shape.lower_join_shape`, shared with the sqlglot-free mirror
  (:mod:`giql_amd.transpile`), decides accept / decline / error;
* every node it reads is checked against a WHITELIST of the args it understands (:func:`_only`): a clause
  sqlglot stores under any other key declines instead of being dropped.

Errors -- which is which.  Shapes that are merely unsupported HERE raise :class:`HipDeclined`, which the
expander catches: the naive predicate then runs the query, exactly as the reference's ``_DeclineIEJoin``
does (``intersects_duckdb.py:797-801``).  A plain ``ValueError`` is kept for what the reference rejects as
well and surfaces from ``giql.transpile`` there too (``_UnqualifiedProjectionError`` -> ``ValueError``,
``:803-804, 925-958``): an unqualified or unknown-qualifier column, a column carrying a catalog / schema
qualifier, a right-side column under SEMI / ANTI, an unqualified HAVING / ORDER BY name that is no output
column (``giql_amd.shape.lower_join_shape``).  So the expander does let ``ValueError`` through -- by design.

Because nothing here needs sqlglot to run, the expander is exercised on CPU with plain stand-ins
for ``node`` / ``ExpansionContext`` / ``OperatorResolution`` (``tests/test_plugin_doubles.py``).
"""

from __future__ import annotations

import re
from dataclasses import dataclass

from .plan import PlanSide
from .shape import (AGG_FUNCS, ColRef, HipDeclined, JoinShape, OrderKey, SelItem, TableRef, condition_terms, decline,
                    lower_join_shape, norm)

SPATIAL_KEYS = ("intersects", "contains", "within", "spatialsetpredicate")
SPATIAL_PREDICATE_META = "giql_spatial_predicate"   # what the generic spatial expanders stamp on their output
_CMP = {"eq": "=", "neq": "!=", "gt": ">", "gte": ">=", "lt": "<", "lte": "<="}
_ENCODING_OF_OFFSETS = {(0, 0): ("0based", "half_open"), (0, 1): ("0based", "closed"),
                        (-1, -1): ("1based", "half_open"), (-1, 0): ("1based", "closed")}


# ------------------------------------------------------------- reading a tree by key / args
def _key(n) -> str:
    return getattr(n, "key", "") if n is not None else ""


def _arg(n, *names):
    a = getattr(n, "args", None) or {}
    for k in names:
        if a.get(k) is not None:
            return a[k]
    return None


def _only(n, handled, what: str) -> None:
    """Whitelist guard: every NON-EMPTY arg of node ``n`` must be one this lowering reads.

    The lowering reads the statement by ``args`` key, so a clause sqlglot stores under a key it does not
    know (QUALIFY, GROUP BY ROLLUP / CUBE / GROUPING SETS / ALL, TABLESAMPLE, PIVOT, LATERAL VIEW, named
    WINDOWs, ``LIMIT ... OFFSET`` inside Limit, join hints, ...) would otherwise be dropped silently and a
    plan emitted that ignores it.  The reference carries such clauses verbatim in its outer
    wrapper; this target declines them, so the naive predicate runs the query instead."""
    for k, v in (getattr(n, "args", None) or {}).items():
        if k in handled or v is None or v is False or v == [] or v == () or v == "":
            continue
        raise decline(f"{what}: {k} is not supported by dialect='hip'")


def _ident(n) -> tuple[str, bool]:
    """(text, quoted) of an Identifier node (or a plain string)."""
    if n is None:
        return "", False
    if isinstance(n, str):
        return n, False
    if _key(n) == "identifier":
        return str(_arg(n, "this")), bool(_arg(n, "quoted"))
    return str(_arg(n, "this") or ""), False


def _children(n):
    for v in (getattr(n, "args", None) or {}).values():
        if isinstance(v, (list, tuple)):
            for x in v:
                if hasattr(x, "args"):
                    yield x
        elif hasattr(v, "args"):
            yield v


def _walk(n):
    yield n
    for c in _children(n):
        yield from _walk(c)


def _root(n):
    while getattr(n, "parent", None) is not None:
        n = n.parent
    return n


def _colref(n) -> ColRef:
    """Column node -> ColRef (``a.*`` parses as a Column whose ``this`` is a Star)."""
    if _key(n) == "star":
        return ColRef(None, False, "*", star=True)
    if _arg(n, "db") is not None or _arg(n, "catalog") is not None:
        # a user mistake the reference rejects as well (``_UnqualifiedProjectionError`` -> ValueError at
        # intersects_duckdb.py:803-804, 951-958): the qualifier would survive the alias rewrite and address
        # another relation.  NOT a decline.
        raise ValueError("a column with a catalog / schema qualifier cannot be attributed to a join side; "
                         "it must be qualified with the table alias only")
    t, tq = _ident(_arg(n, "table"))
    this = _arg(n, "this")
    if _key(this) == "star":
        return ColRef(t or None, tq, "*", star=True)
    c, _cq = _ident(this)
    return ColRef(t or None, tq, c)


def _tableref(n) -> TableRef:
    if _key(n) != "table":
        raise decline("join operand that is not a base table")
    name, _q = _ident(_arg(n, "this"))
    if not name or _key(_arg(n, "this")) not in ("identifier", ""):
        raise decline("table function / unnamed relation as a join operand")   # e.g. DISJOIN(genes)
    if _arg(n, "db") is not None or _arg(n, "catalog") is not None:
        raise decline("catalog/schema-qualified tables")
    _only(n, ("this", "alias", "db", "catalog"), "table operand")   # TABLESAMPLE, PIVOT, hints, time travel ...
    alias_node = _arg(n, "alias")
    if alias_node is not None:
        if _key(alias_node) == "tablealias":
            _only(alias_node, ("this",), "table alias")               # an alias with a column list
        a, aq = _ident(_arg(alias_node, "this") if _key(alias_node) == "tablealias" else alias_node)
        if a:
            return TableRef(name, a, aq)
    return TableRef(name, name, _q)


def _literal(n):
    if _key(n) == "neg" and _key(_arg(n, "this")) == "literal" and not _arg(_arg(n, "this"), "is_string"):
        v = _literal(_arg(n, "this"))
        return ("lit", -v[1])
    if _key(n) != "literal":
        return None
    text = str(_arg(n, "this"))
    if _arg(n, "is_string"):
        return ("lit", text)
    return ("lit", float(text) if "." in text else int(text))


_ARITH = {"add": "+", "sub": "-", "mul": "*", "div": "/"}


def _operand(n):
    """A comparison operand: a literal, a column, or arithmetic over them (``+ - * /``, unary minus, parentheses,
    LEAST / GREATEST / ABS: the overlap-fraction recipes, docs/recipes/intersect.rst:144-190) -> ``("lit", v)`` |
    ``("col", ColRef)`` | ``("fn", op, [operands])``."""
    lit = _literal(n)
    if lit is not None:
        return lit
    k = _key(n)
    if k == "column":
        return ("col", _colref(n))
    if k == "boolean":
        raise decline("boolean literal in a join condition")
    if k in _ARITH:
        _only(n, ("this", "expression"), "arithmetic")     # (Div carries typed / safe flags when a dialect sets them)
        return ("fn", _ARITH[k], [_operand(_arg(n, "this")), _operand(_arg(n, "expression"))])
    if k == "neg":
        _only(n, ("this",), "unary minus")
        return ("fn", "neg", [_operand(_arg(n, "this"))])
    if k == "paren":
        _only(n, ("this",), "parentheses")
        return _operand(_arg(n, "this"))
    if k in ("least", "greatest"):
        _only(n, ("this", "expressions"), k.upper())
        return ("fn", k, [_operand(_arg(n, "this"))] + [_operand(e) for e in (_arg(n, "expressions") or [])])
    if k == "abs":
        _only(n, ("this",), "ABS")
        return ("fn", "abs", [_operand(_arg(n, "this"))])
    raise decline(f"join condition operand of kind {k!r}")


def _cond_tree(n, operand=None):
    """A condition node -> the tree ``shape.condition_terms`` normalises (AND / OR / NOT / parentheses over
    INTERSECTS, comparisons, BETWEEN, IN (literals), IS [NOT] NULL).  The reference inlines any extra that
    holds no INTERSECTS / sub-query / aggregate / window as SQL text (``_classify_extras``,
    intersects_duckdb.py:889-912); what has no evaluator here -- LIKE, arithmetic, functions, sub-queries,
    TRUE -- declines, so the naive predicate runs the query."""
    k = _key(n)
    operand = operand or _operand
    if k in ("and", "or"):
        _only(n, ("this", "expression"), k.upper())
        return (k, [_cond_tree(_arg(n, "this"), operand), _cond_tree(_arg(n, "expression"), operand)])
    if k == "not":
        _only(n, ("this",), "NOT")
        return ("not", _cond_tree(_arg(n, "this"), operand))
    if k == "paren":
        _only(n, ("this",), "parentheses")
        return _cond_tree(_arg(n, "this"), operand)
    if k == "intersects":
        l, r = _arg(n, "this"), _arg(n, "expression")
        if _key(l) != "column" or _key(r) != "column":
            raise decline("INTERSECTS operand that is not a column")
        return ("leaf", ("intersects", _colref(l), _colref(r)))
    if k in _CMP:
        return ("leaf", ("cmp", operand(_arg(n, "this")), _CMP[k], operand(_arg(n, "expression"))))
    if k == "between":
        _only(n, ("this", "low", "high"), "BETWEEN")
        x = operand(_arg(n, "this"))
        return ("and", [("leaf", ("cmp", x, ">=", operand(_arg(n, "low")))),
                        ("leaf", ("cmp", x, "<=", operand(_arg(n, "high"))))])
    if k == "in":
        _only(n, ("this", "expressions"), "IN")      # IN (sub-query) / IN UNNEST(...) carry other args
        x = operand(_arg(n, "this"))
        values = [_literal(v) for v in (_arg(n, "expressions") or [])]
        if not values or any(v is None for v in values):
            raise decline("IN list with a non-literal member")
        return ("or", [("leaf", ("cmp", x, "=", v)) for v in values])
    if k == "is":
        _only(n, ("this", "expression"), "IS")
        x = operand(_arg(n, "this"))
        if _key(_arg(n, "expression")) != "null" or x[0] == "lit":
            raise decline("IS predicate other than <column> IS [NOT] NULL")
        return ("leaf", ("cmp", x, "isnull", ("lit", 0)))
    if k in ("contains", "within"):
        raise decline(f"{k.upper()} predicate")
    raise decline(f"join condition of kind {k!r}")


def _terms(cond):
    return condition_terms(_cond_tree(cond), trees=True) if cond is not None else []


def _select_item(n) -> SelItem:
    alias = None
    if _key(n) == "alias":
        alias, _q = _ident(_arg(n, "alias"))
        n = _arg(n, "this")
    k = _key(n)
    if k == "column":
        return SelItem(_colref(n), alias)
    if k == "star":
        return SelItem(ColRef(None, False, "*", star=True), alias)
    if k.upper() in AGG_FUNCS:
        arg = _arg(n, "this")
        distinct = False
        if _key(arg) == "distinct":
            exprs = _arg(arg, "expressions") or []
            if len(exprs) != 1:
                raise decline("aggregate over several DISTINCT expressions")
            arg, distinct = exprs[0], True
        if _key(arg) == "star":
            if k != "count" or distinct:
                raise decline(f"{k.upper()}(*)")
            return SelItem(None, alias, "COUNT", False)
        if _key(arg) != "column":
            raise decline("aggregate over an expression")        # #204 / #205
        ref = _colref(arg)
        if ref.star:
            raise decline("star inside an aggregate")            # COUNT(a.*), #204
        if k == "count" and not distinct:
            ref.count = True
        return SelItem(ref, alias, k.upper(), distinct)
    # expressions, window aggregates, FILTER clauses, scalar sub-queries, literals (#204, #205)
    raise decline(f"projection of kind {k!r}")


# ------------------------------------------------------- the operands from the resolution
_FRAG = re.compile(r'^\(?\s*(?:"(?P<qa>(?:[^"]|"")+)"|(?P<a>[A-Za-z_][A-Za-z_0-9]*))\s*\.\s*'
                   r'(?:"(?P<qc>(?:[^"]|"")+)"|(?P<c>[A-Za-z_][A-Za-z_0-9]*))\s*(?:(?P<sign>[+-])\s*1\s*\))?\s*$')


def _parse_fragment(sql: str):
    """``a."start"`` / ``(a."start" - 1)`` / ``(a."end" + 1)`` -> (alias, column, delta)."""
    m = _FRAG.match(sql or "")
    if not m:
        raise decline(f"operand fragment {sql!r} is not an alias-qualified column")
    alias = m.group("qa").replace('""', '"') if m.group("qa") is not None else norm(m.group("a"))
    col = m.group("qc").replace('""', '"') if m.group("qc") is not None else m.group("c")
    delta = 0 if not m.group("sign") else (1 if m.group("sign") == "+" else -1)
    return alias, col, delta


def side_from_resolution(resolved, ref: TableRef, tables) -> PlanSide:
    """One operand of the join as a :class:`PlanSide`: physical columns and encoding from the
    resolver's ``ResolvedColumn`` when there is one, else from the table registry by name."""
    alias = norm(ref.alias, ref.alias_quoted)
    if resolved is None:
        from .shape import table_side

        return table_side(ref, tables)
    a1, chrom, d0 = _parse_fragment(resolved.chrom)
    a2, start, ds = _parse_fragment(resolved.start)
    a3, end, de = _parse_fragment(resolved.end)
    if d0 != 0 or len({a1, a2, a3}) != 1:
        raise decline("operand columns that do not share one alias")
    if a1 != alias:
        raise decline("resolved operand alias differs from the joined table's alias")
    table = getattr(resolved, "table", None)
    if table is not None:
        # still carrying its Table: the canonicalizer left it alone (already canonical); a table that
        # was NOT canonical arrives wrapped with `table` blanked, and its offsets are in the fragments
        enc = (table.coordinate_system, table.interval_type)
        if (ds, de) != (0, 0):
            raise decline("operand fragments wrapped although their table is still attached")
    else:
        enc = _ENCODING_OF_OFFSETS.get((ds, de))
        if enc is None:
            raise decline(f"operand offsets ({ds}, {de}) match no coordinate encoding")
    return PlanSide(table=ref.name, alias=alias, chrom_col=chrom, start_col=start, end_col=end,
                    coordinate_system=enc[0], interval_type=enc[1])


# --------------------------------------------------------------- AST -> JoinShape -> plan
def is_column_intersects(node) -> bool:
    l, r = _arg(node, "this"), _arg(node, "expression")
    return (_key(node) == "intersects" and _key(l) == "column" and _key(r) == "column"
            and _arg(l, "table") is not None and _arg(r, "table") is not None)


def has_sibling_spatial_predicate(node, root) -> bool:
    """Any OTHER spatial predicate in the statement (or a node a generic spatial expander already
    rewrote): the join must defer to the naive predicate (intersects_duckdb.py:1650-1671)."""
    for cand in _walk(root):
        if cand is node:
            continue
        if _key(cand) in SPATIAL_KEYS:
            return True
        if (getattr(cand, "meta", None) or {}).get(SPATIAL_PREDICATE_META):
            return True
    return False


_SELECT_ARGS = ("with", "with_", "kind", "expressions", "distinct", "from", "from_", "joins", "where", "group", "having",
                "order", "limit", "offset")


def shape_from_ast(root, node, ctx) -> JoinShape:
    if _arg(root, "with_", "with") is not None:
        raise decline("top-level WITH")
    _only(root, _SELECT_ARGS, "SELECT")   # QUALIFY, WINDOW, LATERAL VIEW, PIVOT, TABLESAMPLE, INTO, hints, locks ...
    if _arg(root, "kind"):
        raise decline("SELECT AS STRUCT / VALUE")
    distinct_node = _arg(root, "distinct")
    if distinct_node is not None and _arg(distinct_node, "on") is not None:
        raise decline("DISTINCT ON")
    if distinct_node is not None:
        _only(distinct_node, ("on",), "DISTINCT")
    from_node = _arg(root, "from_", "from")
    if from_node is None:
        raise decline("no FROM clause")
    _only(from_node, ("this",), "FROM")
    from_ref = _tableref(_arg(from_node, "this"))
    joins = _arg(root, "joins") or []
    if len(joins) != 1:
        raise decline("a third table" if len(joins) > 1 else "no join (a single-table predicate)")
    j = joins[0]
    _only(j, ("this", "on", "side", "kind", "method", "using"), "JOIN")   # hints, ASOF match_condition, GLOBAL ...
    if _arg(j, "method"):
        raise decline("NATURAL join")
    side = str(_arg(j, "side") or "").upper()
    kind = str(_arg(j, "kind") or "").upper()
    if side in ("RIGHT", "FULL"):
        raise decline(f"{side} outer join")
    if kind in ("SEMI", "ANTI"):
        jkind = kind
    elif side == "LEFT":
        jkind = "LEFT"
    elif kind in ("", "INNER", "CROSS"):
        jkind = "INNER"
    else:
        raise decline(f"{kind} join")
    join_ref = _tableref(_arg(j, "this"))
    using = [_ident(u)[0] for u in (_arg(j, "using") or [])]
    on = _arg(j, "on")
    where = _arg(root, "where")
    shape = JoinShape(items=[_select_item(e) for e in (_arg(root, "expressions") or [])], from_ref=from_ref,
                      join_ref=join_ref, kind=jkind, on_seen=on is not None, using=using,
                      distinct=distinct_node is not None)
    shape.on_terms = _terms(on)
    shape.where_terms = _terms(_arg(where, "this") if where is not None else None)
    if where is not None:
        _only(where, ("this",), "WHERE")
    group = _arg(root, "group")
    if group is not None:
        _only(group, ("expressions",), "GROUP BY")   # ROLLUP / CUBE / GROUPING SETS / ALL / WITH TOTALS
        for g in _arg(group, "expressions") or []:
            if _key(g) != "column":
                raise decline("GROUP BY expression")
            shape.group_by.append(_colref(g))
    having = _arg(root, "having")
    if having is not None:
        _only(having, ("this",), "HAVING")
        # the reference hands the clause to the engine verbatim (intersects_duckdb.py:1374-1380); here: a
        # conjunction of comparisons between plain aggregates / key columns and literals
        def having_operand(n):
            if _key(n).upper() in AGG_FUNCS:
                it = _select_item(n)
                if it.ref is not None:
                    it.ref.count = False
                return ("agg", it)
            if _key(n) in ("subquery", "select", "paren"):
                raise decline("parenthesised / sub-query HAVING condition")
            return _operand(n)

        shape.having = condition_terms(_cond_tree(_arg(having, "this"), having_operand))
        if any(t[0] not in ("cmp", "or") for t in shape.having):
            raise decline("spatial predicate in HAVING")
    order = _arg(root, "order")
    if order is not None:
        _only(order, ("expressions",), "ORDER BY")   # ORDER SIBLINGS BY
        for o in _arg(order, "expressions") or []:
            if _key(o) == "ordered":
                _only(o, ("this", "desc", "nulls_first"), "ORDER BY key")   # WITH FILL
            target = _arg(o, "this") if _key(o) == "ordered" else o
            if _key(target) != "column":
                raise decline("ORDER BY expression")    # incl. sub-queries (intersects_duckdb.py:690-701)
            ordered = _key(o) == "ordered"
            # sqlglot's parser fills nulls_first by the dialect's NULL ordering when the query does not say
            # (giql's dialect: NULLs are small), so what the node holds IS the placement; absent: the same default
            nf = _arg(o, "nulls_first") if ordered else None
            shape.order_by.append(OrderKey(_colref(target), bool(_arg(o, "desc")) if ordered else False,
                                           None if nf is None else bool(nf)))
    for clause in ("limit", "offset"):
        c = _arg(root, clause)
        if c is not None:
            _only(c, ("expression",), clause.upper())   # LIMIT a, b / LIMIT ... BY / WITH TIES / PERCENT
            v = _literal(_arg(c, "expression"))
            if v is None or not isinstance(v[1], int):
                raise decline(f"{clause.upper()} that is not an integer literal")
            setattr(shape, clause, v[1])
    # the operands as pass 1 / pass 2 left them on the operator node
    res = getattr(ctx, "resolution", None)
    by_alias = {}
    for arg in ("this", "expression"):
        rc = res.column(arg) if res is not None else None
        if rc is not None:
            by_alias[_parse_fragment(rc.chrom)[0]] = rc
    tables = ctx.tables
    shape.sides = (side_from_resolution(by_alias.get(norm(from_ref.alias, from_ref.alias_quoted)), from_ref, tables),
                   side_from_resolution(by_alias.get(norm(join_ref.alias, join_ref.alias_quoted)), join_ref, tables))
    return shape


def lower_statement(root, node, ctx):
    """The statement around a column-to-column INTERSECTS -> :class:`JoinPlan` (or HipDeclined)."""
    return lower_join_shape(shape_from_ast(root, node, ctx), ctx.tables)


def make_expander(fallback, make_command):
    """The ``(HipTarget, Intersects)`` expander with its two giql-side effects injected:
    ``fallback(node, ctx)`` = ``_expand_spatial_op(node, ctx, "intersects")`` and
    ``make_command(payload)`` = ``exp.Command(this=payload)``."""

    def expand_intersects_hip(node, ctx):
        if is_column_intersects(node):
            root = _root(node)
            if _key(root) == "select" and not has_sibling_spatial_predicate(node, root):
                try:
                    plan = lower_statement(root, node, ctx)
                except HipDeclined:
                    plan = None     # valid GIQL this target does not run: the naive predicate handles it
                if plan is not None:
                    payload = plan.to_string()
                    ctx.add_statement_finalizer(lambda _root: make_command(payload))
                    return node
        return fallback(node, ctx)

    return expand_intersects_hip


try:
    from giql.expander import register
    from giql.expanders.intersects import _expand_spatial_op
    from giql.expressions import Intersects
    from giql.targets import Capabilities, Target
    from sqlglot import exp

    HAVE_GIQL = True
except Exception:  # ImportError, or giql failing to import without sqlglot
    HAVE_GIQL = False

if HAVE_GIQL:  # the registration itself needs the real package

    @dataclass(frozen=True)
    class HipTarget(Target):
        """MI355X HIP execution target (declared like ``targets.py:57-73``)."""

        name: str = "hip"
        sqlglot_dialect: str | None = None
        capabilities: Capabilities = Capabilities(
            supports_lateral=True, supports_star_replace=False, supports_qualify=False)

    expand_intersects_hip = register(HipTarget, Intersects)(
        make_expander(lambda node, ctx: _expand_spatial_op(node, ctx, "intersects"),
                      lambda payload: exp.Command(this=payload)))
