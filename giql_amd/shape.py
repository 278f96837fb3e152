"""The shape gate of ``dialect="hip"``: ONE function over a small neutral description of a
two-table INTERSECTS statement, shared by the two front ends that can produce it --

* :mod:`giql_amd.transpile` (the sqlglot-free mirror: tokens -> :class:`JoinShape`), and
* :mod:`giql_amd.plugin` (giql's own ``@register(HipTarget, Intersects)`` hook: sqlglot AST +
  ``ExpansionContext`` -> :class:`JoinShape`),

so both accept, decline and reject exactly the same queries.  It restates the whitelist of
``IntersectsDuckDBIEJoinTransformer.transform_to_sql`` / ``_build_sql``
(``src/giql/expanders/intersects_duckdb.py:618-804, 1136-1400``):

* one column-to-column INTERSECTS between two distinct registered-or-default base tables, joined
  INNER / CROSS / comma / SEMI / ANTI (or the count_overlaps ``LEFT JOIN ... COUNT(b.col) ...
  GROUP BY`` shape, ``:432-548``), optionally ``USING (<chrom>)`` (``:727-735, 1190-1201``);
* residual conditions beside it -- comparisons over columns, literals and arithmetic in conjunctive normal form
  (ON residuals join, WHERE residuals filter: ``:889-912, 1164-1177``);
* a projection of qualified columns and plain aggregates (``:1402-1644``); stars, expressions,
  windows, FILTER, sub-queries decline (#202, #204, #205);
* DISTINCT / GROUP BY / ORDER BY / LIMIT / OFFSET over the result, which the reference lets "ride on
  the outer SELECT wrapper" (``:1336-1400``) and this target finishes on the projected Arrow table
  (:func:`giql_amd.execute.execute`); HAVING as a conjunction of comparisons between aggregates / keys
  and literals; DISTINCT ON, sub-queries and expressions decline.

Errors follow the reference's convention: user mistakes (unqualified / unknown-qualifier columns,
right-side columns under SEMI / ANTI) are ``ValueError``; valid GIQL this target does not run is
:class:`HipDeclined` (the reference *declines* such shapes to the naive predicate, ``:1715``).
"""

from __future__ import annotations

from dataclasses import dataclass, field

from .plan import Aggregate, Having, JoinPlan, Operand, PlanSide, Projection, Residual
from .table import Table, Tables


class HipDeclined(ValueError):
    """Valid GIQL that the hip dialect does not execute (reference: decline)."""


def decline(reason: str) -> HipDeclined:
    return HipDeclined(
        f"{reason}: this query shape is valid GIQL but is not executed by dialect='hip' "
        "(the reference declines it to the naive overlap predicate); transpile it with "
        "giql.transpile(...) for a SQL engine instead")


# ------------------------------------------------------- boolean conditions -> CNF
# Both front ends hand a condition over as a tree -- ("leaf", term) | ("and", [nodes]) |
# ("or", [nodes]) | ("not", node), a term being ("intersects", ...), ("intersects_lit", ...)
# or ("cmp", lhs, op, rhs) -- and get the terms of its conjunctive normal form back: the
# select kernel evaluates an AND of ORs (include/giql_hip.h, giql_pred.group).  The
# reference inlines such extras as SQL text (_classify_extras, intersects_duckdb.py:889-912).
NEGATED_OP = {"=": "!=", "!=": "=", "<": ">=", ">=": "<", ">": "<=", "<=": ">", "isnull": "notnull",
              "notnull": "isnull"}
MAX_CONDITION_LEAVES = 12   # a select call takes 16 predicates; a literal-range filter adds three of its own


def _nnf(node, neg: bool = False):
    """Push NOT down to the comparisons.  Exact under three-valued logic: a filter keeps TRUE
    only, De Morgan holds in Kleene logic and NOT (x < y) is TRUE exactly when x >= y is."""
    k = node[0]
    if k == "leaf":
        if not neg:
            return node
        t = node[1]
        if t[0] != "cmp":
            raise decline("NOT over a spatial predicate")
        return ("leaf", ("cmp", t[1], NEGATED_OP[t[2]], t[3]))
    if k == "not":
        return _nnf(node[1], not neg)
    kids = [_nnf(c, neg) for c in node[1]]
    return (("or" if k == "and" else "and") if neg else k, kids)


def _cnf(node) -> list:
    k = node[0]
    if k == "leaf":
        return [[node[1]]]
    if k == "and":
        return [cl for c in node[1] for cl in _cnf(c)]
    acc = [[]]
    for c in node[1]:
        acc = [a + cl for a in acc for cl in _cnf(c)]
        if sum(len(a) for a in acc) > 4 * MAX_CONDITION_LEAVES:
            raise decline("condition too large once normalised")
    return acc


def _bool_term(node):
    """A (NOT-free) condition node as an operand term ``("fn", op, [args])`` for the select kernel's boolean
    programs: comparisons, IS [NOT] NULL, AND / OR (``giql_hip_select_expr_dev``, GIQL_X_EQ .. GIQL_X_OR)."""
    k = node[0]
    if k == "leaf":
        t = node[1]
        if t[0] != "cmp":
            raise decline("spatial predicate under OR")
        if t[2] in ("isnull", "notnull"):
            return ("fn", t[2], [t[1]])
        return ("fn", t[2], [t[1], t[3]])
    return ("fn", k, [_bool_term(c) for c in node[1]])


def condition_terms(tree, trees: bool = False) -> list:
    """The condition's CNF as terms: a one-leaf clause is the leaf itself, a longer one
    ``("or", [cmp, ...])``.  A spatial predicate must be a conjunct of its own (under OR / NOT
    the reference falls back too: ``_classify_extras``).

    ``trees`` (the join's residuals, round 4): a condition whose normal form outgrows the cap is not declined --
    its top-level conjuncts stay as they are, a plain comparison a term as before, anything nested ONE term
    ``("tree", ("fn", op, [...]))`` that the select kernel evaluates as a boolean program under three-valued
    logic (the reference inlines such a condition as text, intersects_duckdb.py:889-957).  Small conditions keep
    their normal form: its clauses are what places a one-table condition BEFORE the join."""
    nnf = _nnf(tree)
    if trees:
        try:
            return condition_terms(tree)
        except HipDeclined as exc:
            if "too large" not in str(exc):
                raise
        out = []
        stack = [nnf]
        conjuncts = []
        while stack:                      # the top-level ANDs, in order
            n = stack.pop()
            if n[0] == "and":
                stack.extend(reversed(n[1]))
            else:
                conjuncts.append(n)
        for n in conjuncts:
            out.append(n[1] if n[0] == "leaf" else ("tree", _bool_term(n)))
        return out
    out, leaves = [], 0
    for clause in _cnf(nnf):
        uniq = []
        for t in clause:
            if t not in uniq:
                uniq.append(t)
        if len(uniq) == 1:
            out.append(uniq[0])
        else:
            if any(t[0] != "cmp" for t in uniq):
                raise decline("spatial predicate under OR")
            out.append(("or", uniq))
        leaves += sum(t[0] == "cmp" for t in uniq)
    if leaves > MAX_CONDITION_LEAVES:
        raise decline("condition too large once normalised")
    return out


def norm(name: str, quoted: bool = False) -> str:
    # unquoted identifiers are case-insensitive (intersects_duckdb.py:119-128)
    return name if quoted else name.casefold()


AGG_FUNCS = ("COUNT", "SUM", "MIN", "MAX", "AVG")


# ------------------------------------------------------------- the description
@dataclass
class ColRef:
    table: str | None
    table_quoted: bool
    column: str
    star: bool = False
    count: bool = False  # COUNT(<this column>) -- kept for the count_overlaps shape


@dataclass
class TableRef:
    name: str
    alias: str
    alias_quoted: bool = False


@dataclass
class SelItem:
    """One SELECT-list item: a column, or a plain aggregate ``FUNC([DISTINCT] <col> | *)``."""

    ref: ColRef | None           # the column / the aggregate's argument (None: COUNT(*))
    alias: str | None = None
    func: str | None = None      # None = plain column
    distinct: bool = False


@dataclass
class OrderKey:
    ref: ColRef                  # qualified column, or an output name (table None)
    desc: bool = False
    nulls_first: bool | None = None   # NULLS FIRST / NULLS LAST; None = not written (giql's dialect default: NULLs are small)


@dataclass
class JoinShape:
    """What either front end hands to :func:`lower_join_shape`."""

    items: list[SelItem]
    from_ref: TableRef
    join_ref: TableRef
    kind: str = "INNER"                      # INNER (also CROSS / comma) | SEMI | ANTI | LEFT
    on_seen: bool = False
    # ("intersects", ColRef, ColRef) | ("cmp", lhs, op, rhs) | ("or", [cmp, ...]) | ("tree", boolean program): see condition_terms
    on_terms: list = field(default_factory=list)
    where_terms: list = field(default_factory=list)
    using: list[str] = field(default_factory=list)
    distinct: bool = False
    group_by: list[ColRef] = field(default_factory=list)
    # HAVING terms ("cmp", lhs, op, rhs) | ("or", [cmp, ...]); an operand is ("lit", value), ("col", ColRef) or ("agg", SelItem)
    having: list = field(default_factory=list)
    order_by: list[OrderKey] = field(default_factory=list)
    limit: int | None = None
    offset: int | None = None
    # the two operands as resolved by the caller (the plugin reads them from
    # ``ctx.resolution.column(...)``); None = derive them from ``tables`` by table name
    sides: tuple[PlanSide, PlanSide] | None = None


# ------------------------------------------------------------------ helpers
def table_side(ref: TableRef, tables: Tables) -> PlanSide:
    t = tables.get(ref.name)
    if t is None:
        # an unregistered table uses default column names, like the naive plan
        # (intersects_duckdb.py:1179-1188)
        t = Table(ref.name)
    return PlanSide(table=ref.name, alias=norm(ref.alias, ref.alias_quoted),
                    chrom_col=t.chrom_col, start_col=t.start_col, end_col=t.end_col,
                    coordinate_system=t.coordinate_system, interval_type=t.interval_type)


def genomic_col(name: str, tables: Tables) -> str:
    t = tables.get(name)
    return t.genomic_col if t is not None else "interval"


def _side_of(ref: ColRef, left: PlanSide, right: PlanSide, where: str) -> str:
    if ref.table is None:
        raise ValueError(
            f"Unqualified column {ref.column!r} in {where}: the hip join path has no live schema to "
            "attribute it to a side; it must be qualified with a table alias")
    q = norm(ref.table, ref.table_quoted)
    if q == left.alias:
        return "l"
    if q == right.alias:
        return "r"
    raise ValueError(f"Unknown table qualifier {ref.table!r} in {where}")


def resolve_projection(items, left: PlanSide, right: PlanSide, left_only: bool,
                       distance_alias: str | None = None) -> tuple[Projection, ...]:
    out = []
    for it in items:
        ref = it.ref
        if it.func is not None or ref.count:
            raise decline("COUNT(...) outside the count_overlaps LEFT JOIN ... GROUP BY shape")
        if ref.star:
            # schema-less star enumeration would narrow the result (#202)
            raise decline("star projection")
        side = _side_of(ref, left, right, "the SELECT list")
        name = it.alias or ref.column
        if side == "l":
            out.append(Projection("l", ref.column, name))
        elif left_only:
            raise ValueError(
                f"Column {ref.table}.{ref.column} references the right side of a SEMI/ANTI "
                "join, which is out of scope in the SELECT list (left-side columns only)")
        elif distance_alias is not None and ref.column == "distance":
            out.append(Projection("distance", "distance", name))
        else:
            out.append(Projection("r", ref.column, name))
    return tuple(out)


COMPARISONS = ("=", "!=", "<", "<=", ">", ">=", "isnull", "notnull")


def bind_expression(o, bind_leaf) -> Operand:
    """An operand term -- ``("lit", v)`` / a column term / ``("fn", op, [terms])`` -- as a plan operand; columns and
    literals through ``bind_leaf``.  Arithmetic becomes ``Operand("expr", ["fn", op, [children]])`` whose leaves
    are ``[kind, value]`` pairs (lists throughout: the plan's JSON form gives lists back)."""
    if o[0] != "fn":
        return bind_leaf(o)

    def tree(t, strings_ok=False):
        if t[0] == "fn":
            cmp = t[1] in COMPARISONS       # (strings compare and can be NULL; they take no part in arithmetic)
            return ["fn", t[1], [tree(c, cmp) for c in t[2]]]
        leaf = bind_leaf(t)
        if leaf.kind == "str" and not strings_ok:
            raise ValueError(f"a string ({leaf.value!r}) cannot take part in arithmetic")
        return [leaf.kind, leaf.value]

    return Operand("expr", tree(o))


def operand_sides(o: Operand) -> set:
    """The tables ("l" / "r") an operand reads."""
    if o.kind in ("l", "r"):
        return {o.kind}
    if o.kind != "expr":
        return set()

    def walk(t):
        if t[0] == "fn":
            return set().union(*[walk(c) for c in t[2]]) if t[2] else set()
        return {t[0]} if t[0] in ("l", "r") else set()

    return walk(o.value)


#: what one call of the select kernel takes (``check_program`` in ``csrc/giql_hip.hip``: ``SEL_X_STACK`` values
#: live, 64 postfix nodes per call): the gate declines what the target cannot run instead of failing at run time
MAX_EXPR_DEPTH = 12
MAX_EXPR_NODES = 256


def expression_cost(o: Operand) -> tuple:
    """``(postfix nodes, values live at once)`` of an operand as ``HipEngine._flatten_expr`` lays it out: a leaf is
    one node and one value; an n-ary function folds pairwise (n - 1 operator nodes, its first argument's result
    stays on the stack while the next is evaluated); a unary one adds a node."""
    if o.kind != "expr":
        return (0, 0)

    def walk(t):
        if t[0] != "fn":
            return (1, 1)
        kids = [walk(c) for c in t[2]]
        if not kids:
            return (1, 1)
        nodes = sum(k[0] for k in kids) + max(len(kids) - 1, 1)
        depth = max([kids[0][1]] + [1 + k[1] for k in kids[1:]])
        return (nodes, depth)

    return walk(o.value)


def check_expression_sizes(residuals, kind: str) -> None:
    """Decline (never a run-time error) a condition whose arithmetic one select call cannot hold: an operand deeper
    than ``MAX_EXPR_DEPTH``, or more than ``MAX_EXPR_NODES`` nodes among the residuals ``execute()`` evaluates in
    one call -- those reading the same tables (left only: before the join; right only; both: on the pairs), and
    for SEMI / ANTI the WHERE residuals on their own."""
    per_call: dict = {}
    for r in residuals:
        nodes = 0
        for o in (r.lhs, r.rhs):
            n, d = expression_cost(o)
            if d > MAX_EXPR_DEPTH:
                raise decline(f"expression too deep for the select kernel ({d} values live, at most {MAX_EXPR_DEPTH})")
            nodes += n
        sides = frozenset(operand_sides(r.lhs) | operand_sides(r.rhs))
        call = ("where" if kind in ("SEMI", "ANTI") and r.clause == "where" else "on", sides)
        per_call[call] = per_call.get(call, 0) + nodes
        if per_call[call] > MAX_EXPR_NODES:
            raise decline(f"expressions too large for one select call ({per_call[call]} nodes, at most {MAX_EXPR_NODES})")


def resolve_residuals(clause_terms, left: PlanSide, right: PlanSide, kind: str) -> tuple:
    """``[(clause, term)]`` -> residuals.  A term is a comparison ``("cmp", lhs, op, rhs)`` or
    a disjunction ``("or", [cmp, ...])``; the members of one disjunction share a fresh group id."""
    out, group = [], 0
    for clause, t in clause_terms:
        if t[0] == "or":
            group += 1
            out.extend(resolve_residual(clause, leaf, left, right, kind, group) for leaf in t[1])
        elif t[0] == "tree":    # a nested condition as one boolean program: kept when it IS TRUE
            out.append(resolve_residual(clause, ("cmp", t[1], "istrue", ("lit", 0)), left, right, kind))
        else:
            out.append(resolve_residual(clause, t, left, right, kind))
    check_expression_sizes(out, kind)
    return tuple(out)


def resolve_residual(clause: str, term, left: PlanSide, right: PlanSide, kind: str, group: int = 0) -> Residual:
    """Bind a comparison's operands to the two sides; qualifier mistakes are user
    errors, as in ``_validate_extra_qualifiers`` (intersects_duckdb.py:914-959)."""
    _, lhs, op, rhs = term

    def bind(o) -> Operand:
        if o[0] == "lit":
            v = o[1]
            return Operand("str" if isinstance(v, str) else ("float" if isinstance(v, float) else "int"), v)
        ref: ColRef = o[1]
        if ref.star:
            raise decline("star in a join condition")
        if ref.table is None:
            raise ValueError(
                f"dialect='hip' cannot inline the extra predicate: column {ref.column!r} must be "
                f"qualified with {left.alias!r} or {right.alias!r}")
        q = norm(ref.table, ref.table_quoted)
        if q == left.alias:
            return Operand("l", ref.column)
        if q == right.alias:
            if kind in ("SEMI", "ANTI") and clause == "where":
                raise ValueError(f"{kind} join: the WHERE clause cannot reference the right side "
                                 f"({ref.table}.{ref.column})")
            return Operand("r", ref.column)
        raise ValueError(f"dialect='hip' cannot inline the extra predicate: unknown table qualifier "
                         f"{ref.table!r}; expected {left.alias!r} or {right.alias!r}")

    a, b = bind_expression(lhs, bind), bind_expression(rhs, bind)
    if not operand_sides(a) and not operand_sides(b):
        raise decline("constant predicate in the join condition")
    return Residual(clause, a, op, b, group)


def resolve_count_projection(items, group_cols, left: PlanSide, right: PlanSide):
    """The count_overlaps projection: left key columns + ONE aliased COUNT(<right col>);
    GROUP BY must be exactly the projected left columns (intersects_duckdb.py:484-539)."""
    out = []
    n_count = 0
    keys = set()
    for it in items:
        ref = it.ref
        is_count = it.func == "COUNT" or (ref is not None and ref.count)
        if it.func not in (None, "COUNT") or it.distinct:
            raise decline("count_overlaps with an aggregate other than COUNT(<right column>)")
        if ref is None or ref.star or ref.table is None:
            raise decline("count_overlaps projection that is not a qualified column")
        q = norm(ref.table, ref.table_quoted)
        if is_count:
            n_count += 1
            if q != right.alias:
                raise decline("COUNT over a left-side column")
            if not it.alias:
                raise decline("count_overlaps COUNT without an alias")
            out.append(Projection("count", ref.column, it.alias))
        else:
            if q != left.alias:
                raise decline("count_overlaps key from the right side")
            keys.add(ref.column)
            out.append(Projection("l", ref.column, it.alias or ref.column))
    if n_count != 1 or not keys:
        raise decline("count_overlaps needs left key columns and exactly one COUNT")
    gkeys = set()
    for g in group_cols:
        if g.table is None or norm(g.table, g.table_quoted) != left.alias:
            raise decline("GROUP BY column that is not a left-side qualified column")
        gkeys.add(g.column)
    if gkeys != keys:
        raise decline("GROUP BY keys differ from the projected left columns")
    names = [p.name for p in out]
    if len(set(names)) != len(names):
        raise decline("duplicate output names in count_overlaps")
    return tuple(out)


def _resolve_grouped(shape: JoinShape, left: PlanSide, right: PlanSide, left_only: bool):
    """Projection with plain aggregates and / or a GROUP BY: key columns + aggregates, finished on
    the projected table (the reference rebuilds them over the wrapper relation,
    intersects_duckdb.py:1402-1644).  Returns (projection, aggregates, group names)."""
    proj: list[Projection] = []
    aggs: list[Aggregate] = []
    names: list[str] = []
    key_names: dict[tuple[str, str], str] = {}
    for i, it in enumerate(shape.items):
        ref = it.ref
        if it.func is None:
            if ref.star:
                raise decline("star projection")
            side = _side_of(ref, left, right, "the SELECT list")
            if side == "r" and left_only:
                raise ValueError(f"Column {ref.table}.{ref.column} references the right side of a SEMI/ANTI join "
                                 "(left-side columns only)")
            name = it.alias or ref.column
            proj.append(Projection(side, ref.column, name))
            key_names[(side, ref.column)] = name
            names.append(name)
            continue
        if it.func not in AGG_FUNCS:
            raise decline(f"aggregate {it.func}")
        if it.distinct and it.func != "COUNT":
            raise decline(f"{it.func}(DISTINCT ...)")
        if ref is None:                       # COUNT(*)
            if it.func != "COUNT":
                raise decline(f"{it.func}(*)")
            side, column = "*", "*"
        else:
            if ref.star:
                raise decline("star inside an aggregate")   # #204
            side = _side_of(ref, left, right, "an aggregate argument")
            if side == "r" and left_only:
                raise ValueError(f"Column {ref.table}.{ref.column} references the right side of a SEMI/ANTI join "
                                 "(left-side columns only)")
            column = ref.column
        name = it.alias or f"{it.func.lower()}_{i}"
        aggs.append(Aggregate(it.func, side, column, name, it.distinct))
        names.append(name)
    if len(set(names)) != len(names):
        raise decline("duplicate output names")
    groups: list[str] = []
    for g in shape.group_by:
        side = _side_of(g, left, right, "GROUP BY")
        if side == "r" and left_only:
            raise ValueError(f"Column {g.table}.{g.column} references the right side of a SEMI/ANTI join")
        if (side, g.column) not in key_names:
            # a key that is not projected still groups: carried as a hidden column
            hidden = f"__giql_g{len(groups)}"
            proj.append(Projection(side, g.column, hidden))
            key_names[(side, g.column)] = hidden
        groups.append(key_names[(side, g.column)])
    plain = [p for p in proj if not p.name.startswith("__giql_g")]
    if aggs and any(p.name not in groups for p in plain):
        raise ValueError("a projected column must appear in GROUP BY or inside an aggregate")
    if not aggs and shape.group_by and any(p.name not in groups for p in plain):
        raise ValueError("a projected column must appear in GROUP BY or inside an aggregate")
    return tuple(proj), tuple(aggs), tuple(groups), tuple(names)


def _resolve_having(shape: JoinShape, proj, aggs, left: PlanSide, right: PlanSide, left_only: bool):
    """HAVING conjuncts -> comparisons over the grouped result's columns.  An aggregate that the SELECT
    list does not hold is added as a hidden ``__giql_h<n>`` aggregate (dropped from the output); a column
    must be a grouping key (by its qualified name or its output name).  Returns (aggregates, having)."""
    aggs = list(aggs)
    by_col = {(p.side, p.column): p.name for p in proj}
    out_names = {p.name for p in proj if not p.name.startswith("__giql_")} | {a.name for a in aggs}
    having: list[Having] = []

    def bind(o) -> Operand:
        if o[0] == "lit":
            v = o[1]
            return Operand("str" if isinstance(v, str) else ("float" if isinstance(v, float) else "int"), v)
        if o[0] == "col":
            ref: ColRef = o[1]
            if ref.star:
                raise decline("star in HAVING")
            if ref.table is None:
                if ref.column in out_names:
                    return Operand("name", ref.column)
                raise ValueError(f"HAVING {ref.column!r}: not an output column; qualify it with a table alias")
            side = _side_of(ref, left, right, "HAVING")
            name = by_col.get((side, ref.column))
            if name is None:
                raise ValueError(f"HAVING {ref.table}.{ref.column}: the column must appear in GROUP BY "
                                 "or inside an aggregate")
            return Operand("name", name)
        if o[0] == "fn":
            raise decline("arithmetic in HAVING")
        it: SelItem = o[1]
        if it.func not in AGG_FUNCS:
            raise decline(f"aggregate {it.func}")
        if it.distinct and it.func != "COUNT":
            raise decline(f"{it.func}(DISTINCT ...)")
        if it.ref is None:
            if it.func != "COUNT":
                raise decline(f"{it.func}(*)")
            side, column = "*", "*"
        else:
            if it.ref.star:
                raise decline("star inside an aggregate")
            side = _side_of(it.ref, left, right, "an aggregate argument")
            if side == "r" and left_only:
                raise ValueError(f"Column {it.ref.table}.{it.ref.column} references the right side of a SEMI/ANTI join "
                                 "(left-side columns only)")
            column = it.ref.column
        for a in aggs:
            if (a.func, a.side, a.column, a.distinct) == (it.func, side, column, it.distinct):
                return Operand("name", a.name)
        hidden = Aggregate(it.func, side, column, f"__giql_h{sum(a.name.startswith('__giql_h') for a in aggs)}",
                           it.distinct)
        aggs.append(hidden)
        return Operand("name", hidden.name)

    group = 0
    for term in shape.having:
        leaves, g = [term], 0
        if term[0] == "or":
            group += 1
            leaves, g = term[1], group
        for _tag, lhs, op, rhs in leaves:
            a, b = bind(lhs), bind(rhs)
            if a.kind != "name" and b.kind != "name":
                raise decline("constant HAVING predicate")
            having.append(Having(a, op, b, g))
    return tuple(aggs), tuple(having)


def _resolve_order(shape: JoinShape, proj, aggs, left: PlanSide, right: PlanSide, left_only: bool, grouped: bool):
    """ORDER BY keys -> output column names; a qualified column that is not projected rides along as
    a hidden column (dropped after the sort).  Returns (extra hidden projections, order spec)."""
    out_names = [p.name for p in proj] + [a.name for a in aggs]
    by_col = {(p.side, p.column): p.name for p in proj}
    hidden: list[Projection] = []
    order: list[tuple[str, bool, bool]] = []
    for k in shape.order_by:
        nulls_first = (not k.desc) if k.nulls_first is None else bool(k.nulls_first)
        ref = k.ref
        if ref.star:
            raise decline("ORDER BY *")
        if ref.table is None:
            if ref.column in out_names:        # an output name (alias)
                order.append((ref.column, k.desc, nulls_first))
                continue
            raise ValueError(f"ORDER BY {ref.column!r}: not an output column; qualify it with a table alias")
        side = _side_of(ref, left, right, "ORDER BY")
        if side == "r" and left_only:
            raise ValueError(f"ORDER BY {ref.table}.{ref.column} references the right side of a SEMI/ANTI join")
        name = by_col.get((side, ref.column))
        if name is None:
            if grouped or shape.distinct:
                raise decline("ORDER BY a column that is neither projected nor grouped")
            name = f"__giql_o{len(hidden)}"
            hidden.append(Projection(side, ref.column, name))
            by_col[(side, ref.column)] = name
        order.append((name, k.desc, nulls_first))
    return tuple(hidden), tuple(order)


# ------------------------------------------------------------------- the gate
def lower_join_shape(shape: JoinShape, tables: Tables) -> JoinPlan:
    """:class:`JoinShape` -> :class:`JoinPlan`, or :class:`HipDeclined` / ``ValueError``."""
    kind = shape.kind
    items = shape.items
    if shape.sides is not None:
        left, right = shape.sides
    else:
        left = table_side(shape.from_ref, tables)
        right = table_side(shape.join_ref, tables)
    has_count_item = any((it.func == "COUNT" and it.ref is not None and not it.distinct) for it in items)
    if kind == "LEFT":
        # count_overlaps: LEFT [OUTER] JOIN ... COUNT(b.col) ... GROUP BY left keys
        # (_match_count_overlaps, intersects_duckdb.py:432-548); every other outer join declines (:661-662)
        if not has_count_item:
            raise decline("LEFT outer join")
        kind = "COUNT"
    if kind == "COUNT" and not shape.on_seen:
        raise decline("count_overlaps without an ON clause")
    if shape.using:
        # the single-column form whose column is both tables' chrom column: the per-chromosome
        # partition IS that equi-join (intersects_duckdb.py:727-735, 1190-1201)
        if len(shape.using) != 1:
            raise decline("multi-column USING")
        u = shape.using[0].casefold()
        if left.chrom_col.casefold() != right.chrom_col.casefold() or left.chrom_col.casefold() != u:
            raise decline("USING on a column that is not both tables' chromosome column")
    on_terms, where_terms = list(shape.on_terms), list(shape.where_terms)
    if not shape.on_seen and not shape.using and kind in ("SEMI", "ANTI"):
        raise decline("SEMI/ANTI join with its INTERSECTS outside ON")  # #201
    n_int = sum(t[0] == "intersects" for t in on_terms + where_terms)
    if n_int == 0:
        raise decline("join without an INTERSECTS predicate")
    if n_int > 1:
        raise decline("more than one INTERSECTS")
    if kind in ("SEMI", "ANTI") and not any(t[0] == "intersects" for t in on_terms):
        raise decline("SEMI/ANTI join with its INTERSECTS outside ON")  # #201
    _, lhs, rhs = [t for t in on_terms + where_terms if t[0] == "intersects"][0]
    cmp_terms = ([("on", t) for t in on_terms if t[0] in ("cmp", "or", "tree")]
                 + [("where", t) for t in where_terms if t[0] in ("cmp", "or", "tree")])
    if kind == "COUNT" and (cmp_terms or where_terms):
        raise decline("count_overlaps with predicates beside the INTERSECTS")  # bare ON only (:432-548)
    if kind == "COUNT" and not shape.group_by:
        raise decline("count_overlaps without GROUP BY")

    for side_ref in (lhs, rhs):
        if side_ref.table is None or side_ref.star:
            raise decline("INTERSECTS operand that is not a table-qualified column")
    la, ra = norm(lhs.table, lhs.table_quoted), norm(rhs.table, rhs.table_quoted)
    if left.alias == right.alias:
        raise decline("same alias on both sides")
    if left.table == right.table:
        raise decline("self-join")
    # FROM-side orientation swap (intersects_duckdb.py:359-410)
    if la == left.alias and ra == right.alias:
        l_col, r_col = lhs.column, rhs.column
    elif ra == left.alias and la == right.alias:
        l_col, r_col = rhs.column, lhs.column
    else:
        raise decline("INTERSECTS operands that do not name the two joined tables")
    if l_col != genomic_col(left.table, tables) or r_col != genomic_col(right.table, tables):
        raise ValueError(
            f"INTERSECTS operands must be the tables' genomic columns "
            f"({genomic_col(left.table, tables)!r} / {genomic_col(right.table, tables)!r})")
    if kind == "COUNT":
        if shape.distinct:
            raise decline("DISTINCT with count_overlaps")
        if shape.order_by or shape.limit is not None or shape.offset is not None or shape.having:
            raise decline("HAVING / ORDER BY / LIMIT over count_overlaps")
        return JoinPlan("COUNT", left, right, resolve_count_projection(items, shape.group_by, left, right))

    left_only = kind in ("SEMI", "ANTI")
    grouped = bool(shape.group_by) or any(it.func is not None for it in items) or bool(shape.having)
    output: tuple[str, ...] = ()
    having: tuple[Having, ...] = ()
    if grouped:
        proj, aggs, groups, output = _resolve_grouped(shape, left, right, left_only)
        visible_aggs = aggs
        aggs, having = _resolve_having(shape, proj, aggs, left, right, left_only)
        if not visible_aggs and not groups and aggs:
            # HAVING over the whole result as ONE group, nothing aggregated in the SELECT list: the plain
            # columns it projects are neither grouped nor aggregated
            raise ValueError("a projected column must appear in GROUP BY or inside an aggregate")
    else:
        proj, aggs, groups = resolve_projection(items, left, right, left_only), (), ()
        visible_aggs = ()
    hidden, order = _resolve_order(shape, proj, visible_aggs, left, right, left_only, grouped)
    residuals = resolve_residuals(cmp_terms, left, right, kind)
    return JoinPlan(kind, left, right, tuple(proj) + hidden, shape.distinct, residuals=residuals,
                    aggregates=aggs, group_by=groups, having=having, order_by=order,
                    limit=shape.limit, offset=shape.offset, output=output)
