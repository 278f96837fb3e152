"""``transpile(giql, tables, dialect="hip")`` -- host-side mirror of the reference's
``giql.transpile`` (``src/giql/transpile.py:55-214``) for the one path this
backend executes: the column-to-column INTERSECTS join (INNER / SEMI / ANTI, with
residual conditions beside the INTERSECTS: comparisons over columns, literals and
arithmetic, combined with AND / OR / NOT, BETWEEN, IN, IS NULL), the count_overlaps
shape, the correlated NEAREST join (any k, stranded), CLUSTER / MERGE over one table
(with a ``predicate :=``), and the single-table literal-range filter.

The reference parses with sqlglot, which is not installable here, so this module
carries a small hand-written parser for exactly the query shapes the reference's
IEJoin override engages on (the whitelist of
``IntersectsDuckDBIEJoinTransformer.transform_to_sql``,
``src/giql/expanders/intersects_duckdb.py:618-804``).  Same call signature, same
``Table`` argument, same error convention:

* user mistakes (unqualified / unknown-alias columns, right-side columns under
  SEMI / ANTI)                                    -> ``ValueError``
  (``_UnqualifiedProjectionError`` -> ``ValueError``, intersects_duckdb.py:803-804);
* valid GIQL the hip path does not execute (stars, outer joins, self-joins, an INTERSECTS
  under OR / NOT, functions / LIKE / sub-queries in a condition, other aggregates,
  3+ tables, ...)                                                            -> :class:`HipDeclined`
  (a ``ValueError``): the reference *declines* such shapes to the naive predicate
  (intersects_duckdb.py:1715); without sqlglot there is no naive emitter to fall
  back to, so the caller is told to use ``giql.transpile`` for that query.

When the real ``giql`` package is importable, :mod:`giql_amd.plugin` registers the
same lowering through giql's own ``@register(HipTarget, Intersects)`` hook instead.

For ``dialect=None`` only the literal-range predicate of BASELINE config 1
(``WHERE interval INTERSECTS 'chr1:1000-2000'``) is emitted, with the reference's
exact text (``tests/expanders/test_intersects.py:83-85``).
"""

from __future__ import annotations

import re
from dataclasses import dataclass

from .plan import JoinPlan, Operand, PlanSide, Projection, Residual
from .shape import AGG_FUNCS, ColRef as _ColRef, HipDeclined, JoinShape, OrderKey, SelItem, TableRef as _TableRef
from .shape import bind_expression as _bind_expression
from .shape import condition_terms as _condition_terms
from .shape import operand_sides as _operand_sides
from .shape import decline as _decline
from .shape import genomic_col as _genomic_col
from .shape import lower_join_shape, resolve_projection
from .shape import norm as _norm
from .shape import table_side as _table_side
from .table import Table, Tables, build_tables, encoding_of


# ------------------------------------------------------------------- tokens
_TOKEN = re.compile(
    r"""\s*(?:
        (?P<str>'(?:[^']|'')*')
      | (?P<qid>"(?:[^"]|"")*")
      | (?P<num>\d+(?:\.\d+)?)
      | (?P<assign>:=)
      | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
      | (?P<punct>[(),.*;=<>+\-/!%|&^~\[\]:])
    )""",
    re.X,
)

_KEYWORDS = {
    "SELECT", "DISTINCT", "FROM", "JOIN", "INNER", "CROSS", "SEMI", "ANTI", "LEFT", "RIGHT",
    "FULL", "OUTER", "NATURAL", "ON", "USING", "WHERE", "AND", "OR", "NOT", "AS", "INTERSECTS",
    "CONTAINS", "WITHIN", "GROUP", "ORDER", "BY", "HAVING", "LIMIT", "OFFSET", "LATERAL",
    "NEAREST", "WITH", "UNION", "ANY", "ALL", "EXISTS", "TRUE", "FALSE", "BETWEEN", "IN", "IS", "NULL", "LIKE",
}


@dataclass
class Tok:
    kind: str   # kw / id / qid / str / num / punct / assign / end
    text: str   # keyword upper-cased; identifiers verbatim (quotes stripped)
    quoted: bool = False


def _tokenize(sql: str) -> list[Tok]:
    out: list[Tok] = []
    pos = 0
    n = len(sql)
    while pos < n:
        m = _TOKEN.match(sql, pos)
        if not m:
            if sql[pos:].strip() == "":
                break
            raise ValueError(f"Parse error: unexpected character {sql[pos:].lstrip()[:1]!r}")
        pos = m.end()
        if m.group("str") is not None:
            out.append(Tok("str", m.group("str")[1:-1].replace("''", "'")))
        elif m.group("qid") is not None:
            out.append(Tok("id", m.group("qid")[1:-1].replace('""', '"'), quoted=True))
        elif m.group("num") is not None:
            out.append(Tok("num", m.group("num")))
        elif m.group("assign") is not None:
            out.append(Tok("assign", ":="))
        elif m.group("id") is not None:
            t = m.group("id")
            out.append(Tok("kw", t.upper()) if t.upper() in _KEYWORDS else Tok("id", t))
        else:
            out.append(Tok("punct", m.group("punct")))
    out.append(Tok("end", ""))
    return out


# ------------------------------------------------------------------- parser
class _Parser:
    def __init__(self, sql: str):
        self.toks = _tokenize(sql)
        self.i = 0

    # -- token helpers
    def peek(self, k: int = 0) -> Tok:
        return self.toks[min(self.i + k, len(self.toks) - 1)]

    def next(self) -> Tok:
        t = self.toks[self.i]
        self.i = min(self.i + 1, len(self.toks) - 1)
        return t

    def at_kw(self, *kws: str) -> bool:
        t = self.peek()
        return t.kind == "kw" and t.text in kws

    def at_punct(self, p: str) -> bool:
        t = self.peek()
        return t.kind == "punct" and t.text == p

    def expect_kw(self, kw: str) -> None:
        if not self.at_kw(kw):
            raise ValueError(f"Parse error: expected {kw} near {self.peek().text!r}")
        self.next()

    def expect_punct(self, p: str) -> None:
        if not self.at_punct(p):
            raise ValueError(f"Parse error: expected {p!r} near {self.peek().text!r}")
        self.next()

    def ident(self) -> Tok:
        t = self.peek()
        if t.kind != "id":
            raise ValueError(f"Parse error: expected an identifier near {t.text!r}")
        return self.next()

    # -- grammar pieces
    def colref(self) -> _ColRef:
        if self.at_punct("*"):
            self.next()
            return _ColRef(None, False, "*", star=True)
        first = self.ident()
        if self.at_punct("."):
            self.next()
            if self.at_punct("*"):
                self.next()
                return _ColRef(first.text, first.quoted, "*", star=True)
            second = self.next()
            if second.kind not in ("id", "kw"):
                raise ValueError(f"Parse error: expected a column name near {second.text!r}")
            # `a.start` / `a.end`: keywords are fine as column names after a dot
            col = second.text if second.kind == "id" else second.text.lower()
            return _ColRef(first.text, first.quoted, col)
        return _ColRef(None, False, first.text)

    def table_ref(self) -> _TableRef:
        name = self.ident()
        if self.at_punct("."):
            raise HipDeclined("catalog/schema-qualified tables are not handled by dialect='hip'")
        if self.at_punct("("):
            raise HipDeclined("table functions as join operands are not handled by dialect='hip'")
        alias, aq = name.text, name.quoted
        if self.at_kw("AS"):
            self.next()
            a = self.ident()
            alias, aq = a.text, a.quoted
        elif self.peek().kind == "id":
            a = self.next()
            alias, aq = a.text, a.quoted
        return _TableRef(name.text, alias, aq)


_CLAUSE_END = ("WHERE", "GROUP", "ORDER", "HAVING", "LIMIT", "OFFSET", "UNION", "JOIN", "INNER", "LEFT",
               "RIGHT", "FULL", "CROSS", "SEMI", "ANTI", "NATURAL")


_ARITH_FUNCS = {"LEAST": "least", "GREATEST": "greatest", "ABS": "abs"}


def _parse_atom(p: _Parser, expr=None):
    """A residual operand without arithmetic: [-]number, 'string', a column reference, or -- when ``expr`` (the
    parser of a full expression) is given -- ``LEAST / GREATEST / ABS ( ... )``."""
    neg = False
    if p.at_punct("-") and p.peek(1).kind == "num":
        p.next()
        neg = True
    t = p.peek()
    if t.kind == "num":
        p.next()
        v = float(t.text) if "." in t.text else int(t.text)
        return ("lit", -v if neg else v)
    if t.kind == "str":
        p.next()
        return ("lit", t.text)
    if t.kind == "kw" and t.text in ("TRUE", "FALSE"):
        raise _decline("boolean literal in a join condition")
    if t.kind != "id":
        raise _decline(f"join condition operand near {t.text!r}")
    if expr is not None and not t.quoted and t.text.upper() in _ARITH_FUNCS and p.peek(1).kind == "punct" \
            and p.peek(1).text == "(":
        name = _ARITH_FUNCS[p.next().text.upper()]
        p.expect_punct("(")
        args = [expr(p)]
        while p.at_punct(","):
            p.next()
            args.append(expr(p))
        p.expect_punct(")")
        if name == "abs" and len(args) != 1:
            raise ValueError("Parse error: ABS takes one argument")
        return ("fn", name, args)
    ref = p.colref()
    if p.at_punct("("):
        raise _decline("function call in a join condition")
    return ("col", ref)


def _parse_operand(p: _Parser, level: int = 0):
    """An arithmetic expression over columns and literals: ``+ -`` < ``* /`` < unary minus < ``( ... )``, LEAST /
    GREATEST / ABS -- what the overlap-fraction recipes put beside the INTERSECTS (docs/recipes/intersect.rst:144-190;
    the reference inlines the text, intersects_duckdb.py:889-912).  ``("lit", v)`` | ``("col", ref)`` | ``("fn", op, args)``."""
    if level < 2:
        ops = "+-" if level == 0 else "*/"
        lhs = _parse_operand(p, level + 1)
        while p.peek().kind == "punct" and p.peek().text in ops:
            op = p.next().text
            lhs = ("fn", op, [lhs, _parse_operand(p, level + 1)])
        return lhs
    if p.at_punct("-") and p.peek(1).kind != "num":
        p.next()
        return ("fn", "neg", [_parse_operand(p, 2)])
    if p.at_punct("("):
        if p.peek(1).kind == "kw" and p.peek(1).text in ("SELECT", "WITH"):
            raise _decline("sub-query in a condition")
        p.next()
        v = _parse_operand(p, 0)
        p.expect_punct(")")
        return v
    return _parse_atom(p, _parse_operand)


def _parse_comparison_op(p: _Parser):
    t = p.peek()
    if t.kind != "punct" or t.text not in "=<>!":
        return None
    p.next()
    op = t.text
    n = p.peek()
    if n.kind == "punct" and ((op == "<" and n.text in "=>") or (op == ">" and n.text == "=")
                              or (op == "!" and n.text == "=") or (op == "=" and n.text == "=")):
        p.next()
        op += n.text
    if op == "!":
        return None
    return {"<>": "!=", "==": "="}.get(op, op)


def _no_arithmetic(p: _Parser) -> None:
    if p.peek().kind == "punct" and p.peek().text in "+-*/%|&^":
        raise _decline("arithmetic in a join condition")


def _parse_predicate(p: _Parser, allow_literal: bool, operand=None):
    """One predicate -> a condition tree node (``shape.condition_terms``): ``<col> INTERSECTS
    <col | 'chr:lo-hi'>``, ``<operand> op <operand>``, ``[NOT] BETWEEN``, ``[NOT] IN (literals)``,
    ``IS [NOT] NULL``.  LIKE, arithmetic, functions, sub-queries decline."""
    if p.at_kw("EXISTS"):
        raise _decline("EXISTS condition")
    operand = operand or _parse_operand     # a CLUSTER predicate also reads PREV(col)
    lhs = operand(p)
    if p.at_kw("INTERSECTS"):
        p.next()
        if lhs[0] != "col":
            raise _decline("INTERSECTS with a literal on the left")
        if p.peek().kind == "str":
            if not allow_literal:
                raise _decline("literal-range INTERSECTS inside a join")
            return ("leaf", ("intersects_lit", lhs[1], p.next().text))
        if p.at_kw("ANY", "ALL"):
            raise _decline("INTERSECTS ANY/ALL")
        if p.peek().kind != "id":
            raise _decline("INTERSECTS operand that is not a column")
        return ("leaf", ("intersects", lhs[1], p.colref()))
    if p.at_kw("CONTAINS", "WITHIN"):
        raise _decline(f"{p.peek().text} predicate")
    _no_arithmetic(p)
    negated = False
    if p.at_kw("NOT") and p.peek(1).kind == "kw" and p.peek(1).text in ("BETWEEN", "IN", "LIKE"):
        p.next()
        negated = True
    if p.at_kw("LIKE"):
        raise _decline("LIKE predicate")
    if p.at_kw("BETWEEN"):
        p.next()
        lo = operand(p)
        _no_arithmetic(p)
        p.expect_kw("AND")
        hi = operand(p)
        _no_arithmetic(p)
        node = ("and", [("leaf", ("cmp", lhs, ">=", lo)), ("leaf", ("cmp", lhs, "<=", hi))])
    elif p.at_kw("IN"):
        p.next()
        p.expect_punct("(")
        if p.at_kw("SELECT", "WITH"):
            raise _decline("IN (sub-query)")
        values = []
        while True:
            v = operand(p)
            if v[0] != "lit":
                raise _decline("IN list with a non-literal member")
            values.append(v)
            if p.at_punct(","):
                p.next()
                continue
            break
        p.expect_punct(")")
        node = ("or", [("leaf", ("cmp", lhs, "=", v)) for v in values])
    elif p.at_kw("IS"):
        p.next()
        is_not = False
        if p.at_kw("NOT"):
            p.next()
            is_not = True
        if not p.at_kw("NULL"):
            raise _decline("IS predicate other than IS [NOT] NULL")
        p.next()
        if lhs[0] == "lit":
            raise _decline("IS NULL over a literal")
        return ("leaf", ("cmp", lhs, "notnull" if is_not else "isnull", ("lit", 0)))
    else:
        op = _parse_comparison_op(p)
        if op is None:
            raise _decline("join condition other than INTERSECTS / simple comparisons")
        rhs = operand(p)
        _no_arithmetic(p)
        return ("leaf", ("cmp", lhs, op, rhs))
    return ("not", node) if negated else node


def _parse_bool(p: _Parser, allow_literal: bool, level: int = 0, operand=None):
    """``OR`` < ``AND`` < ``NOT`` < ``( ... )`` | predicate."""
    if level == 0 or level == 1:
        word, kind = ("OR", "or") if level == 0 else ("AND", "and")
        kids = [_parse_bool(p, allow_literal, level + 1, operand)]
        while p.at_kw(word):
            p.next()
            kids.append(_parse_bool(p, allow_literal, level + 1, operand))
        return kids[0] if len(kids) == 1 else (kind, kids)
    if p.at_kw("NOT"):
        p.next()
        return ("not", _parse_bool(p, allow_literal, 2, operand))
    if p.at_punct("("):
        if p.peek(1).kind == "kw" and p.peek(1).text in ("SELECT", "WITH"):
            raise _decline("sub-query in a condition")
        # a boolean group, or the opening parenthesis of an arithmetic operand ("(a.end - a.start) > 5"): try the
        # group first; what follows its ")" tells
        save = p.i
        p.depth = getattr(p, "depth", 0) + 1
        if p.depth > 12:   # (each level may be parsed twice: bound the work on adversarial nesting)
            raise _decline("condition nested too deeply")
        try:
            p.next()
            node = _parse_bool(p, allow_literal, 0, operand)
            p.expect_punct(")")
            t = p.peek()
            if not ((t.kind == "punct" and t.text in "+-*/=<>!") or (t.kind == "kw" and t.text in ("BETWEEN", "IN", "IS", "LIKE"))
                    or (t.kind == "kw" and t.text == "NOT" and p.peek(1).kind == "kw" and p.peek(1).text in ("BETWEEN", "IN", "LIKE"))):
                return node
        except HipDeclined as exc:
            if "nested too deeply" in str(exc):
                raise
        except ValueError:
            pass
        finally:
            p.depth -= 1
        p.i = save
        try:
            return _parse_predicate(p, allow_literal, operand)
        except HipDeclined:
            raise
        except ValueError:   # e.g. "(a.x > 1) = TRUE": a condition used as a value -- valid SQL, no evaluator here
            raise _decline("parenthesised condition used as a value")
    return _parse_predicate(p, allow_literal, operand)


def _parse_conjunction(p: _Parser, allow_literal: bool = False, trees: bool = False):
    """A boolean condition -> the terms of its conjunctive normal form: ``("intersects", ...)``,
    ``("cmp", lhs, op, rhs)`` and ``("or", [cmp, ...])`` (``shape.condition_terms``).  The
    reference inlines any such extra beside the INTERSECTS as SQL text (``_classify_extras``,
    intersects_duckdb.py:889-912); arithmetic, functions, LIKE and sub-queries have no
    evaluator here and decline, as does a spatial predicate under OR / NOT (where the
    reference falls back too)."""
    return _condition_terms(_parse_bool(p, allow_literal), trees=trees)


def _own_table_residuals(terms, own) -> list:
    """WHERE terms over ONE table (literal-range filter, CLUSTER / MERGE) -> residuals; ``own``
    checks a column's qualifier and returns its name."""
    def bind(o) -> Operand:
        if o[0] == "lit":
            v = o[1]
            return Operand("str" if isinstance(v, str) else ("float" if isinstance(v, float) else "int"), v)
        return Operand("l", own(o[1]))

    out, group = [], 0
    for t in terms:
        leaves, g = ([t], 0)
        if t[0] == "or":
            group += 1
            leaves, g = t[1], group
        for _, lhs, op, rhs in leaves:
            a, b = _bind_expression(lhs, bind), _bind_expression(rhs, bind)
            if not _operand_sides(a) and not _operand_sides(b):
                raise _decline("constant predicate")
            out.append(Residual("where", a, op, b, g))
    return out


_UNSUPPORTED_TAIL = ("GROUP", "ORDER", "HAVING", "LIMIT", "OFFSET", "UNION")


def _parse_aggregate_call(p: _Parser, where: str):
    """``FUNC([DISTINCT] <col> | *)`` at the cursor -> (func, distinct, ColRef | None)."""
    func = p.peek().text.upper()
    if func not in AGG_FUNCS:
        raise _decline(f"function call in {where}")
    p.next()
    p.next()
    distinct = False
    if p.at_kw("DISTINCT"):
        p.next()
        distinct = True
    if p.at_punct("*"):
        p.next()
        ref = None
        if func != "COUNT" or distinct:
            raise _decline(f"{func}(*)")
    else:
        if p.peek().kind != "id":
            raise _decline("aggregate over an expression")
        ref = p.colref()
        if ref.star:
            raise _decline("star inside an aggregate")   # COUNT(a.*), #204
        if p.peek().kind == "punct" and p.peek().text in "+-/*=<>(":
            raise _decline("aggregate over an expression")
    p.expect_punct(")")
    if p.peek().kind == "id" and not p.peek().quoted and p.peek().text.upper() in ("OVER", "FILTER"):
        raise _decline("window aggregate / FILTER clause")
    return func, distinct, ref


def _parse_having(p: _Parser):
    """A boolean condition over comparisons ``<operand> op <operand>`` (also BETWEEN / IN / IS NULL), an operand
    being a plain aggregate, a column (qualified key or output name) or a literal -> the terms of its conjunctive
    normal form (the reference hands HAVING to the engine verbatim, intersects_duckdb.py:1336-1400; sub-queries
    and arithmetic decline here)."""

    def operand(p: _Parser):
        t = p.peek()
        if p.at_kw("EXISTS", "SELECT"):
            raise _decline("sub-query HAVING condition")
        if t.kind == "id" and not t.quoted and p.peek(1).kind == "punct" and p.peek(1).text == "(":
            func, distinct, ref = _parse_aggregate_call(p, "HAVING")
            return ("agg", SelItem(ref, None, func, distinct))
        return _parse_atom(p)

    terms = _condition_terms(_parse_bool(p, False, 0, operand))
    if any(t[0] not in ("cmp", "or") for t in terms):
        raise _decline("spatial predicate in HAVING")
    return terms


def _parse_projection(p: _Parser) -> list[SelItem]:
    """SELECT list: qualified columns and plain aggregates ``FUNC([DISTINCT] <col>)`` / ``COUNT(*)``
    (the projections ``_resolve_projections`` can rebuild, intersects_duckdb.py:1402-1644); every
    other expression declines (#204, #205)."""
    items: list[SelItem] = []
    while True:
        if p.peek().kind == "kw" and not p.at_kw("FROM"):
            raise _decline(f"projection starting with {p.peek().text}")
        if p.peek().kind in ("num", "str") or p.at_punct("("):
            raise _decline("expression in the SELECT list")
        t = p.peek()
        is_call = t.kind == "id" and not t.quoted and p.peek(1).kind == "punct" and p.peek(1).text == "("
        func = None
        distinct = False
        if is_call:
            func, distinct, ref = _parse_aggregate_call(p, "the SELECT list")
            if ref is not None and func == "COUNT" and not distinct:
                ref.count = True
        else:
            ref = p.colref()
            if p.at_punct("("):
                raise _decline("function call / aggregate in the SELECT list")
        if p.peek().kind == "punct" and p.peek().text in "+-/*=<>":
            raise _decline("expression in the SELECT list")
        alias = None
        if p.at_kw("AS"):
            p.next()
            alias = p.next().text
        elif p.peek().kind == "id":
            alias = p.next().text
        items.append(SelItem(ref, alias, func, distinct))
        if p.at_punct(","):
            p.next()
            continue
        break
    return items


def _parse_nearest(p: _Parser, tables: Tables, from_ref: _TableRef):
    """``CROSS JOIN LATERAL NEAREST(target, reference := a.interval, k := 1, ...) b``
    (src/giql/expanders/nearest.py:240-252, 336-397)."""
    p.expect_kw("NEAREST")
    p.expect_punct("(")
    target = p.ident()
    args: dict[str, object] = {}
    while p.at_punct(","):
        p.next()
        key = p.next()
        if p.peek().kind != "assign":
            raise ValueError("Parse error: NEAREST arguments must be named (name := value)")
        p.next()
        kname = key.text.lower()
        if kname == "reference":
            args["reference"] = p.colref()
        else:
            v = p.next()
            neg = False
            if v.kind == "punct" and v.text == "-":
                neg = True
                v = p.next()
            if v.kind == "num":
                args[kname] = -int(v.text) if neg else int(v.text)
            elif v.kind == "kw" and v.text in ("TRUE", "FALSE"):
                args[kname] = v.text == "TRUE"
            else:
                raise ValueError(f"Parse error: bad value for NEAREST argument {key.text!r}")
    p.expect_punct(")")
    alias = target.text
    aq = target.quoted
    if p.at_kw("AS"):
        p.next()
        a = p.ident()
        alias, aq = a.text, a.quoted
    elif p.peek().kind == "id":
        a = p.next()
        alias, aq = a.text, a.quoted
    ref = args.get("reference")
    if not isinstance(ref, _ColRef) or ref.table is None:
        raise _decline("NEAREST without a correlated column reference")
    if _norm(ref.table, ref.table_quoted) != _norm(from_ref.alias, from_ref.alias_quoted):
        raise ValueError(f"NEAREST reference {ref.table}.{ref.column} does not name the FROM table")
    if ref.column != _genomic_col(from_ref.name, tables):
        raise ValueError(f"{ref.table}.{ref.column} is not the genomic column of {from_ref.name}")
    k = int(args.get("k", 1))
    if k < 1:
        raise ValueError("NEAREST k must be a positive integer")
    if k > 1 << 20:
        raise _decline("NEAREST with k > 2^20")   # giql_hip_nearest_k_dev's cap (n_a * k must stay below 2^31)
    md = args.get("max_distance")
    return (_TableRef(target.text, alias, aq), (None if md is None else int(md)), bool(args.get("signed", False)), k,
            bool(args.get("stranded", False)))


def _literal_range_sql(p: _Parser, proj_text: str, from_ref: _TableRef, tables: Tables) -> str:
    """BASELINE config 1: the generic literal-range predicate
    (src/giql/expanders/intersects.py:85-107, 204-222)."""
    col = p.colref()
    p.expect_kw("INTERSECTS")
    lit = p.next()
    if lit.kind != "str":
        raise _decline("non-literal right operand")
    if p.peek().kind != "end" and not p.at_punct(";"):
        raise _decline("extra clauses after the literal predicate")
    m = re.match(r"^(?P<chr>[\w.]+):(?P<start>\d+)-(?P<end>\d+)$", lit.text.strip())
    if not m:
        raise _decline("literal range formats other than 'chr:start-end'")
    start, end = int(m.group("start")), int(m.group("end"))
    if start >= end:
        raise ValueError(f"Start must be less than end: {start} >= {end}")
    t = tables.get(from_ref.name) or Table(from_ref.name)
    if col.column != t.genomic_col:
        raise ValueError(f"{col.column!r} is not the genomic column of {from_ref.name}")
    if encoding_of(t) != ("0based", "half_open"):
        raise _decline("literal predicate over a non-canonical table")
    q = (col.table + ".") if col.table else ""
    chrom = m.group("chr").replace("'", "''")
    return (f"SELECT {proj_text} FROM {from_ref.name}"
            + (f" AS {from_ref.alias}" if from_ref.alias != from_ref.name else "")
            + f" WHERE ({q}\"{t.chrom_col}\" = '{chrom}' AND {q}\"{t.start_col}\" < {end} "
            f"AND {q}\"{t.end_col}\" > {start})")


def _is_single_table_filter(p: _Parser) -> bool:
    """``... FROM <one table> WHERE ...`` with a literal-range INTERSECTS and no join."""
    depth = 0
    seen_from = False
    k = 0
    toks = p.toks
    while k < len(toks):
        t = toks[k]
        if t.kind == "punct" and t.text in "()":
            depth += 1 if t.text == "(" else -1
        if depth == 0 and t.kind == "kw":
            if t.text == "FROM":
                seen_from = True
            elif seen_from and t.text in ("JOIN", "LATERAL"):
                return False
            elif seen_from and t.text == "WHERE":
                return any(a.kind == "kw" and a.text == "INTERSECTS" and b.kind == "str"
                           for a, b in zip(toks[k:], toks[k + 1:]))
        if depth == 0 and seen_from and t.kind == "punct" and t.text == ",":
            return False
        k += 1
    return False


def _lower_filter(p: _Parser, tbls: Tables) -> JoinPlan:
    """``SELECT <cols | *> FROM t WHERE interval INTERSECTS 'chr:lo-hi' [AND comparisons]``
    (BASELINE config 1; the literal-range predicate of src/giql/expanders/intersects.py:85-107,
    204-222: ``chrom = 'chr' AND start < hi AND end > lo`` on a canonical table).  Lowered to
    three residual comparisons for the select kernel."""
    p.expect_kw("SELECT")
    if p.at_kw("DISTINCT"):
        raise _decline("DISTINCT over a literal-range filter")
    items = []
    while True:
        if p.peek().kind in ("num", "str") or p.at_punct("("):
            raise _decline("expression in the SELECT list")
        ref = p.colref()
        if p.at_punct("("):
            raise _decline("function call in the SELECT list")
        if p.peek().kind == "punct" and p.peek().text in "+-/*=<>":
            raise _decline("expression in the SELECT list")
        alias = None
        if p.at_kw("AS"):
            p.next()
            alias = p.next().text
        elif p.peek().kind == "id":
            alias = p.next().text
        items.append((ref, alias))
        if p.at_punct(","):
            p.next()
            continue
        break
    p.expect_kw("FROM")
    ref_t = p.table_ref()
    p.expect_kw("WHERE")
    terms = _parse_conjunction(p, allow_literal=True)
    if p.peek().kind == "kw" or p.at_punct(","):
        raise _decline(f"{p.peek().text} clause after the literal predicate")
    if p.peek().kind != "end" and not p.at_punct(";"):
        raise _decline(f"trailing input near {p.peek().text!r}")
    side = _table_side(ref_t, tbls)
    table = tbls.get(ref_t.name) or Table(ref_t.name)
    lits = [t for t in terms if t[0] == "intersects_lit"]
    if len(lits) != 1 or any(t[0] == "intersects" for t in terms):
        raise _decline("more than one spatial predicate")

    def own(refc: _ColRef) -> str:
        if refc.table is not None and _norm(refc.table, refc.table_quoted) != side.alias:
            raise ValueError(f"Unknown table qualifier {refc.table!r}; expected {side.alias!r}")
        return refc.column

    _, col, text = lits[0]
    if own(col) != table.genomic_col:
        raise ValueError(f"{col.column!r} is not the genomic column of {ref_t.name}")
    m = re.match(r"^(?P<chr>[\w.]+):(?P<start>\d+)-(?P<end>\d+)$", text.strip())
    if not m:
        raise _decline("literal range formats other than 'chr:start-end'")
    lo, hi = int(m.group("start")), int(m.group("end"))
    if lo >= hi:
        raise ValueError(f"Start must be less than end: {lo} >= {hi}")
    if encoding_of(table) != ("0based", "half_open"):
        raise _decline("literal predicate over a non-canonical table")
    residuals = [Residual("where", Operand("l", side.chrom_col), "=", Operand("str", m.group("chr"))),
                 Residual("where", Operand("l", side.start_col), "<", Operand("int", hi)),
                 Residual("where", Operand("l", side.end_col), ">", Operand("int", lo))]
    residuals += _own_table_residuals([t for t in terms if t[0] in ("cmp", "or")], own)
    proj = []
    for refc, alias in items:
        if refc.star:
            if refc.table:
                own(refc)
            proj.append(Projection("star", "*", "*"))
        else:
            proj.append(Projection("l", own(refc), alias or refc.column))
    return JoinPlan("FILTER", side, None, tuple(proj), residuals=tuple(residuals))


def _has_cluster_or_merge(p: _Parser) -> bool:
    depth = 0
    for k, t in enumerate(p.toks):
        if t.kind == "punct" and t.text in "()":
            depth += 1 if t.text == "(" else -1
        if t.kind == "kw" and t.text == "FROM" and depth == 0:
            return False
        if (t.kind == "id" and not t.quoted and t.text.upper() in ("CLUSTER", "MERGE")
                and p.toks[k + 1].kind == "punct" and p.toks[k + 1].text == "("):
            return True
    return False


def _parse_prev_operand(p: _Parser):
    """An operand of a CLUSTER predicate: ``PREV(<column>)`` (the sorted predecessor's value), a column of the
    current row, or a literal.  ``PREV`` takes exactly one column and does not nest (cluster.py:623-633)."""
    t = p.peek()
    if t.kind == "id" and not t.quoted and t.text.upper() == "PREV" and p.peek(1).kind == "punct" and p.peek(1).text == "(":
        p.next()
        p.expect_punct("(")
        if p.at_punct(")"):
            raise ValueError("PREV() takes exactly one column argument; got 0.")
        inner = p.peek()
        if inner.kind == "id" and not inner.quoted and inner.text.upper() == "PREV" and p.peek(1).kind == "punct" \
                and p.peek(1).text == "(":
            raise ValueError("PREV() cannot be nested; a CLUSTER/MERGE predicate compares only the immediate predecessor.")
        if inner.kind != "id":
            raise _decline("PREV() of an expression")
        ref = p.colref()
        if p.at_punct(","):
            raise ValueError("PREV() takes exactly one column argument; got 2.")
        if not p.at_punct(")"):
            raise _decline("PREV() of an expression")
        p.expect_punct(")")
        return ("prev", ref)
    return _parse_atom(p)


def _parse_cluster_predicate(p: _Parser):
    """The ``predicate :=`` argument up to its end -> the terms of its conjunctive normal form, ``("cmp", lhs, op,
    rhs)`` / ``("or", [cmp, ...])`` over columns, ``PREV(col)`` and literals (the reference inlines the text into its
    adjacency CASE, cluster.py:281-296; NULL -> not adjacent either way).  Arithmetic / functions decline."""
    terms = _condition_terms(_parse_bool(p, False, 0, _parse_prev_operand))
    if any(t[0] not in ("cmp", "or") for t in terms):
        raise _decline("spatial predicate inside a CLUSTER predicate")
    return terms


def _parse_cluster_call(p: _Parser):
    """``CLUSTER(`` / ``MERGE(`` argument list -> ``(genomic colref, distance, stranded, predicate terms)``;
    named arguments accept ``:=``, ``=`` and ``=>`` (tests/test_cluster_parsing.py:22-75)."""
    p.expect_punct("(")
    this = None
    distance = 0
    stranded = False
    predicate = []
    n_pos = 0
    while not p.at_punct(")"):
        named = None
        if p.peek().kind == "id" and (p.peek(1).kind == "assign" or (p.peek(1).kind == "punct" and p.peek(1).text == "=")):
            named = p.next().text.lower()
            p.next()
            if p.at_punct(">"):
                p.next()
        if named is None:
            if n_pos == 0:
                if p.peek().kind != "id":
                    raise ValueError("CLUSTER requires a genomic interval column as its first argument.")
                this = p.colref()
            elif n_pos == 1:
                if p.peek().kind != "num" or "." in p.peek().text:
                    raise _decline("non-literal CLUSTER / MERGE distance")
                distance = int(p.next().text)
            else:
                raise _decline("more than two positional CLUSTER / MERGE arguments")
            n_pos += 1
        elif named == "stranded":
            if not p.at_kw("TRUE", "FALSE"):
                raise _decline("non-literal stranded argument")
            stranded = p.next().text == "TRUE"
        elif named == "distance":
            if p.peek().kind != "num" or "." in p.peek().text:
                raise _decline("non-literal CLUSTER / MERGE distance")
            distance = int(p.next().text)
        elif named == "predicate":
            predicate = _parse_cluster_predicate(p)
        else:
            raise _decline(f"CLUSTER / MERGE argument {named!r}")
        if p.at_punct(","):
            p.next()
    p.expect_punct(")")
    if this is None:
        raise ValueError("CLUSTER requires a genomic interval column as its first argument.")
    return this, distance, stranded, predicate


def _lower_cluster(p: _Parser, tbls: Tables) -> JoinPlan:
    """``SELECT <cols | *>, CLUSTER(interval[, d][, stranded := b]) AS id FROM t [WHERE ...]`` and
    ``SELECT MERGE(interval[, d][, stranded := b]) [, COUNT(*) AS n] FROM t [WHERE ...]``
    (src/giql/expanders/cluster.py:81-205, src/giql/expanders/merge.py:62-183).  A WHERE
    of simple comparisons filters the rows before clustering, as in the reference, where
    it lands inside the inner ``__giql_lag_calc`` subquery (cluster.py:404-420)."""
    p.expect_kw("SELECT")
    if p.at_kw("DISTINCT"):
        raise _decline("DISTINCT with CLUSTER / MERGE")
    items = []   # (kind, payload, alias)
    while True:
        t = p.peek()
        if t.kind == "id" and not t.quoted and t.text.upper() in ("CLUSTER", "MERGE") and p.peek(1).kind == "punct" \
                and p.peek(1).text == "(":
            op = p.next().text.upper()
            items.append((op, _parse_cluster_call(p)))
        elif t.kind == "id" and not t.quoted and t.text.upper() == "COUNT" and p.peek(1).kind == "punct" \
                and p.peek(1).text == "(":
            p.next()
            p.next()
            if not p.at_punct("*"):
                raise _decline("aggregate other than COUNT(*) beside MERGE")
            p.next()
            p.expect_punct(")")
            items.append(("COUNT", None))
        elif t.kind in ("num", "str") or p.at_punct("("):
            raise _decline("expression in the SELECT list")
        else:
            ref = p.colref()
            if p.at_punct("("):
                raise _decline("function call in the SELECT list")
            items.append(("COL", ref))
        if p.peek().kind == "punct" and p.peek().text in "+-/*=<>":
            raise _decline("expression in the SELECT list")
        alias = None
        if p.at_kw("AS"):
            p.next()
            alias = p.next().text
        elif p.peek().kind == "id":
            alias = p.next().text
        items[-1] = items[-1] + (alias,)
        if p.at_punct(","):
            p.next()
            continue
        break
    p.expect_kw("FROM")
    if p.at_punct("("):
        raise _decline("CLUSTER / MERGE over a sub-query")
    ref = p.table_ref()
    where_terms = []
    if p.at_kw("WHERE"):
        p.next()
        where_terms = _parse_conjunction(p)
        if any(t[0] not in ("cmp", "or") for t in where_terms):
            raise _decline("spatial predicate beside CLUSTER / MERGE")
    if p.peek().kind == "kw" or p.at_punct(","):
        raise _decline(f"{p.peek().text} clause with CLUSTER / MERGE")
    if p.peek().kind != "end" and not p.at_punct(";"):
        raise _decline(f"trailing input near {p.peek().text!r}")

    side = _table_side(ref, tbls)
    table = tbls.get(ref.name) or Table(ref.name)
    ops = [it for it in items if it[0] in ("CLUSTER", "MERGE")]
    if {it[0] for it in ops} == {"CLUSTER", "MERGE"}:
        raise ValueError("CLUSTER and MERGE cannot be combined in one SELECT")  # reject_cluster_merge_mix
    if len(ops) > 1:
        raise ValueError(f"Multiple {ops[0][0]} expressions not yet supported")  # cluster.py:176-181, merge.py:173-175
    op, (this, distance, stranded, predicate), op_alias = ops[0]
    if this.star or (this.table is not None and _norm(this.table, this.table_quoted) != side.alias) \
            or this.column != table.genomic_col:
        raise ValueError(f"{op} operand must be the table's genomic column ({table.genomic_col!r})")
    if stranded and not table.strand_col:
        raise ValueError(f"{op}(stranded := true) needs a strand column on table {ref.name!r}")

    def own(refc: _ColRef) -> str:
        if refc.table is not None and _norm(refc.table, refc.table_quoted) != side.alias:
            raise ValueError(f"Unknown table qualifier {refc.table!r}; expected {side.alias!r}")
        return refc.column

    proj = []
    if op == "CLUSTER":
        if not op_alias:
            raise _decline("CLUSTER without an alias")
        if sum(1 for it in items if it[0] == "COL" and it[1].star) > 1:
            raise ValueError("CLUSTER does not support multiple star projections "
                             "(e.g. SELECT *, *, CLUSTER(...)); project a single star")  # cluster.py:182-196
        for kind, payload, alias in items:
            if kind == "CLUSTER":
                proj.append(Projection("cluster", "", op_alias))
            elif kind == "COUNT":
                raise _decline("aggregate beside CLUSTER")
            elif payload.star:
                own(payload) if payload.table else None
                proj.append(Projection("star", "*", "*"))
            else:
                proj.append(Projection("l", own(payload), alias or payload.column))
    else:
        for kind, payload, alias in items:
            if kind == "MERGE":
                continue
            if kind == "COUNT":
                if not alias:
                    raise _decline("COUNT(*) without an alias beside MERGE")
                proj.append(Projection("count", "*", alias))
            elif payload.star:
                raise ValueError("MERGE cannot be combined with a star projection (e.g. SELECT *, MERGE(...))")  # merge.py:96-133
            else:
                # grouping keys (chrom, strand when stranded) are already projected by MERGE (merge.py:297-304)
                keys = {side.chrom_col.lower()} | ({table.strand_col.lower()} if stranded else set())
                if own(payload).lower() not in keys or (alias and alias.lower() != payload.column.lower()):
                    raise ValueError(f"MERGE cannot project the non-aggregated column {payload.column!r}")
        names = [pp.name.lower() for pp in proj]
        if len(set(names)) != len(names) or set(names) & {side.chrom_col.lower(), side.start_col.lower(),
                                                          side.end_col.lower()}:
            raise ValueError("MERGE cannot project a column that collides with chrom/start/end")  # merge.py:317-323
    residuals = _own_table_residuals(where_terms, own)
    def pbind(o) -> Operand:
        if o[0] == "lit":
            v = o[1]
            return Operand("str" if isinstance(v, str) else ("float" if isinstance(v, float) else "int"), v)
        return Operand("r" if o[0] == "prev" else "l", own(o[1]))

    cluster_pred, pgroup = [], 0
    for t in predicate:
        leaves, g = ([t], 0)
        if t[0] == "or":
            pgroup += 1
            leaves, g = t[1], pgroup
        for _, lhs, cmp_op, rhs in leaves:
            cluster_pred.append(Residual("predicate", pbind(lhs), cmp_op, pbind(rhs), g))
    return JoinPlan(op, side, None, tuple(proj), residuals=tuple(residuals), distance=distance,
                    stranded=stranded, strand_col=table.strand_col if stranded else None,
                    cluster_predicate=tuple(cluster_pred))


def build_plan(giql: str, tables=None) -> JoinPlan:
    """Parse *giql* and lower the INTERSECTS / NEAREST join (or a CLUSTER / MERGE
    query) to a :class:`JoinPlan`."""
    probe = _Parser(giql)
    if probe.at_kw("SELECT") and _has_cluster_or_merge(probe):
        return _lower_cluster(probe, tables if isinstance(tables, Tables) else build_tables(tables))
    if probe.at_kw("SELECT") and _is_single_table_filter(probe):
        return _lower_filter(probe, tables if isinstance(tables, Tables) else build_tables(tables))
    plan = _lower(giql, tables, want_sql=False)
    assert isinstance(plan, JoinPlan)
    return plan


def _lower(giql: str, tables, want_sql: bool):
    tbls = tables if isinstance(tables, Tables) else build_tables(tables)
    p = _Parser(giql)
    if p.at_kw("WITH"):
        raise _decline("top-level WITH")
    p.expect_kw("SELECT")
    distinct = False
    if p.at_kw("DISTINCT"):
        p.next()
        if p.at_kw("ON"):
            raise _decline("DISTINCT ON")
        distinct = True
    proj_start = p.i
    if want_sql:
        # literal form: keep the projection text verbatim
        depth = 0
        while not (p.at_kw("FROM") and depth == 0) and p.peek().kind != "end":
            depth += p.at_punct("(") - p.at_punct(")")
            p.next()
        proj_text = _render(p.toks[proj_start:p.i])
        items = None
    else:
        items = _parse_projection(p)
        proj_text = ""
    p.expect_kw("FROM")
    from_ref = p.table_ref()

    if want_sql:
        p.expect_kw("WHERE")
        return _literal_range_sql(p, ("DISTINCT " if distinct else "") + proj_text, from_ref, tbls)

    # ---- the join
    kind = "INNER"
    nearest = None
    join_ref = None
    on_seen = False
    using: list[str] = []
    if p.at_punct(","):
        p.next()
        join_ref = p.table_ref()
    else:
        if p.at_kw("NATURAL"):
            raise _decline("NATURAL join")
        if p.at_kw("LEFT", "RIGHT", "FULL"):
            side = p.next().text
            if p.at_kw("SEMI", "ANTI"):
                kind = p.next().text
            elif side == "LEFT":
                # count_overlaps (LEFT [OUTER] JOIN ... COUNT(b.col) ... GROUP BY left keys) is decided
                # by the gate (_match_count_overlaps, intersects_duckdb.py:432-548)
                if p.at_kw("OUTER"):
                    p.next()
                kind = "LEFT"
            else:
                raise _decline(f"{side} outer join")  # intersects_duckdb.py:661-662
        elif p.at_kw("INNER", "CROSS", "SEMI", "ANTI"):
            k = p.next().text
            kind = "INNER" if k in ("INNER", "CROSS") else k
        if not p.at_kw("JOIN"):
            raise _decline("no join (a single-table predicate)")
        p.next()
        if p.at_kw("LATERAL"):
            p.next()
            join_ref, max_distance, signed, k_near, stranded = _parse_nearest(p, tbls, from_ref)
            nearest = (max_distance, signed, k_near, stranded)
            kind = "NEAREST"
        else:
            join_ref = p.table_ref()
            if p.at_kw("USING"):
                p.next()
                p.expect_punct("(")
                while True:
                    using.append(p.ident().text)
                    if p.at_punct(","):
                        p.next()
                        continue
                    break
                p.expect_punct(")")
            if p.at_kw("ON"):
                p.next()
                on_seen = True

    if kind == "NEAREST":
        left = _table_side(from_ref, tbls)
        right = _table_side(join_ref, tbls)
        if p.peek().kind != "end" and not p.at_punct(";"):
            raise _decline("extra clauses after NEAREST")
        if left.alias == right.alias:
            raise _decline("same alias on both sides")
        proj = resolve_projection(items, left, right, False, distance_alias=right.alias)
        strand_col = None
        if nearest[3]:
            # stranded := true matches targets on the reference row's strand (nearest.py:313-333); both
            # tables need a strand column, the TARGET's name is what the reference reads (output_table.strand_col)
            lt_, rt_ = tbls.get(from_ref.name), tbls.get(join_ref.name)
            ls = lt_.strand_col if lt_ is not None else "strand"
            rs = rt_.strand_col if rt_ is not None else "strand"
            if not ls or not rs:
                raise _decline("stranded NEAREST over a table without a strand column")
            strand_col = f"{ls},{rs}"
        return JoinPlan("NEAREST", left, right, proj, distinct, nearest[2], nearest[0], nearest[1],
                        stranded=nearest[3], strand_col=strand_col)

    shape = JoinShape(items=items, from_ref=from_ref, join_ref=join_ref, kind=kind, on_seen=on_seen, using=using,
                      distinct=distinct)
    if on_seen:
        if p.peek().kind not in ("id", "num", "str") and not p.at_punct("-") and not p.at_punct("(") \
                and not p.at_kw("NOT"):
            raise _decline("join condition other than INTERSECTS / simple comparisons")
        shape.on_terms = _parse_conjunction(p, trees=True)   # (a join's residuals may be boolean programs)
    if p.at_kw("WHERE"):
        p.next()
        shape.where_terms = _parse_conjunction(p, trees=True)
    # the clauses the reference lets ride on its outer SELECT wrapper (intersects_duckdb.py:1336-1400)
    if p.at_kw("GROUP"):
        p.next()
        p.expect_kw("BY")
        while True:
            if p.peek().kind != "id":
                raise _decline("GROUP BY expression")
            shape.group_by.append(p.colref())
            if p.at_punct("(") or (p.peek().kind == "punct" and p.peek().text in "+-/*"):
                raise _decline("GROUP BY expression")
            if p.at_punct(","):
                p.next()
                continue
            break
    if p.at_kw("HAVING"):
        p.next()
        shape.having = _parse_having(p)
    if p.at_kw("ORDER"):
        p.next()
        p.expect_kw("BY")
        while True:
            if p.peek().kind != "id":
                raise _decline("ORDER BY expression")
            ref = p.colref()
            if p.at_punct("(") or (p.peek().kind == "punct" and p.peek().text in "+-/*"):
                raise _decline("ORDER BY expression")
            desc = False
            if p.peek().kind == "id" and not p.peek().quoted and p.peek().text.upper() in ("ASC", "DESC"):
                desc = p.next().text.upper() == "DESC"
            nulls_first = None
            if p.peek().kind == "id" and not p.peek().quoted and p.peek().text.upper() == "NULLS":
                p.next()
                t = p.next()
                if t.kind not in ("id", "kw") or t.text.upper() not in ("FIRST", "LAST"):
                    raise ValueError("ORDER BY ... NULLS must be followed by FIRST or LAST")
                nulls_first = t.text.upper() == "FIRST"
            shape.order_by.append(OrderKey(ref, desc, nulls_first))
            if p.at_punct(","):
                p.next()
                continue
            break
    for clause in ("LIMIT", "OFFSET"):
        if p.at_kw(clause):
            p.next()
            t = p.next()
            if t.kind != "num" or "." in t.text:
                raise _decline(f"{clause} that is not an integer literal")
            setattr(shape, clause.lower(), int(t.text))
    if p.at_kw("LIMIT") and shape.limit is None:   # OFFSET n LIMIT m
        p.next()
        t = p.next()
        if t.kind != "num" or "." in t.text:
            raise _decline("LIMIT that is not an integer literal")
        shape.limit = int(t.text)
    if p.peek().kind == "kw" and p.peek().text in _CLAUSE_END:
        raise _decline(f"{p.peek().text} clause")
    if p.at_punct(","):
        raise _decline("a third table")
    if p.peek().kind != "end" and not p.at_punct(";"):
        raise _decline(f"trailing input near {p.peek().text!r}")
    return lower_join_shape(shape, tbls)


def _render(toks: list[Tok]) -> str:
    out = ""
    for t in toks:
        text = f"'{t.text}'" if t.kind == "str" else (f'"{t.text}"' if t.quoted else t.text)
        if out and not (text in (",", ".", ")") or out.endswith((".", "("))):
            out += " "
        out += text
    return out


def transpile(giql: str, tables=None, *, dialect: str | None = None) -> str:
    """Mirror of ``giql.transpile`` for this backend's path.

    ``dialect="hip"``: returns the plan string of the INTERSECTS / NEAREST join
    (hand it to :func:`giql_amd.execute`).  ``dialect=None``: returns SQL for the
    literal-range predicate (plumbing, BASELINE config 1).  Other dialects belong to
    the reference package.
    """
    if dialect == "hip":
        return build_plan(giql, tables).to_string()
    if dialect is None:
        return _lower(giql, tables, want_sql=True)
    raise ValueError(
        f"Unknown dialect: {dialect!r}. giql_amd serves 'hip' (and None for the literal-range "
        "predicate); 'duckdb' / 'datafusion' are emitted by the giql package")
