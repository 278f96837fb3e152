"""Hand-built stand-ins for what giql hands an operator expander: a sqlglot-shaped AST (nodes with
``key`` / ``args`` / ``parent`` / ``meta``), an ``ExpansionContext`` (``tables``, ``resolution``,
``add_statement_finalizer``) and the resolver's ``OperatorResolution`` / ``ResolvedColumn`` -- plain
classes, the way tests/expanders/test_intersects.py:300-336 of the reference hand-builds a context.
Field and arg names follow sqlglot (``Select.args``: expressions / from_ / joins / where / group /
having / order / limit / offset / distinct / with_; ``Join.args``: this / on / using / kind / side /
method; ``Column.args``: this / table; ``Table.args``: this / alias) and the reference's resolver
(src/giql/resolver.py:256-299, 334-389)."""

from dataclasses import dataclass, field


class N:
    def __init__(self, key, **args):
        self.key = key
        self.args = args
        self.parent = None
        self.meta = {}
        for v in args.values():
            for c in (v if isinstance(v, (list, tuple)) else [v]):
                if isinstance(c, N):
                    c.parent = self


def ident(name, quoted=False):
    return N("identifier", this=name, quoted=quoted)


def col(table, name):
    return N("column", this=ident(name), table=ident(table) if table else None)


def star(table=None):
    return N("column", this=N("star"), table=ident(table)) if table else N("star")


def tbl(name, alias=None):
    return N("table", this=ident(name), alias=N("tablealias", this=ident(alias)) if alias else None)


def lit(v):
    return N("literal", this=str(v), is_string=isinstance(v, str))


def alias(node, name):
    return N("alias", this=node, alias=ident(name))


def intersects(l, r):
    return N("intersects", this=l, expression=r)


def cmp(key, l, r):
    return N(key, this=l, expression=r)


def conj(*terms):
    out = terms[0]
    for t in terms[1:]:
        out = N("and", this=out, expression=t)
    return out


def agg(func, arg=None, distinct=False):
    a = N("star") if arg is None else arg
    if distinct:
        a = N("distinct", expressions=[a])
    return N(func.lower(), this=a)


def join(table, on=None, kind=None, side=None, using=None, method=None):
    return N("join", this=table, on=on, kind=kind, side=side, method=method,
             using=[ident(u) for u in using] if using else None)


def select(items, frm, joins, where=None, group=None, having=None, order=None, limit=None, offset=None,
           distinct=False, with_=None):
    return N("select", expressions=list(items), from_=N("from", this=frm), joins=list(joins),
             where=N("where", this=where) if where is not None else None,
             group=N("group", expressions=list(group)) if group else None,
             having=N("having", this=having) if having is not None else None,
             # (key, desc) or (key, desc, nulls_first): sqlglot's parser always fills nulls_first (dialect default)
             order=N("order", expressions=[N("ordered", this=o[0], desc=o[1], nulls_first=o[2] if len(o) > 2 else None)
                                           for o in order]) if order else None,
             limit=N("limit", expression=lit(limit)) if limit is not None else None,
             offset=N("offset", expression=lit(offset)) if offset is not None else None,
             distinct=N("distinct") if distinct is True else distinct or None, with_=with_)


@dataclass(frozen=True)
class ResolvedColumn:           # src/giql/resolver.py:256-299
    chrom: str
    start: str
    end: str
    strand: str | None = None
    table: object = None


@dataclass
class OperatorResolution:       # src/giql/resolver.py:334-389
    operator: str = "Intersects"
    slots: dict = field(default_factory=dict)
    deferrals: dict = field(default_factory=dict)
    columns: dict = field(default_factory=dict)

    def column(self, arg):
        return self.columns.get(arg)


@dataclass
class ExpansionContext:         # src/giql/expander.py:120-186
    tables: object
    resolution: object = None
    finalizers: list = field(default_factory=list)

    def add_statement_finalizer(self, fn):
        self.finalizers.append(fn)


def resolved(alias_, table, chrom="chrom", start="start", end="end"):
    """What pass 1 + pass 2 leave for one operand: alias-qualified fragments; a non-canonical table
    arrives WRAPPED with ``table`` blanked (src/giql/canonicalizer.py:320-378)."""
    s, e = f'{alias_}."{start}"', f'{alias_}."{end}"'
    if table is None or (table.coordinate_system, table.interval_type) == ("0based", "half_open"):
        return ResolvedColumn(f'{alias_}."{chrom}"', s, e, None, table)
    if table.coordinate_system == "1based":
        s = f"({s} - 1)"
    key = (table.coordinate_system, table.interval_type)
    if key == ("0based", "closed"):
        e = f"({e} + 1)"
    elif key == ("1based", "half_open"):
        e = f"({e} - 1)"
    return ResolvedColumn(f'{alias_}."{chrom}"', s, e, None, None)
