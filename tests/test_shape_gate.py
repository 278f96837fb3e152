"""The dialect='hip' shape gate against the reference's accept / decline / raise matrix
(tests/golden/shape_gate.json: the reference's own test cases transcribed as data) -- CPU only.

Both front ends feed ONE gate (giql_amd.shape.lower_join_shape): the sqlglot-free mirror here, giql's
plugin hook in test_plugin_doubles.py.
"""

import pytest

import _golden as G
from giql_amd.plan import JoinPlan
from giql_amd.transpile import HipDeclined, build_plan

GATE = G.load("shape_gate.json")


@pytest.mark.parametrize("case", GATE["accept"], ids=lambda c: c["q"][:70])
def test_gate_accepts_what_the_reference_engages(case):
    plan = build_plan(case["q"], GATE["tables"])
    assert plan.kind == case["kind"]
    assert JoinPlan.from_string(plan.to_string()) == plan   # every accepted plan survives its string form
    for key in ("limit", "offset", "distinct"):
        if key in case:
            assert getattr(plan, key) == case[key], key
    if "order_by" in case:   # [name, descending] or [name, descending, NULLs first]
        assert [list(o)[:len(w)] for o, w in zip(plan.order_by, case["order_by"])] == case["order_by"]
        assert len(plan.order_by) == len(case["order_by"])
    if "having" in case:
        assert [[h.lhs.value, h.op, h.rhs.value] for h in plan.having] == case["having"]
    if "group_by" in case:
        assert list(plan.group_by) == case["group_by"]
    if "aggregates" in case:
        assert len(plan.aggregates) == case["aggregates"]


@pytest.mark.parametrize("case", GATE["decline"], ids=lambda c: c["q"][:70])
def test_gate_declines_what_the_reference_declines(case):
    with pytest.raises(HipDeclined):
        build_plan(case["q"], GATE["tables"])


@pytest.mark.parametrize("case", GATE["raise"], ids=lambda c: c["q"][:70])
def test_gate_raises_on_user_mistakes(case):
    with pytest.raises(ValueError, match=case["match"]) as ei:
        build_plan(case["q"], GATE["tables"])
    assert not isinstance(ei.value, HipDeclined)


def test_hidden_order_column_rides_along_and_is_marked():
    plan = build_plan("SELECT a.start AS s FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval ORDER BY b.score DESC",
                      ["peaks", "genes"])
    assert [(p.side, p.column, p.name) for p in plan.projection] == [("l", "start", "s"), ("r", "score", "__giql_o0")]
    assert plan.order_by == (("__giql_o0", True, False),)   # DESC: NULLs last unless the query says otherwise


def test_using_must_name_both_chromosome_columns():
    from giql_amd.table import Table

    tables = [Table("peaks", chrom_col="seqid"), Table("genes", chrom_col="seqid")]
    q = "SELECT a.name FROM peaks a JOIN genes b USING ({}) WHERE a.interval INTERSECTS b.interval"
    assert build_plan(q.format("seqid"), tables).kind == "INNER"
    with pytest.raises(HipDeclined):
        build_plan(q.format("chrom"), tables)
    with pytest.raises(HipDeclined):   # the chromosome columns differ between the two tables
        build_plan(q.format("seqid"), [Table("peaks", chrom_col="seqid"), "genes"])
