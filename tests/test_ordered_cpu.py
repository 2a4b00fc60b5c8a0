"""OrderedAggregate: the pure-Python restatement (tests/ordered_oracle.py) pinned on the reference's own vectors,
and the library's Draw string — no GPU needed."""
import pytest

from frostdb_amd.logicalplan import Col, DynCol, Sum
from tests.golden.ordered_cases import ORDERED_CASES
from tests.ordered_oracle import SUM, OrderedAggregate


def run_case(case):
    o = OrderedAggregate(SUM, "vals", [("group%d" % i, False) for i in range(case["ncols"])])
    for groups, vals in case["records"]:
        rec = {"group%d" % i: [g or None for g in col] for i, col in enumerate(groups) if col}
        rec["vals"] = [v or None for v in vals]
        o.callback(rec)
    return o.finish()["rows"]


@pytest.mark.parametrize("case", ORDERED_CASES, ids=[c["id"] for c in ORDERED_CASES])
def test_python_restatement_reproduces_the_reference_vectors(case):
    assert run_case(case) == case["expected"], case["cite"]


def test_python_restatement_dynamic_columns_come_and_go():
    """TestOrderedAggregateDynCols (ordered_aggregate_test.go:253-343): labels.0 is always there, labels.i only in record i; the
    group value never changes, so there are four groups of ten rows each."""
    o = OrderedAggregate(SUM, "value", [("labels", True)])
    for i in range(4):
        rec = {"labels.0": [b"group"] * 10}
        if i:
            rec["labels.%d" % i] = [b"group"] * 10
        rec["value"] = [1] * 10
        o.callback(rec)
    out = o.finish()
    assert len(out["rows"]) == 4 and len(out["columns"]) == 5
    assert all(r[-1] == 10 for r in out["rows"])


def test_ordered_plans_draw_like_the_reference_and_take_one_aggregation():
    from frostdb_amd import physicalplan as pp
    s = pp.explain(None, [Sum(Col("value"))], [DynCol("labels")], ordered=True)
    assert s.startswith("OrderedAggregate (value by labels)")  # ordered_aggregate.go:154-158
    with pytest.raises(pp.UnsupportedError):
        pp.explain(None, [Sum(Col("value")), Sum(Col("other"))], [Col("a")], ordered=True)
