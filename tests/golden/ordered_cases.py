"""query/physicalplan/ordered_aggregate_test.go: TestOrderedAggregate's vectors (:28-148; "" is a NULL group value, 0 a NULL value)
and TestOrderedAggregateDynCols (:253-343)."""
ORDERED_CASES = [
    dict(id="SingleGroupCol", ncols=1, records=[([list("aabcc")], [1, 1, 1, 1, 1])], expected=[("a", 2), ("b", 1), ("c", 2)], cite="ordered_aggregate_test.go:36-55"),
    dict(id="MultipleRecords", ncols=1, records=[([list("aaa")], [1, 1, 1]), ([list("bb")], [1, 1])], expected=[("a", 3), ("b", 2)], cite=":56-81"),
    dict(id="MultiGroupCol", ncols=2, records=[([list("aaacd"), list("bbccd")], [1, 1, 1, 1, 1])],
         expected=[("a", "b", 2), ("a", "c", 1), ("c", "c", 1), ("d", "d", 1)], cite=":82-103"),
    dict(id="PartialOrdering", ncols=1, records=[([list("aabcabc")], [1, 1, 2, 3, 1, 2, 3])], expected=[("a", 3), ("b", 4), ("c", 6)], cite=":104-123"),
    dict(id="PartialOrderingMultiRecord", ncols=1, records=[([list("aabc")], [1, 1, 2, 3]), ([list("abc")], [1, 2, 3])],
         expected=[("a", 3), ("b", 4), ("c", 6)], cite=":124-148"),
]
