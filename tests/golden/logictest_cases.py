"""Golden vectors transcribed from the reference's data-driven SQL logic tests.

Source: /root/reference/logictest/testdata/exec/{filter,aggregate}/* (cockroachdb/datadriven format:
``createtable`` / ``insert cols=(…)`` / ``exec`` + expected tab-separated rows; ``null`` is the NULL literal,
logictest/runner.go:27). The SQL of each ``exec`` is transcribed by hand into the logicalplan builders the
SQL front end (sqlparse/) would produce; the expected rows are copied verbatim. Each case cites file:line.

Also here: exec/distinct/* (incl. boolean projections), the two exec/projection vectors that run through the aggregate,
the operator strings of plan/{aggregate,filter}/* (explain), and known answers of the root aggregate_test.go
(TestAggregateInconsistentSchema, TestDurationAggregation, TestAggregationProjection).
Only queries the hot path serves are listed (filter leaves / AND / OR, SUM/MIN/MAX/COUNT by label columns).
AVG is lowered by the reference to SUM + COUNT + a Projection (logicalplan/builder.go:205-238); its cases
carry ``avg_of=(sum_col, count_col)`` and the harness performs the reference's division (integer division
for int64, builder.go:224-226) on the operator output. ``limit`` cases are checked as "a prefix-sized subset".

Schema "default" = dynparquet.SampleDefinitionWithFloat() (logictest/logic_test.go:41): labels.* and
stacktrace are RLE-dictionary strings (→ Arrow dictionary<uint32, binary>, pqarrow/convert/convert.go:64-70),
timestamp/value int64, floatvalue nullable float64. One ``insert`` = one Arrow record (an L0 part).
"""
from frostdb_amd.logicalplan import Literal, And, Col, Count, DynCol, Max, Min, Or, Sum

AGG_FILE = "logictest/testdata/exec/aggregate/aggregate"
NULLS_FILE = "logictest/testdata/exec/aggregate/aggregate_nulls"
FILTER_FILE = "logictest/testdata/exec/filter/filter"
FPROJ_FILE = "logictest/testdata/exec/filter/filter_projection"
WINDOW_FILE = "logictest/testdata/exec/aggregate/window"
MATH_FILE = "logictest/testdata/exec/aggregate/math"
DISTINCT_FILE = "logictest/testdata/exec/distinct/distinct"

# ---- tables -------------------------------------------------------------------------------------

AGG_TABLE = dict(  # aggregate:4-14
    cols=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "stacktrace", "timestamp", "value", "floatvalue"],
    inserts=[
        """
        value1  value2  null    null    stack1  1   1   1.1
        value2  value2  value3  null    stack1  2   2   2.2
        value3  value2  null    value4  stack1  3   3   3.3
        """,
        """
        value4  value2  null    null    stack1  4   4   4.4
        value5  value2  value3  null    stack1  5   5   5.5
        value6  value2  null    value4  stack1  6   6   6.6
        """,
    ],
)

NULLS_TABLE = dict(  # aggregate_nulls:4-8
    cols=["labels.label1", "labels.label2", "stacktrace", "timestamp", "value"],
    inserts=[
        """
        value1  null    stack1  1   1
        null    value2  stack1  2   2
        null    value2  stack1  3   3
        """,
    ],
)

FILTER_TABLE = dict(  # filter:4-8 (and filter_projection:4-8)
    cols=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "stacktrace", "timestamp", "value"],
    inserts=[
        """
        value1  value2  null    null    stack1  1   1
        value2  value2  value3  null    stack1  2   2
        value3  value2  null    value4  stack1  3   3
        """,
    ],
)

WINDOW_TABLE = dict(  # window:6-11
    cols=["labels.label1", "stacktrace", "timestamp", "value"],
    inserts=[
        """
        value1  stack1  120000  1
        value2  stack1  121000  2
        value3  stack1  122000  3
        value4  stack1  123000  4
        """,
    ],
)

L2 = Col("labels.label2")

# ---- aggregate cases: (id, cite, table, filter, aggs, groups, out_cols, expected_rows[, extra]) ---
# out_cols name operator output columns ("sum(value)" …) or group columns, in the SELECT's order.

AGG_CASES = [
    dict(id="sum_by_label2", cite=f"{AGG_FILE}:16-19", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value"))], groups=[L2], out=["sum(value)", "labels.label2"],
         expected=[(21, b"value2")]),
    dict(id="sumfloat_by_label2", cite=f"{AGG_FILE}:21-24", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("floatvalue"))], groups=[L2], out=["labels.label2", "sum(floatvalue)"],
         expected=[(b"value2", "23.100000")]),
    dict(id="max_by_label2", cite=f"{AGG_FILE}:26-29", table=AGG_TABLE, filter=None,
         aggs=[Max(Col("value"))], groups=[L2], out=["labels.label2", "max(value)"],
         expected=[(b"value2", 6)]),
    dict(id="maxfloat_by_label2", cite=f"{AGG_FILE}:31-34", table=AGG_TABLE, filter=None,
         aggs=[Max(Col("floatvalue"))], groups=[L2], out=["labels.label2", "max(floatvalue)"],
         expected=[(b"value2", "6.600000")]),
    dict(id="minfloat_by_label2", cite=f"{AGG_FILE}:36-39", table=AGG_TABLE, filter=None,
         aggs=[Min(Col("floatvalue"))], groups=[L2], out=["labels.label2", "min(floatvalue)"],
         expected=[(b"value2", "1.100000")]),
    dict(id="count_by_label2", cite=f"{AGG_FILE}:41-44", table=AGG_TABLE, filter=None,
         aggs=[Count(Col("value"))], groups=[L2], out=["labels.label2", "count(value)"],
         expected=[(b"value2", 6)]),
    dict(id="avg_by_label2", cite=f"{AGG_FILE}:46-54", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value")), Count(Col("value"))], groups=[L2], out=["labels.label2", "avg"],
         avg_of=("sum(value)", "count(value)"), expected=[(b"value2", 3)]),
    dict(id="avgfloat_by_label2", cite=f"{AGG_FILE}:56-59", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("floatvalue")), Count(Col("floatvalue"))], groups=[L2], out=["labels.label2", "avg"],
         avg_of=("sum(floatvalue)", "count(floatvalue)"), expected=[(b"value2", "3.850000")]),
    dict(id="avg_by_label4_nullgroup", cite=f"{AGG_FILE}:61-65", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value")), Count(Col("value"))], groups=[Col("labels.label4")], out=["labels.label4", "avg"],
         avg_of=("sum(value)", "count(value)"), expected=[(None, 3), (b"value4", 4)]),
    dict(id="sum_count_by_stacktrace", cite=f"{AGG_FILE}:67-70", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value")), Count(Col("value"))], groups=[Col("stacktrace")],
         out=["stacktrace", "sum(value)", "count(value)"], expected=[(b"stack1", 21, 6)]),
    dict(id="sumfloat_count_by_stacktrace", cite=f"{AGG_FILE}:72-75", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("floatvalue")), Count(Col("floatvalue"))], groups=[Col("stacktrace")],
         out=["stacktrace", "sum(floatvalue)", "count(floatvalue)"], expected=[(b"stack1", "23.100000", 6)]),
    dict(id="sum_count_alias_by_stacktrace", cite=f"{AGG_FILE}:77-80", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value")), Count(Col("value"))], groups=[Col("stacktrace")],
         out=["stacktrace", "sum(value)", "count(value)"], expected=[(b"stack1", 21, 6)]),
    dict(id="four_aggs_by_label2", cite=f"{AGG_FILE}:82-85", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value")), Count(Col("value")), Min(Col("value")), Max(Col("value"))], groups=[L2],
         out=["labels.label2", "sum(value)", "count(value)", "min(value)", "max(value)"],
         expected=[(b"value2", 21, 6, 1, 6)]),
    dict(id="four_float_aggs_by_label2", cite=f"{AGG_FILE}:87-90", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("floatvalue")), Count(Col("floatvalue")), Min(Col("floatvalue")), Max(Col("floatvalue"))],
         groups=[L2], out=["labels.label2", "sum(floatvalue)", "count(floatvalue)", "min(floatvalue)", "max(floatvalue)"],
         expected=[(b"value2", "23.100000", 6, "1.100000", "6.600000")]),
    dict(id="sum_where_ts_by_label1", cite=f"{AGG_FILE}:92-100", table=AGG_TABLE, filter=Col("timestamp") >= 1,
         aggs=[Sum(Col("value"))], groups=[Col("labels.label1")], out=["labels.label1", "sum(value)"],
         expected=[(b"value1", 1), (b"value2", 2), (b"value3", 3), (b"value4", 4), (b"value5", 5), (b"value6", 6)]),
    dict(id="sum_by_all_labels", cite=f"{AGG_FILE}:102-110", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value"))], groups=[DynCol("labels")],
         out=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "sum(value)"],
         expected=[(b"value1", b"value2", None, None, 1), (b"value2", b"value2", b"value3", None, 2),
                   (b"value3", b"value2", None, b"value4", 3), (b"value4", b"value2", None, None, 4),
                   (b"value5", b"value2", b"value3", None, 5), (b"value6", b"value2", None, b"value4", 6)]),
    dict(id="sum_by_label3_limit3", cite=f"{AGG_FILE}:112-117", table=AGG_TABLE, filter=None,
         aggs=[Sum(Col("value"))], groups=[Col("labels.label3")], out=["sum(value)", "labels.label3"],
         expected=[(14, None), (7, b"value3")]),
    # aggregate_nulls
    dict(id="nulls_sum", cite=f"{NULLS_FILE}:12-16", table=NULLS_TABLE, filter=None,
         aggs=[Sum(Col("value"))], groups=[L2], out=["labels.label2", "sum(value)"],
         expected=[(b"value2", 5), (None, 1)]),
    dict(id="nulls_max", cite=f"{NULLS_FILE}:18-22", table=NULLS_TABLE, filter=None,
         aggs=[Max(Col("value"))], groups=[L2], out=["labels.label2", "max(value)"],
         expected=[(b"value2", 3), (None, 1)]),
    dict(id="nulls_count", cite=f"{NULLS_FILE}:24-28", table=NULLS_TABLE, filter=None,
         aggs=[Count(Col("value"))], groups=[L2], out=["labels.label2", "count(value)"],
         expected=[(b"value2", 2), (None, 1)]),
    dict(id="nulls_sum_count", cite=f"{NULLS_FILE}:30-34", table=NULLS_TABLE, filter=None,
         aggs=[Sum(Col("value")), Count(Col("value"))], groups=[L2], out=["labels.label2", "sum(value)", "count(value)"],
         expected=[(b"value2", 5, 2), (None, 1, 1)]),
]

# window: `(timestamp/N)*N as timestamp_bucket` is a pre-aggregate Projection (a "next" row);
# the harness materialises that int64 column the way the Projection operator would and feeds the aggregate,
# which pins int64 group keys. cases: (N, aggs, out, expected)
WINDOW_CASES = [
    dict(id="window_1000", cite=f"{WINDOW_FILE}:13-19", bucket=1000, aggs=[Sum(Col("value"))],
         groups=[Col("timestamp_bucket")], out=["sum(value)", "timestamp_bucket"],
         expected=[(1, 120000), (2, 121000), (3, 122000), (4, 123000)]),
    dict(id="window_2000", cite=f"{WINDOW_FILE}:21-25", bucket=2000, aggs=[Sum(Col("value"))],
         groups=[Col("timestamp_bucket")], out=["sum(value)", "timestamp_bucket"],
         expected=[(3, 120000), (7, 122000)]),
    dict(id="window_3000", cite=f"{WINDOW_FILE}:27-31", bucket=3000, aggs=[Sum(Col("value"))],
         groups=[Col("timestamp_bucket")], out=["sum(value)", "timestamp_bucket"],
         expected=[(6, 120000), (4, 123000)]),
    dict(id="window_3000_count", cite=f"{WINDOW_FILE}:33-37", bucket=3000, aggs=[Sum(Col("value")), Count(Col("value"))],
         groups=[Col("timestamp_bucket")], out=["sum(value)", "count(value)", "timestamp_bucket"],
         expected=[(6, 3, 120000), (4, 1, 123000)]),
    dict(id="window_4000", cite=f"{WINDOW_FILE}:39-42", bucket=4000, aggs=[Sum(Col("value"))],
         groups=[Col("timestamp_bucket")], out=["sum(value)", "timestamp_bucket"],
         expected=[(10, 120000)]),
    dict(id="window_5000_by_label", cite=f"{WINDOW_FILE}:44-51", bucket=5000, aggs=[Sum(Col("value"))],
         groups=[Col("labels.label1"), Col("timestamp_bucket")], out=["labels.label1", "timestamp_bucket", "sum(value)"],
         expected=[(b"value1", 120000, 1), (b"value2", 120000, 2), (b"value3", 120000, 3), (b"value4", 120000, 4)]),
    dict(id="window_2000_count_ts", cite=f"{WINDOW_FILE}:53-57", bucket=2000, aggs=[Sum(Col("value")), Count(Col("timestamp"))],
         groups=[Col("timestamp_bucket")], out=["timestamp_bucket", "sum(value)", "count(timestamp)"],
         expected=[(120000, 3, 2), (122000, 7, 2)]),
    dict(id="window_3000_count_ts", cite=f"{WINDOW_FILE}:59-63", bucket=3000, aggs=[Count(Col("timestamp"))],
         groups=[Col("timestamp_bucket")], out=["timestamp_bucket", "count(timestamp)"],
         expected=[(120000, 3), (123000, 1)]),
]

# ---- arithmetic projections (math:4-9 table; expected = the per-row results the reference prints, in insert order) ----
# The fused operator only sees projections under an aggregate, so each case is run as `sum(<expr>) group by timestamp`
# (timestamps 1, 3, 5, 11 are unique: one group per row ⇒ the sums ARE the per-row values), and NULL-ness of a division by
# zero — invisible in a SUM — as `count(value) group by <expr>`.
MATH_TABLE = dict(
    cols=["labels.label1", "timestamp", "value"],
    inserts=[
        """
        value1 1 2
        value1 3 4
        value1 5 6
        value1 11 0
        """,
    ],
)
V, T = Col("value"), Col("timestamp")
MATH_CASES = [
    dict(id="value_times_timestamp", cite=f"{MATH_FILE}:11-17", expr=V * T, expected=[2, 12, 30, 0]),
    dict(id="value_times_2", cite=f"{MATH_FILE}:19-25", expr=V * 2, expected=[4, 8, 12, 0]),
    dict(id="value_times_lit_product", cite=f"{MATH_FILE}:35-41", expr=V * (Literal(2) * 3), expected=[12, 24, 36, 0]),
    dict(id="value_times_ts_times_2_nested", cite=f"{MATH_FILE}:43-49", expr=V * (T * 2), expected=[4, 24, 60, 0]),
    dict(id="value_times_ts_times_2", cite=f"{MATH_FILE}:51-57", expr=V * T * 2, expected=[4, 24, 60, 0]),
    dict(id="value_times_ts_plus_2_minus_1", cite=f"{MATH_FILE}:59-65", expr=V * T + 2 - 1, expected=[3, 13, 31, 1]),
    dict(id="value_times_ts_times_diff", cite=f"{MATH_FILE}:67-73", expr=V * T * (Literal(2) - 1), expected=[2, 12, 30, 0]),
    dict(id="timestamp_div_value", cite=f"{MATH_FILE}:132-138", expr=T / V, expected=[0, 0, 0, None]),
]

# ---- filter cases: (id, cite, filter, expected selected row numbers of FILTER_TABLE (0-based)) ------
# Expected rows in the testdata are the full rows value1/value2/value3 (timestamps 1/2/3) → row 0/1/2.
L = lambda k: Col(f"labels.label{k}")  # noqa: E731
TS = Col("timestamp")

FILTER_CASES = [
    dict(id="ts_eq", cite=f"{FILTER_FILE}:10-13", filter=TS == 2, rows=[1]),
    dict(id="ts_neq", cite=f"{FILTER_FILE}:15-19", filter=TS != 2, rows=[0, 2]),
    dict(id="ts_lt", cite=f"{FILTER_FILE}:21-24", filter=TS < 2, rows=[0]),
    dict(id="ts_le", cite=f"{FILTER_FILE}:26-30", filter=TS <= 2, rows=[0, 1]),
    dict(id="ts_gt", cite=f"{FILTER_FILE}:37-40", filter=TS > 2, rows=[2]),
    dict(id="ts_ge", cite=f"{FILTER_FILE}:42-46", filter=TS >= 2, rows=[1, 2]),
    dict(id="label4_eq", cite=f"{FILTER_FILE}:48-51", filter=L(4) == "value4", rows=[2]),
    dict(id="label1_or_label2", cite=f"{FILTER_FILE}:53-58", filter=Or(L(1) == "value1", L(2) == "value2"), rows=[0, 1, 2]),
    dict(id="missing_neq", cite=f"{FILTER_FILE}:60-65", filter=L(5) != "value4", rows=[0, 1, 2]),
    dict(id="missing_eq_empty", cite=f"{FILTER_FILE}:67-72", filter=L(5) == "", rows=[0, 1, 2]),
    dict(id="regex_and_eq", cite=f"{FILTER_FILE}:74-79", filter=And(L(1).RegexMatch("value."), L(2) == "value2"), rows=[0, 1, 2]),
    dict(id="missing_regex_empty", cite=f"{FILTER_FILE}:81-86", filter=L(5).RegexMatch(""), rows=[0, 1, 2]),
    dict(id="missing_not_regex", cite=f"{FILTER_FILE}:88-93", filter=L(5).RegexNotMatch("foo"), rows=[0, 1, 2]),
    dict(id="regex_missing_eq", cite=f"{FILTER_FILE}:95-98",
         filter=And(L(3).RegexMatch("value."), L(5).RegexMatch(""), L(2) == "value2"), rows=[1]),
    dict(id="regex_eq_neq", cite=f"{FILTER_FILE}:100-104",
         filter=And(L(1).RegexMatch("value."), L(2) == "value2", L(1) != "value3"), rows=[0, 1]),
    dict(id="regex_simple", cite=f"{FILTER_FILE}:106-112", filter=L(1).RegexMatch("value."), rows=[0, 1, 2]),
    dict(id="regex_nomatch", cite=f"{FILTER_FILE}:114-117", filter=L(1).RegexMatch("values."), rows=[]),
    dict(id="regex_and_missing_empty", cite=f"{FILTER_FILE}:119-124", filter=And(L(1).RegexMatch("value."), L(5) == ""), rows=[0, 1, 2]),
    dict(id="regex_and_or", cite=f"{FILTER_FILE}:126-129",
         filter=And(L(3).RegexMatch("value."), Or(L(1) == "value1", L(1) == "value2")), rows=[1]),
    dict(id="or_and", cite=f"{FILTER_FILE}:131-135",
         filter=Or(L(4) == "value4", And(L(2).RegexMatch("value."), L(1) == "value2")), rows=[1, 2]),
    dict(id="eq_null", cite=f"{FILTER_FILE}:137-141", filter=L(4) == None, rows=[0, 1]),  # noqa: E711
    dict(id="neq_null", cite=f"{FILTER_FILE}:143-146", filter=L(4) != None, rows=[2]),  # noqa: E711
    dict(id="missing_gt", cite=f"{FILTER_FILE}:148-151", filter=Col("doesntexist") > 4, rows=[]),
    dict(id="missing_lt", cite=f"{FILTER_FILE}:153-155", filter=Col("doesntexist") < 4, rows=[]),
    dict(id="missing_ge", cite=f"{FILTER_FILE}:157-159", filter=Col("doesntexist") >= 4, rows=[]),
    dict(id="missing_le", cite=f"{FILTER_FILE}:161-163", filter=Col("doesntexist") <= 4, rows=[]),
    dict(id="like", cite=f"{FILTER_FILE}:165-170", filter=Col("stacktrace").Contains("ack"), rows=[0, 1, 2]),
    dict(id="like_nomatch", cite=f"{FILTER_FILE}:172-174", filter=Col("stacktrace").Contains("ack2"), rows=[]),
    dict(id="not_like", cite=f"{FILTER_FILE}:176-178", filter=Col("stacktrace").NotContains("ack"), rows=[]),
    dict(id="not_like_all", cite=f"{FILTER_FILE}:180-185", filter=Col("stacktrace").NotContains("ack2"), rows=[0, 1, 2]),
    dict(id="not_like_and_like", cite=f"{FILTER_FILE}:187-191",
         filter=And(L(1).NotContains("ue2"), Col("stacktrace").Contains("ack")), rows=[0, 2]),
    # filter_projection
    dict(id="proj_ts_ge", cite=f"{FPROJ_FILE}:11-21", filter=TS >= 2, rows=[1, 2]),
    dict(id="proj_null_and_notnull", cite=f"{FPROJ_FILE}:23-27", filter=And(L(5) == None, L(3) != None), rows=[1]),  # noqa: E711
    dict(id="proj_inverse_null", cite=f"{FPROJ_FILE}:29-32", filter=And(L(5) != None, L(3) != None), rows=[]),  # noqa: E711
    dict(id="proj_multi_null", cite=f"{FPROJ_FILE}:34-39",
         filter=Or(And(L(3) == "value3", L(5) == None), And(L(3) == None, L(5) == "a")), rows=[1]),  # noqa: E711
]

# ---- filter_contains: the `bytes` schema (logic_test.go:110-146): dynamic dictionary labels, a UINT64 timestamp and a PLAIN
# (DELTA_LENGTH_BYTE_ARRAY → Arrow binary, no dictionary) `value` column; LIKE / NOT LIKE on the plain column ---------------------
CONTAINS_FILE = "logictest/testdata/exec/filter/filter_contains"
CONTAINS_TABLE = dict(  # filter_contains:4-8
    cols=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "timestamp", "value"],
    rows=[
        [b"value1", b"value2", None, None, 1, b"foo"],
        [b"value2", b"value2", b"value3", None, 2, b"bar"],
        [b"value3", b"value2", None, b"value4", 3, b"baz"],
    ],
)
CONTAINS_CASES = [
    dict(id="bytes_ts_eq", cite=f"{CONTAINS_FILE}:10-13", filter=TS == 2, rows=[1]),
    dict(id="bytes_like", cite=f"{CONTAINS_FILE}:15-19", filter=Col("value").Contains("a"), rows=[1, 2]),
    dict(id="bytes_not_like", cite=f"{CONTAINS_FILE}:21-24", filter=Col("value").NotContains("a"), rows=[0]),
]

# aggregate_test.go:23-148 TestAggregateInconsistentSchema: three single-row records with different label
# sets; GROUP BY labels.label2 (a concrete column that the first record lacks). Expected values, sorted
# descending like the test does (aggregate_test.go:141-145).
INCONSISTENT_SCHEMA = dict(
    records=[
        dict(cols=["labels.label1", "stacktrace", "timestamp", "value"], rows="value1 s 1 1"),
        dict(cols=["labels.label2", "stacktrace", "timestamp", "value"], rows="value2 s 2 2"),
        dict(cols=["labels.label2", "stacktrace", "timestamp", "value"], rows="value2 s 3 3"),
    ],
    cite="aggregate_test.go:85-114",
    expected={"sum": [5, 1], "min": [2, 1], "max": [3, 1], "count": [2, 1], "avg": [2, 1]},
)


# ---- Distinct (distinct:4-8 table): `select distinct(cols…) [where …]` = TableScan → Filter → Distinction ------------------
# A plan with NO aggregations and the distinct columns as group matchers; expected = the distinct rows, in the SELECT's order.
DISTINCT_TABLE = dict(
    cols=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "labels.label5"],
    inserts=[
        """
        value1  value1  null    null    value1
        value2  value2  value3  null    value1
        value3  value1  null    value4  value1
        """,
    ],
)
_b = lambda *xs: tuple(None if x is None else x.encode() for x in xs)  # noqa: E731
DISTINCT_CASES = [
    dict(id="label1", cite=f"{DISTINCT_FILE}:10-15", filter=None, groups=[L(1)], out=["labels.label1"],
         expected=[_b("value1"), _b("value2"), _b("value3")]),
    dict(id="label2", cite=f"{DISTINCT_FILE}:17-21", filter=None, groups=[L(2)], out=["labels.label2"], expected=[_b("value1"), _b("value2")]),
    dict(id="label3_with_null", cite=f"{DISTINCT_FILE}:23-27", filter=None, groups=[L(3)], out=["labels.label3"], expected=[_b(None), _b("value3")]),
    dict(id="label1_label2", cite=f"{DISTINCT_FILE}:29-34", filter=None, groups=[L(1), L(2)], out=["labels.label1", "labels.label2"],
         expected=[_b("value1", "value1"), _b("value2", "value2"), _b("value3", "value1")]),
    dict(id="label1_2_3", cite=f"{DISTINCT_FILE}:38-43", filter=None, groups=[L(1), L(2), L(3)], out=["labels.label1", "labels.label2", "labels.label3"],
         expected=[_b("value1", "value1", None), _b("value2", "value2", "value3"), _b("value3", "value1", None)]),
    dict(id="label1_2_4", cite=f"{DISTINCT_FILE}:45-50", filter=None, groups=[L(1), L(2), L(4)], out=["labels.label1", "labels.label2", "labels.label4"],
         expected=[_b("value1", "value1", None), _b("value2", "value2", None), _b("value3", "value1", "value4")]),
    dict(id="label1_2_5", cite=f"{DISTINCT_FILE}:52-57", filter=None, groups=[L(1), L(2), L(5)], out=["labels.label1", "labels.label2", "labels.label5"],
         expected=[_b("value1", "value1", "value1"), _b("value2", "value2", "value1"), _b("value3", "value1", "value1")]),
    dict(id="label1_2_3_4", cite=f"{DISTINCT_FILE}:59-64", filter=None, groups=[L(1), L(2), L(3), L(4)],
         out=["labels.label1", "labels.label2", "labels.label3", "labels.label4"],
         expected=[_b("value1", "value1", None, None), _b("value2", "value2", "value3", None), _b("value3", "value1", None, "value4")]),
    dict(id="dynamic_labels", cite=f"{DISTINCT_FILE}:67-72", filter=None, groups=[DynCol("labels")],
         out=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "labels.label5"],
         expected=[_b("value1", "value1", None, None, "value1"), _b("value2", "value2", "value3", None, "value1"), _b("value3", "value1", None, "value4", "value1")]),
    dict(id="with_filter", cite=f"{DISTINCT_FILE}:75-78", filter=And(L(2) == "value1", L(4) == "value4"), groups=[L(1)], out=["labels.label1"],
         expected=[_b("value3")]),
]


# ---- Distinct over boolean projections (distinct_proj, distinct_partial_scan_opt): `select distinct(l1, l2, value > 0)` ----------
DPROJ_FILE = "logictest/testdata/exec/distinct/distinct_proj"
DPART_FILE = "logictest/testdata/exec/distinct/distinct_partial_scan_opt"
DPROJ_INSERT_1 = """
        value1  value2  1   0
        value1  value2  2   0
        """
DPROJ_INSERT_2 = """
        value2  value2  1   1
        value2  value2  2   2
        """
DPROJ_COLS = ["labels.label1", "labels.label2", "timestamp", "value"]
DISTINCT_PROJ_CASES = [
    dict(id="always_true", cite=f"{DPROJ_FILE}:9-13", table=dict(cols=DPROJ_COLS, inserts=[DPROJ_INSERT_1]),
         groups=[L(1), L(2), TS > 0], out=["labels.label1", "labels.label2", "timestamp > 0"], expected=[(b"value1", b"value2", True)]),
    dict(id="always_false", cite=f"{DPROJ_FILE}:15-19", table=dict(cols=DPROJ_COLS, inserts=[DPROJ_INSERT_1]),
         groups=[L(1), L(2), Col("value") > 0], out=["labels.label1", "labels.label2", "value > 0"], expected=[(b"value1", b"value2", False)]),
    dict(id="mixed", cite=f"{DPROJ_FILE}:21-32", table=dict(cols=DPROJ_COLS, inserts=[DPROJ_INSERT_1, DPROJ_INSERT_2]),
         groups=[L(1), L(2), Col("value") > 0], out=["labels.label1", "labels.label2", "value > 0"],
         expected=[(b"value1", b"value2", False), (b"value2", b"value2", True)]),
    dict(id="partial_scan_opt", cite=f"{DPART_FILE}:5-25",
         table=dict(cols=["labels.label1", "timestamp", "value"], inserts=["""
        value1  0   1
        value2  1   1
        """, """
        value2  1   1
        value2  1   1
        """]),
         groups=[L(1), TS, Col("value") > 0], out=["labels.label1", "timestamp", "value > 0"],
         expected=[(b"value1", 0, True), (b"value2", 1, True)]),
]


# ---- logictest/testdata/exec/projection: the two vectors that run through this path ---------------------------------------
PMATH_FILE = "logictest/testdata/exec/projection/math_projection"
PBOOL_FILE = "logictest/testdata/exec/projection/bool"
PROJ_MATH_TABLE = dict(
    cols=["labels.label1", "stacktrace", "timestamp", "value"],
    inserts=[
        """
        value1 stack1 1 2
        value1 stack1 3 4
        value1 stack2 5 6
        """,
    ],
)
# `select stacktrace, sum(value * timestamp) group by stacktrace` (math_projection:17-21)
PROJ_MATH_GROUPED = dict(cite=f"{PMATH_FILE}:17-21", aggs=[Sum(V * T)], groups=[Col("stacktrace")], out=["stacktrace", "sum(value * timestamp)"],
                         expected=[(b"stack1", 14), (b"stack2", 30)])
# schema simple_bool (logictest/logic_test.go:43-60): name string (dictionary), found bool; `select name where found = 'true'`
# — the SQL front end turns the quoted true into a boolean literal (sqlparse/visitor.go:265-271) (bool:10-14)
BOOL_TABLE_ROWS = [(b"test0", True), (b"test1", True), (b"test2", False)]
BOOL_FILTER_CASES = [
    dict(id="found_eq_true", cite=f"{PBOOL_FILE}:10-14", filter=Col("found") == True, rows=[0, 1]),  # noqa: E712
]


# ---- root aggregate_test.go: end-to-end known answers that run through this path ------------------------------------------
# TestDurationAggregation (aggregate_test.go:260-343): group by logicalplan.Duration(time.Second) — HashAggregate only uses the
# expression's MatchColumn, i.e. it groups by the stored `timestamp` column (logicalplan/expr.go:1127-1129); sums 16 and 5.
DURATION_CASE = dict(
    cite="aggregate_test.go:260-343",
    rows=[(1_000_000_000, b"stack1", 3), (1_000_000_000, b"stack2", 5), (1_000_000_000, b"stack3", 8), (2_000_000_000, b"stack1", 2), (2_000_000_000, b"stack2", 3)],
    aggs=[Sum(Col("value"))], groups=[Col("timestamp")], out=["timestamp", "sum(value)"],
    expected=[(1_000_000_000, 16), (2_000_000_000, 5)],
)
# TestAggregationProjection (aggregate_test.go:150-258): three single-row records with different label sets, sum + max by
# (labels, timestamp): 3 rows; every label column seen, timestamp and both aggregates are in the result.
AGG_PROJECTION_CASE = dict(
    cite="aggregate_test.go:150-258",
    records=[
        dict(cols=["labels.label1", "labels.label2", "timestamp", "value"], rows="value1 value2 1 1"),
        dict(cols=["labels.label1", "labels.label2", "labels.label3", "timestamp", "value"], rows="value2 value2 value3 2 2"),
        dict(cols=["labels.label1", "labels.label2", "labels.label4", "timestamp", "value"], rows="value3 value2 value4 3 3"),
    ],
    aggs=[Sum(Col("value")), Max(Col("value"))], groups=[DynCol("labels"), Col("timestamp")],
    fields=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "timestamp", "sum(value)", "max(value)"],
    out=["labels.label1", "labels.label2", "labels.label3", "labels.label4", "timestamp", "sum(value)", "max(value)"],
    expected=[(b"value1", b"value2", None, None, 1, 1, 1), (b"value2", b"value2", b"value3", None, 2, 2, 2), (b"value3", b"value2", None, b"value4", 3, 3, 3)],
)


# ---- explain vectors: the operator strings (PhysicalPlan.Draw) of the fused operators -------------------------------------
# logictest/testdata/plan/{aggregate/aggregate, aggregate/window, filter/filter}: each expected string is the fragment(s) of the
# reference's explain line that belong to the operators this library replaces (PredicateFilter, the per-chain HashAggregate).
def _explain_cases():
    from frostdb_amd.logicalplan import And, Col, Count, DynCol, Max, Min, Sum
    V, T = Col("value"), Col("timestamp")
    A = "logictest/testdata/plan/aggregate/aggregate"
    return [
        dict(id="static_and_dynamic_member", cite=A + ":8-10", filter=None, aggs=[Sum(V)], groups=[Col("example_type"), Col("labels.label1")],
             expected="HashAggregate (sum(value) by example_type,labels.label1)"),
        dict(id="static_and_dynamic_set", cite=A + ":14-16", filter=None, aggs=[Sum(V)], groups=[Col("example_type"), DynCol("labels")],
             expected="HashAggregate (sum(value) by example_type,labels)"),
        dict(id="dynamic_set_first", cite=A + ":21-23", filter=None, aggs=[Sum(V)], groups=[DynCol("labels"), Col("example_type")],
             expected="HashAggregate (sum(value) by labels,example_type)"),
        dict(id="eq_filter", cite=A + ":35-37", filter=Col("example_type") == "some_value", aggs=[Sum(V)], groups=[DynCol("labels")],
             expected="PredicateFilter (example_type == some_value) - HashAggregate (sum(value) by labels)"),
        dict(id="eq_filter_two_groups", cite=A + ":40-42", filter=Col("example_type") == "some_value", aggs=[Sum(V)], groups=[DynCol("labels"), T],
             expected="PredicateFilter (example_type == some_value) - HashAggregate (sum(value) by labels,timestamp)"),
        dict(id="gt_filter", cite=A + ":46-48", filter=Col("example_type") > "some_value", aggs=[Sum(V)], groups=[DynCol("labels")],
             expected="PredicateFilter (example_type > some_value) - HashAggregate (sum(value) by labels)"),
        dict(id="int_filter", cite=A + ":52-54", filter=T >= 1, aggs=[Sum(V)], groups=[Col("labels.label1")],
             expected="PredicateFilter (timestamp >= 1) - HashAggregate (sum(value) by labels.label1)"),
        dict(id="two_aggs", cite=A + ":57-59", filter=None, aggs=[Sum(V), Count(V)], groups=[Col("labels.label2")],
             expected="HashAggregate (sum(value),count(value) by labels.label2)"),
        dict(id="math_agg", cite=A + ":67-69", filter=None, aggs=[Sum(V * T)], groups=[Col("stacktrace")],
             expected="HashAggregate (sum(value * timestamp) by stacktrace)"),
        dict(id="avg_lowered", cite=A + ":72-74", filter=None, aggs=[Sum(V), Count(V)], groups=[Col("stacktrace")],
             expected="HashAggregate (sum(value),count(value) by stacktrace)"),
        dict(id="four_aggs", cite=A + ":77-79", filter=None, aggs=[Max(V), Min(V), Sum(V), Count(V)], groups=[Col("labels.label1")],
             expected="HashAggregate (max(value),min(value),sum(value),count(value) by labels.label1)"),
        dict(id="window_alias_key", cite="logictest/testdata/plan/aggregate/window:7-9", filter=None, aggs=[Sum(V)],
             groups=[(T / 1000 * 1000).Alias("timestamp_bucket")], expected="HashAggregate (sum(value) by timestamp_bucket)"),
        dict(id="contains", cite="logictest/testdata/plan/filter/filter:5-7", filter=Col("stacktrace").Contains("ack"), aggs=[], groups=[],
             expected="PredicateFilter (stacktrace contains ack)"),
        dict(id="not_contains", cite="logictest/testdata/plan/filter/filter:10-12", filter=Col("stacktrace").NotContains("ack"), aggs=[], groups=[],
             expected="PredicateFilter (stacktrace not contains ack)"),
        dict(id="and_of_contains", cite="logictest/testdata/plan/filter/filter:15-17",
             filter=And(Col("labels.label1").NotContains("ue2"), Col("stacktrace").Contains("ack")), aggs=[], groups=[],
             expected="PredicateFilter ((labels.label1 not contains ue2 AND stacktrace contains ack))"),
    ]


EXPLAIN_CASES = _explain_cases()
