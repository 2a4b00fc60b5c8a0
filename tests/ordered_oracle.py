"""Pure-Python restatement of the reference's OrderedAggregate (query/physicalplan/ordered_aggregate.go) for SMALL test cases —
test infrastructure, like oracle/.

It follows the operator's own structure rather than "group by, then sort": records are cut into GROUPS (consecutive rows with the
same key tuple) and ORDERED SETS (a new set starts where some group column compares lower than in the previous row) by the
per-column scan of arrowutils.GetGroupsAndOrderedSetRanges (pqarrow/arrowutils/groupranges.go:12-211 — there a NULL compares
LOWER than any value, nullComparison :213-227); the last group of a record is carried into the next call
(ordered_aggregate.go:289-307); Finish aggregates the carry, and, if more than one ordered set was seen, merges the sets' records
sorted by every group column ascending with NULLs LAST (ordered_aggregate.go:449-470 → arrowutils.MergeRecords with zero-value
SortingColumns, merge.go:84-112) and aggregates runs of equal keys once more (final-stage semantics: counts are summed,
aggregate.go:965-969).
"""
from functools import cmp_to_key

SUM, MIN, MAX, COUNT = 1, 2, 3, 4


def _null_cmp(left_null, right_null):  # groupranges.go:213-227
    if not left_null and not right_null:
        return 0, False
    if left_null:
        return (0, True) if right_null else (-1, True)
    return 1, True


def _cmp(a, b):
    return (a > b) - (a < b)


def group_and_set_ranges(cur_group, columns, n):
    """≙ GetGroupsAndOrderedSetRanges: sorted unique row indices where a group / an ordered set begins, and the last group."""
    cur = list(cur_group)
    groups, sets = set(), set()
    for c, col in enumerate(columns):
        for j in range(n):
            v = col[j] if col is not None else None
            cmp, ok = _null_cmp(cur[c] is None, v is None)
            if not ok:
                cmp = _cmp(cur[c], v)
            if cmp != 0:
                groups.add(j)
                if cmp == 1:
                    sets.add(j)
                cur[c] = v
    return sorted(groups), sorted(sets), cur


def run_aggregation(final_stage, func, arrays):  # aggregate.go:955-971 + the reducers' raw-slot semantics
    out = []
    for a in arrays:
        raw = [0 if v is None else v for v in a]  # a NULL slot reads as its zeroed raw value (optbuilders.go:337-340)
        if func == SUM or (func == COUNT and final_stage):
            out.append(sum(raw))
        elif func == COUNT:
            out.append(len(a))
        elif func == MIN:
            out.append(min(raw) if raw else None)
        else:
            out.append(max(raw) if raw else None)
    return out


class OrderedAggregate:
    def __init__(self, func, agg_column, group_matchers, final_stage=True):
        """group_matchers: [(name, dynamic)]"""
        self.func, self.col, self.matchers, self.final = func, agg_column, group_matchers, final_stage
        self.order = []            # groupColOrdering
        self.cur = {}              # curGroup
        self.first = True
        self.builders = {}         # group values of the current ordered set
        self.group_results = []    # per closed set: {name: [values]}
        self.carry = []            # arrayToAggCarry
        self.agg_builder = []      # aggregation results of the current set
        self.agg_results = []      # per closed set

    def _matches(self, name):
        return any((name.startswith(m + ".") if dyn else name == m) for m, dyn in self.matchers)

    def callback(self, rec):
        """rec: {column name: list of values} (insertion order = field order)."""
        n = len(next(iter(rec.values())))
        by_name = {}
        found_new = False
        for name in rec:
            if self._matches(name):
                by_name[name] = rec[name]
                if name not in self.builders:
                    self.order.append(name)
                    # a column that appears later is NULL for the groups buffered so far (:197-204)
                    self.builders[name] = [None] * (len(self.builders[self.order[0]]) if not self.first and self.order[:-1] else 0)
                    found_new = True
        if self.col not in rec:
            raise KeyError("aggregate field not found, aggregations are not possible without it")
        vals = rec[self.col]
        if found_new:
            for gr in self.group_results:
                k = len(next(iter(gr.values()))) if gr else 0
                for name in self.order:
                    gr.setdefault(name, [None] * k)
        arrays = []
        for name in self.order:
            col = by_name.get(name)  # None ≙ a virtual NULL column (:236-241)
            arrays.append(col)
            if self.first:
                self.cur[name] = col[0] if col is not None else None
        scratch_cur = [self.cur.get(name) for name in self.order]
        self.first = False
        groups, sets, last = group_and_set_ranges(scratch_cur, arrays, n)
        to_agg, set_idxs = [], []
        start, set_cursor = 0, 0
        for end in groups:
            if end == 0:
                chunk = self.carry
                self.carry = []
            else:
                chunk = self.carry + list(vals[start:end])
                self.carry = []
            to_agg.append(chunk)
            new_set = set_cursor < len(sets) and sets[set_cursor] == end
            if new_set:
                set_cursor += 1
                set_idxs.append(len(to_agg))
                self.group_results.append({})
            for i, name in enumerate(self.order):
                v = self.cur.get(name) if end == 0 else (arrays[i][start] if arrays[i] is not None else None)
                self.builders[name].append(v)
                if new_set:
                    self.group_results[-1][name] = self.builders[name]
                    self.builders[name] = []
            start = end
        self.carry = self.carry + list(vals[start:n])  # the last group may continue in the next record (:289-307)
        for i, name in enumerate(self.order):
            self.cur[name] = last[i]
        if not to_agg:
            return
        results = run_aggregation(self.final, self.func, to_agg)
        s0 = 0
        for s1 in set_idxs:
            self.agg_results.append(self.agg_builder + results[s0:s1])
            self.agg_builder = []
            s0 = s1
        self.agg_builder = self.agg_builder + results[s0:]

    def finish(self):
        if self.first:
            return None
        if self.carry:
            self.group_results.append({})
            for name in self.order:
                self.builders[name].append(self.cur.get(name))
                self.group_results[-1][name] = self.builders[name]
                self.builders[name] = []
            last = run_aggregation(self.final, self.func, [self.carry])
            self.agg_results.append(self.agg_builder + last)
            self.agg_builder = []
        result_name = self.col if not self.final else {SUM: "sum", MIN: "min", MAX: "max", COUNT: "count"}[self.func] + "(" + self.col + ")"
        records = []
        for gr, ar in zip(self.group_results, self.agg_results):
            records.append([tuple(gr.get(name, [None] * len(ar))[i] for name in self.order) + (ar[i],) for i in range(len(ar))])
        if len(records) == 1:
            rows = records[0]
        else:
            def less(a, b):  # cursorHeap.Less with ascending, NullsFirst = false (merge.go:84-112)
                for x, y in zip(a[:-1], b[:-1]):
                    cmp, ok = _null_cmp(x is None, y is None)
                    if ok:
                        return -1 if cmp == 1 else (1 if cmp == -1 else 0)
                    c = _cmp(x, y)
                    if c:
                        return c
                return 0
            merged = sorted([r for rec in records for r in rec], key=cmp_to_key(less))
            rows, i = [], 0
            while i < len(merged):
                j = i
                while j < len(merged) and merged[j][:-1] == merged[i][:-1]:
                    j += 1
                rows.append(merged[i][:-1] + tuple(run_aggregation(True, self.func, [[r[-1] for r in merged[i:j]]])))
                i = j
        return {"columns": self.order + [result_name], "rows": rows}
