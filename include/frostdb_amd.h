/* frostdb_amd.h — C ABI of the MI355X-native TableScan → PredicateFilter → HashAggregate path.
 *
 * This is the thin cgo surface a FrostDB maintainer binds (see INTEGRATION.md for the Go stub).
 * The reference has no FFI today; the seam is the Go push-operator interface
 *
 *     type PhysicalPlan interface {                     // query/physicalplan/physicalplan.go:24-30
 *         Callback(ctx, arrow.Record) error             //   → fdb_plan_push / fdb_plan_push_batch
 *         Finish(ctx) error                             //   → fdb_plan_finish
 *         SetNext(PhysicalPlan)                         //   (stays in Go: the shim forwards the finish record)
 *         Draw() *Diagram                               //   → fdb_plan_draw
 *         Close()                                       //   → fdb_plan_close
 *     }
 *
 * One fdb_plan replaces one operator chain  PredicateFilter → HashAggregate(final=false)
 * (physicalplan.go:417-474). Plain pointers and sizes only; record batches cross the boundary as
 * Arrow C Data Interface structs (arrow-go: arrow/cdata; pyarrow: _export_to_c).
 *
 * Every function returns FDB_OK (0) or an fdb_status error code and never aborts the process;
 * the text of the last error is available per handle (fdb_plan_last_error) or per thread
 * (fdb_last_error) — the Go shim turns rc != 0 into `errors.New(text)`.
 */
#ifndef FROSTDB_AMD_H
#define FROSTDB_AMD_H

#include <stdint.h>
#include "arrow_c_data.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the prototypes marked FDB_API are its whole dynamic symbol surface
 * (tests/test_capi_cpu.py compares `nm -D --defined-only` with this header), so a cgo binary that links other C++ sees none of
 * the library's internals. */
#if defined(__GNUC__) || defined(__clang__)
#define FDB_API __attribute__((visibility("default")))
#else
#define FDB_API
#endif

typedef enum fdb_status {
  FDB_OK = 0,
  FDB_ERR_INVALID = 1,      /* malformed descriptor / arguments */
  FDB_ERR_UNSUPPORTED = 2,  /* ≙ ErrUnsupportedBooleanExpression (filter.go:46), ErrUnsupportedBinaryOperation
                               (binaryscalarexpr.go:82), ErrUnsupportedSumType/MinType/MaxType (aggregate.go:736,782,862) */
  FDB_ERR_NOT_FOUND = 3,    /* ≙ "aggregate field(s) not found" (aggregate.go:367-380) */
  FDB_ERR_DEVICE = 4,       /* HIP runtime error (text carries hipGetErrorString) */
  FDB_ERR_OOM = 5,          /* host or device allocation failed */
  FDB_ERR_STATE = 6         /* call order violated (push after finish, …) */
} fdb_status;

/* logicalplan.Op (query/logicalplan/expr.go:17-35) — same numeric values as the reference's iota
 * and as storage.proto's Op enum, so a Go shim can pass `int32(expr.Op)` through unchanged. */
typedef enum fdb_op {
  FDB_OP_UNKNOWN = 0,
  FDB_OP_EQ = 1,
  FDB_OP_NOT_EQ = 2,
  FDB_OP_LT = 3,
  FDB_OP_LT_EQ = 4,
  FDB_OP_GT = 5,
  FDB_OP_GT_EQ = 6,
  FDB_OP_REGEX_MATCH = 7,
  FDB_OP_REGEX_NOT_MATCH = 8,
  FDB_OP_AND = 9,
  FDB_OP_OR = 10,
  FDB_OP_ADD = 11,
  FDB_OP_SUB = 12,
  FDB_OP_MUL = 13,
  FDB_OP_DIV = 14,
  FDB_OP_CONTAINS = 15,
  FDB_OP_NOT_CONTAINS = 16
} fdb_op;

/* logicalplan.AggFunc (query/logicalplan/expr.go:718-729), same numeric values. */
typedef enum fdb_agg_func {
  FDB_AGG_UNKNOWN = 0,
  FDB_AGG_SUM = 1,
  FDB_AGG_MIN = 2,
  FDB_AGG_MAX = 3,
  FDB_AGG_COUNT = 4,
  FDB_AGG_AVG = 5,    /* never reaches the operator: lowered to SUM+COUNT+Projection (logicalplan/builder.go:205-238) */
  FDB_AGG_UNIQUE = 6, /* int64 only: the group's value if all its rows carry the same non-NULL value, else NULL (aggregate.go:677-732) */
  FDB_AGG_AND = 7     /* bool only: AND over the valid values, true if there is none (aggregate.go:635-675) */
} fdb_agg_func;

/* scalar.Scalar of a LiteralExpr (the right-hand side of a filter leaf, filter.go:95-103). */
typedef enum fdb_literal_type {
  FDB_LIT_NULL = 0,   /* scalar.ScalarNull: `col = null` ≙ IS NULL, `col != null` ≙ IS NOT NULL */
  FDB_LIT_INT64 = 1,
  FDB_LIT_UINT64 = 2,
  FDB_LIT_FLOAT64 = 3,
  FDB_LIT_STRING = 4, /* scalar.String */
  FDB_LIT_BINARY = 5, /* scalar.Binary */
  FDB_LIT_BOOL = 6
} fdb_literal_type;

typedef struct fdb_literal {
  int32_t type;      /* fdb_literal_type */
  int32_t _pad;
  int64_t i64;       /* INT64, BOOL (0/1) */
  uint64_t u64;      /* UINT64 */
  double f64;        /* FLOAT64 */
  const char* data;  /* STRING / BINARY bytes (not NUL-terminated) */
  int64_t len;
} fdb_literal;

/* One node of the filter expression tree, flattened into an array.
 *   leaf:    op ∈ {EQ … GT_EQ, REGEX_*, CONTAINS, NOT_CONTAINS}, `column` ⟨op⟩ `literal`
 *            (left must be a column, right a literal: filter.go:79-103)
 *   branch:  op ∈ {AND, OR}, `left`/`right` are indices into the same array (filter.go:129-160) */
typedef struct fdb_expr {
  int32_t op;          /* fdb_op */
  int32_t left;        /* child index for AND/OR, else -1 */
  int32_t right;       /* child index for AND/OR, else -1 */
  int32_t _pad;
  const char* column;  /* NUL-terminated exact column name (ArrayRef.ColumnName, binaryscalarexpr.go:18-29) */
  fdb_literal literal;
} fdb_expr;

/* One AggregationFunction (aggregate.go:104-110): `func(column)`; the output column is named
 * "<func>(<column>)" exactly like AggregationFunction.Name() (logicalplan/expr.go:700-702). */
typedef struct fdb_aggregation {
  int32_t func;        /* fdb_agg_func */
  int32_t dynamic;     /* 1 ⇔ the aggregated expression is a DynamicColumn: `column` is the set's name and `max(foo)` means max over
                          every `foo.*` column (Aggregate() marks these, aggregate.go:38-46; HashAggregate turns each matching field
                          into a concrete aggregation when a record first carries it, :306-336). The result column is named after
                          the FIELD by a partial-stage plan and `max(foo.bar)` by a final-stage one; a record without a field adds
                          nothing to it; a record with no field of the set is FDB_ERR_NOT_FOUND. sum / min / max / count only.
                          Such a plan is a family of ordinary plans (one more scan per concrete column): fdb_plan_merge works, the
                          single-table entry points (state arrays, hash exchange) return FDB_ERR_UNSUPPORTED. 0 otherwise. */
  const char* column;
} fdb_aggregation;

/* One group-by matcher (aggregate.go:286-289): Column ≙ exact name (logicalplan/expr.go:353-355),
 * DynamicColumn ≙ every column whose name starts with name+"." (expr.go:564-566). */
typedef struct fdb_group_expr {
  const char* name;
  int32_t dynamic;
  int32_t _pad;
} fdb_group_expr;

/* ---- pre-aggregate Projection (physicalplan/project.go:58-395) --------------------------------------
 * `Projection (value * timestamp, stacktrace) - HashAggregate (sum(value * timestamp) by stacktrace)` and
 * `Projection (value, timestamp / 1000 * 1000 as timestamp_bucket) - HashAggregate (sum(value) by timestamp_bucket)`
 * (logictest/testdata/plan/aggregate/aggregate:66-69, plan/aggregate/window) are fused into the scan: a projection gives a
 * NAME to an arithmetic expression over int64 / float64 columns and literals; an aggregation whose `column`, or a plain
 * (non-dynamic) group matcher whose `name`, equals that name reads the computed value instead of a stored column.
 * Semantics of binaryExprProjection (project.go:73-161 and the Add/Sub/Mul/Div loops :163-399): both operands have the same
 * type (int64 or float64; a literal is a constant array of its own type); + - * use the RAW slot values of their operands
 * and always produce a valid value; / yields NULL where the divisor is 0, else Go's quotient (integers truncate toward
 * zero, MinInt64 / -1 wraps). Only the outermost operation's validity survives: a NULL produced by an inner division is
 * read back as its raw slot (0) by the enclosing + - *. */
typedef struct fdb_proj_node {
  int32_t kind;        /* 0 column, 1 literal, 2 binary arithmetic, 3 comparison → bool (boolExprProjection, project.go:401-470:
                          `distinct(labels.label1, value > 0)`; a NULL operand compares false; usable as a group / distinct key),
                          4 convert(left, float64): int64 → float64 of the RAW slot, always valid (convertProjection, project.go:493-556),
                          5 isnull(left) → bool, always valid; `left` must be a column node (isNullProjection, :558-601),
                          6 if(cond) { left } else { right }: int64 branches; cond (node index in `op`) is a comparison / isnull
                            node; a row takes `left`'s raw slot where cond is true, else `right`'s; always valid
                            (ifExprProjection + conditionalAddInt64, :603-702) */
  int32_t op;          /* binary: FDB_OP_ADD / SUB / MUL / DIV; comparison: FDB_OP_EQ … FDB_OP_GT_EQ over numeric children — or over a
                          COLUMN node and a LITERAL node of any filter literal type (string, binary, NULL …): that comparison is evaluated
                          the way a filter leaf is (boolExprProjection runs BooleanExpression.Eval, project.go:409-447 → BinaryScalarExpr,
                          binaryscalarexpr.go:41-152: dictionary / plain string columns, the missing-column rules) —,
                          or FDB_OP_AND / FDB_OP_OR over two comparison nodes (AndExpr / OrExpr, filter.go:172-220) */
  int32_t left;        /* binary: child indices into the projection's node array */
  int32_t right;
  const char* column;  /* column: exact name (ArrayRef.ColumnName) */
  fdb_literal literal; /* literal: INT64, FLOAT64 or UINT64 (arithmetic: both operands of one type, project.go:104-160 — int64, float64 or uint64; uint64 quotients are unsigned, a zero divisor gives NULL); any filter literal on the right of a column comparison */
} fdb_proj_node;

typedef struct fdb_projection {
  const char* name;    /* BinaryExpr.Name() = "value * timestamp" (logicalplan/expr.go:181-183) or the alias (AliasExpr.Name, :1029-1031) */
  const fdb_proj_node* nodes;
  int32_t n_nodes;
  int32_t root;
} fdb_projection;

/* Optional regular-expression engine of the HOST application. The reference compiles `=~` / `!~` literals with Go's regexp
 * (RE2 syntax, filter.go:105-124) and matches unanchored (regexpfilter.go:84-166). The library evaluates a regex leaf once per
 * DISTINCT value (dictionary entry) on the host — never per row, never on the device — so handing that evaluation back to the
 * caller costs one call per distinct value and makes the result identical to the reference's by construction: the Go shim passes
 * a cgo-exported function around regexp.Regexp.Match. Returns 1 = matches, 0 = does not, < 0 = the pattern
 * does not compile / internal error (the call that triggered it fails with FDB_ERR_INVALID). Without one (NULL) the library
 * uses its own RE2-syntax engine (fdb_regex_match below): flags, named groups, POSIX classes, \Q…\E, \A \z \b, Unicode general
 * categories (\pL \p{Lu} \P{Nd} \p{^Zs} \p{Any}) and case folding by orbits under (?i); Unicode SCRIPT classes (\p{Greek}) are not
 * covered and rejected at fdb_plan_create. Must be callable from any thread that calls into the plan. */
typedef int32_t (*fdb_regex_match_fn)(void* user, const char* pattern, int64_t pattern_len, const uint8_t* value, int64_t value_len);

typedef struct fdb_plan_desc {
  const fdb_expr* filter;      /* NULL / n_filter == 0 ⇒ no PredicateFilter in the chain */
  int32_t n_filter;
  int32_t filter_root;         /* index of the root node */
  const fdb_aggregation* aggs; /* n_aggs == 0: with n_groups > 0 the chain is Filter → Distinction (distinct.go:21-170; push/finish emit
                                  the distinct key tuples), with n_groups == 0 a filter-only plan (only fdb_plan_filter / _select) */
  int32_t n_aggs;
  int32_t n_groups;
  const fdb_group_expr* groups;
  int32_t final_stage;         /* 1 ⇒ behave like HashAggregate(finalStage=true): aggregate columns are matched
                                  by result name and COUNT merges by SUM (aggregate.go:340-348, :965-969) */
  int32_t n_projections;       /* computed columns of the Projection between filter and aggregate (0 ⇒ none) */
  const fdb_projection* projections;
  fdb_regex_match_fn regex_match;  /* NULL ⇒ std::regex */
  void* regex_user;                /* passed back as `user` */
  int32_t ordered;                 /* 1 ⇒ the chain's aggregate is an OrderedAggregate (ordered_aggregate.go; planned by
                                      physicalplan.go:433-449 when the scan is ordered by the group columns): exactly ONE aggregation;
                                      the result record is emitted SORTED by the group columns (first-seen column order, bytewise /
                                      numeric ascending, NULLs last — the order of the reference's merge of its ordered sets,
                                      ordered_aggregate.go:449-470, arrowutils/merge.go:84-112); a partial-stage plan names its result
                                      column after the aggregated COLUMN, a final-stage one after the aggregation (:551-557). The
                                      groups and their values are those of the hash aggregate: the reference's ordered sets merge by
                                      key. Input need NOT arrive ordered for the result to be right: records out of key order (several
                                      ordered sets, none at all) cost a sort of the collected runs on the device at Finish. */
  int32_t _pad2;
} fdb_plan_desc;

typedef struct fdb_plan fdb_plan;   /* one operator chain; push is single-threaded per handle (table.go:783-860) */
typedef struct fdb_batch fdb_batch; /* an Arrow record resident in HBM (a cached part / row group) */

/* ---- library ---------------------------------------------------------------------------------- */
FDB_API const char* fdb_version(void);
/* Text of the last error raised on the calling thread by a call that had no plan handle. */
FDB_API const char* fdb_last_error(void);
FDB_API int fdb_device_count(int* n_devices);
/* Measurement aid ("the measured ceiling of a plain read kernel on the same box in the same run"): streams
 * `bytes` of HBM through a load-only kernel `reps` times (hipEvent-timed, one launch each) and returns the best rate. */
FDB_API int fdb_read_ceiling(int device, int64_t bytes, int32_t reps, double* gb_per_s);

/* Host-only self-check of the Arrow C data code every entry point below relies on — what fdb_plan_push reads (column views
 * at any offset, dictionaries with any index width, plain string / binary columns encoded to distinct values + one id per row)
 * and what fdb_plan_finish / fdb_plan_filter write (dictionary, plain string, bool and fixed-width columns): `batch` comes back
 * in `out` with every column's type, values and NULLs unchanged, except that dictionary indices are uint32 and dictionaries with
 * large value types are narrowed. No device is touched, so it runs where there is no GPU. */
FDB_API int fdb_arrow_roundtrip(struct ArrowArray* batch, struct ArrowSchema* schema, struct ArrowArray* out, struct ArrowSchema* out_schema);
/* The library's built-in regular-expression engine (used for `=~` / `!~` when fdb_plan_desc.regex_match is NULL), exposed for
 * host-only checks: RE2 syntax — what Go's regexp compiles (filter.go:105-124) — matched unanchored on the value's bytes like
 * regexp.Regexp.Match (regexpfilter.go:84-166); linear-time (Thompson / Pike), no backreferences or look-around. *matched = 1 / 0;
 * FDB_ERR_INVALID with Go-style wording ("error parsing regexp: …") when the pattern does not compile. Not covered: Unicode script
 * classes (\p{Greek}); the category and case-folding tables are Unicode 13.0 (Go 1.22: 15.0). */
FDB_API int fdb_regex_match(const char* pattern, int64_t pattern_len, const uint8_t* value, int64_t value_len, int32_t* matched);
/* Host-only self-check of the widening step of a big Finish (dictionary indices cross PCIe at the narrowest width their dictionary
 * allows — `width` 1, 2 or 4 BYTES, or -2 / -4: that many BITS per index, rows packed low bits first — and are widened to Arrow's
 * uint32 by host threads): dst[i] = src[i] for i < n, through the same routine (AVX2 / AVX-512 with streaming stores where the CPU has
 * them). `width` -12 / -14: the 2- / 4-bit road THROUGH a rank → index table (what a Finish that ships the ids present uses), with the
 * table t[r] = 3 r + 5: dst[i] = 3 src[i] + 5. No device is touched. */
FDB_API int fdb_selftest_widen(const void* src, int32_t width, uint32_t* dst, int64_t n);

/* ---- plan life cycle (≙ physicalplan.Build for one chain, physicalplan.go:417-474) -------------- */
/* `desc` and everything it points at (expression nodes, names, literals, patterns) is copied: the caller may free it as soon
 * as the call has returned (the Go shim builds it in C memory and frees that right away, integration/go/gpuplan/operator.go). */
FDB_API int fdb_plan_create(const fdb_plan_desc* desc, int device, fdb_plan** out);
/* ≙ PhysicalPlan.Callback: borrows `batch` for the duration of the call only (the reference releases
 * the record right after Callback returns, table.go:808,:827). The record is validated against the plan (errors it
 * would raise are returned by THIS call) and the columns the plan references are copied out before returning. Records
 * above 8 MiB of referenced data are scanned right away; smaller ones — the reference hands records of ≥ 1 024 rows
 * (table.go:780) — are queued in HBM and scanned together, ONE launch per ≈2 M pending rows, or when the plan's state
 * is next needed (finish, merge, num_groups, state_*, push of resident batches …).
 * Column types (Arrow C data formats) the plan can reference: int64 "l", uint64 "L" (filter only), float64 "g", bool "b"
 * (filter leaves, AND aggregation), dictionary<any integer index, utf8 / binary / large variants> (filter leaves, group keys),
 * and plain utf8 / binary / large_utf8 / large_binary "u" "z" "U" "Z" (filter leaves, group keys: encoded to key ids on the host
 * during this call, emitted with their input type). Columns the plan does not reference may have any type: they are not read. */
FDB_API int fdb_plan_push(fdb_plan* plan, struct ArrowArray* batch, struct ArrowSchema* schema);
/* `n` Callbacks in one call, in order (≙ a chain handing over the records it has collected: table.go:783-860 calls Callback once
 * per record; a host behind cgo / JNI pays its boundary crossing once per call instead of once per record). Stops at the first
 * record that fails and returns its error; *n_pushed (may be NULL) = records accepted. */
FDB_API int fdb_plan_push_many(fdb_plan* plan, struct ArrowArray* const* batches, struct ArrowSchema* const* schemas, int32_t n, int32_t* n_pushed);
/* Same, for a record that is already resident in HBM. */
FDB_API int fdb_plan_push_batch(fdb_plan* plan, const fdb_batch* batch);
/* Same, for `n` resident records at once (≙ the TableScan handing a chain every part it owns): one fused kernel
 * launch scans all of them, so per-launch costs are paid once per scan instead of once per record. */
FDB_API int fdb_plan_push_batches(fdb_plan* plan, const fdb_batch* const* batches, int32_t n);
/* ≙ PhysicalPlan.Finish: waits for the device, emits a record (group columns in first-seen field
 * order with their input Arrow type, then one column per aggregation named "<func>(<column>)")
 * (aggregate.go:543-633). The caller owns `out`/`out_schema` and must call their release().
 * A plan that saw no selected rows emits a zero-row record and sets *n_rows = 0.
 * The reference emits SEVERAL records when a plain string / binary key column would pass 2 GiB in one (a key builder answers
 * ErrMaxSizeReached, the group is rolled back and starts a new aggregate: aggregate.go:426-468, optbuilders.go:221-224). Same here:
 * fdb_plan_finish emits the first record, fdb_plan_finish_next every further one (*emitted = 1) until *emitted = 0 — the Go shim hands
 * each to next.Callback like finishAggregate does (aggregate.go:617-624). $FDB_TEST_MAX_KEY_BYTES lowers the limit (tests). */
FDB_API int fdb_plan_finish(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema, int64_t* n_rows);
FDB_API int fdb_plan_finish_next(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema, int64_t* n_rows, int32_t* emitted);
/* ≙ Synchronizer + HashAggregate(final=true) on one device (synchronize.go:31-53): folds the partial
 * table of `src` into `dst` (SUM of sums and counts, MIN of mins, MAX of maxes). `src` stays valid. */
FDB_API int fdb_plan_merge(fdb_plan* dst, fdb_plan* src);
/* ≙ filter() (filter.go:276-323): the compacted record of the rows that satisfy the plan's filter.
 * *n_selected == 0 ⇒ `out`/`out_schema` are left untouched (the reference skips empty records, filter.go:264-266). */
FDB_API int fdb_plan_filter(fdb_plan* plan, struct ArrowArray* batch, struct ArrowSchema* schema,
                    struct ArrowArray* out, struct ArrowSchema* out_schema, int64_t* n_selected);
/* Selection vector only: ascending row indices of the rows that satisfy the filter, written to the
 * caller's host buffer `indices` (capacity ≥ batch length). */
FDB_API int fdb_plan_select(fdb_plan* plan, struct ArrowArray* batch, struct ArrowSchema* schema,
                    uint32_t* indices, int64_t capacity, int64_t* n_selected);
/* The same two for a record that is resident in HBM, with results that STAY in HBM (a PredicateFilter whose consumer is another
 * device stage; also what the compaction is measured with — inputs and outputs resident): *out is a new resident
 * batch holding the compacted record (same columns, types and dictionaries; release it with fdb_batch_release; zero rows when
 * nothing qualifies); `dev_indices` is a DEVICE buffer of `capacity` ≥ fdb_batch_num_rows entries. Three launches: selection
 * bitmap + per-tile counts, their prefix sums, and ONE streaming pass that compacts every column (wave prefix sums, LDS staging,
 * coalesced stores) — rows keep their order, the output is allocated at its exact size. */
FDB_API int fdb_plan_filter_batch(fdb_plan* plan, const fdb_batch* batch, fdb_batch** out, int64_t* n_selected);
/* ≙ PhysicalPlan.Finish for a consumer that lives on the device (another device stage, the cross-GPU exchange): the result record
 * as a RESIDENT batch — group columns (dictionary<uint32> with the plan's distinct values, int64 / uint64, bool) then one column per
 * aggregation, same names and types as fdb_plan_finish. A big hash table (cfg 5: 10 M groups × 32 label columns) is materialised in
 * HBM by the same two column passes and nothing crosses PCIe but the per-column NULL counts; small tables take the host route and
 * are imported back (microseconds). Release with fdb_batch_release; fdb_batch_export gives the Arrow record on the host.
 * Not for plans with aggregations over a dynamic column set. */
FDB_API int fdb_plan_finish_batch(fdb_plan* plan, fdb_batch** out, int64_t* n_rows);
/* filter() over `n` resident records in ONE launch sequence (≙ PredicateFilter.Callback for every record of a scan, filter.go:255-323;
 * what fdb_plan_push_batches is to the aggregate): out[i] / n_selected[i] are record i's compacted record and row count. All or
 * nothing: on an error no output batch is returned. */
FDB_API int fdb_plan_filter_batches(fdb_plan* plan, const fdb_batch* const* batches, int32_t n, fdb_batch** out, int64_t* n_selected);
FDB_API int fdb_plan_select_batch(fdb_plan* plan, const fdb_batch* batch, uint32_t* dev_indices, int64_t capacity, int64_t* n_selected);
/* ≙ PhysicalPlan.Draw: "PredicateFilter (…) - HashAggregate (sum(value) by labels.path)". Owned by the plan. */
FDB_API const char* fdb_plan_draw(fdb_plan* plan);
/* The same string for a descriptor, without creating a plan or touching a device (≙ `explain`: the operator strings of
 * logictest/testdata/plan/{aggregate,filter}/…): validates `desc` like fdb_plan_create, writes at most `capacity` bytes
 * (NUL-terminated) to `buf` and the size needed to `*needed`. */
FDB_API int fdb_plan_explain(const fdb_plan_desc* desc, char* buf, int64_t capacity, int64_t* needed);
FDB_API const char* fdb_plan_last_error(const fdb_plan* plan);
/* ≙ PhysicalPlan.Close: frees every host and device buffer of the plan. NULL is a no-op. */
FDB_API void fdb_plan_close(fdb_plan* plan);

/* ---- cross-process merge support (the RCCL reduce of per-GPU partial tables) ------- */
/* Number of groups currently in the plan's partial table (waits for the device). */
FDB_API int fdb_plan_num_groups(fdb_plan* plan, int64_t* n_groups);
/* The group-key columns only, one row per table slot, in slot order (same types as fdb_plan_finish). */
FDB_API int fdb_plan_partial_keys(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema);
/* Copies the partial accumulator of aggregation `agg` (n_groups × 8 bytes, slot order; int64 or
 * float64 per fdb_plan_agg_type) to `dst`, a host or device pointer (hipMemcpyDefault). */
FDB_API int fdb_plan_partial_state(fdb_plan* plan, int32_t agg, void* dst, int64_t capacity_bytes);
/* Fast path of the cross-GPU merge: when every rank's table has the SAME slot layout (same group columns, same key
 * dictionaries in the same order — the usual case for parts of one table), slot i means the same group everywhere
 * and the raw table arrays can be all-reduced in place, with no key exchange. `signature` hashes the layout
 * (group column names, key values in id order, radix strides, aggregations); equal signatures ⇔ equal layouts. */
FDB_API int fdb_plan_state_signature(fdb_plan* plan, uint64_t* signature, int64_t* n_slots);
/* Raw table array `array` (0: selected-row counts; 1 + j: accumulator of aggregation j) ⇄ `dst`/`src`, a DEVICE
 * pointer of n_slots × 8 bytes. int64 everywhere except float64 SUM; float64 MIN/MAX are stored as order-preserving
 * int64 keys, so integer MIN/MAX reductions are exact for them too. Both calls wait for the plan's stream. */
/* Zero-copy variant: the device address of array 0, the distance between consecutive arrays (in 8-byte elements) and
 * the slot count, so that a collective library can reduce the arrays IN PLACE on the plan's stream (fdb_plan_stream);
 * nothing is synchronised. Dense tables only (n_slots = 0 otherwise). */
FDB_API int fdb_plan_state_pointers(fdb_plan* plan, void** base, int64_t* array_stride, int64_t* n_slots);
FDB_API int fdb_plan_state_read(fdb_plan* plan, int32_t array, void* dst, int64_t capacity_bytes);
FDB_API int fdb_plan_state_write(fdb_plan* plan, int32_t array, const void* src, int64_t bytes);
/* How table array `array` merges across plans / ranks (0 = the row counts, 1 + j = PHYSICAL accumulator j — UNIQUE owns two,
 * see DESIGN.md §3): 0 unused, 1 integer sum, 2 float64 sum, 3 integer min, 4 integer max. *n_arrays = 1 + accumulators. */
FDB_API int fdb_plan_state_arrays(fdb_plan* plan, int32_t* n_arrays);
FDB_API int fdb_plan_state_array_op(fdb_plan* plan, int32_t array, int32_t* op);
/* 'l' (int64) or 'g' (float64): the Arrow format of aggregation `agg`'s output column; 0 until the first push. */
FDB_API int fdb_plan_agg_type(fdb_plan* plan, int32_t agg, char* format_out);

/* ---- high-cardinality merge: hash-partitioned exchange of partial hash tables ("G large") -------------
 * ≙ Synchronizer + HashAggregate(final=true) (synchronize.go:31-53, aggregate.go:340-348) when the partial tables hold
 * millions of groups: no rank can afford every other rank's table, so each group is sent to the ONE rank that owns
 * fingerprint % n_ranks (all-to-all over the 7 xGMI links), merged there, and the result stays sharded.
 *   1. every rank exports its group schema (fdb_plan_group_schema: a zero-row record whose dictionary columns carry the
 *      distinct key values seen so far, int64 key columns as plain int64, plus one zero-row column per typed aggregate),
 *      the host side agrees on one global schema (frostdb_amd/distributed.py) and seeds a fresh plan of the same
 *      descriptor with it (fdb_plan_seed_groups) — now key ids mean the same thing on every rank;
 *   2. fdb_plan_hash_export(local, layout = the seeded plan, n_parts) re-keys and packs the local table on the device into
 *      n_parts contiguous partitions of fixed-size rows (row_words32 × 4 bytes each; counts[p] rows in partition p);
 *   3. after the exchange, fdb_plan_hash_import(seeded plan, rows, n) merges the received rows (SUM/COUNT add, MIN, MAX);
 *   4. fdb_plan_finish on the seeded plan emits this rank's shard of the final groups.
 * The same two calls with n_parts = 1 are the device-only path of fdb_plan_merge between two plans of one GPU. */
FDB_API int fdb_plan_group_schema(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema);
FDB_API int fdb_plan_seed_groups(fdb_plan* plan, struct ArrowArray* schema_record, struct ArrowSchema* schema);
/* *dev_rows: DEVICE pointer owned by `src` until its next push or close (NULL if the table is empty). */
FDB_API int fdb_plan_hash_export(fdb_plan* src, fdb_plan* layout, int32_t n_parts, void** dev_rows, int64_t* counts, int32_t* row_words32);
/* dev_rows: DEVICE pointer to n_rows rows packed for `plan`'s layout; may be released when the call returns. */
FDB_API int fdb_plan_hash_import(fdb_plan* plan, const void* dev_rows, int64_t n_rows);

/* ---- cross-GPU merge behind the C ABI: RCCL over xGMI, no Python in the loop -------------------------------
 * ≙ Synchronizer + HashAggregate(final=true) (synchronize.go:31-53, physicalplan.go:438-471) when the N chains of a query run
 * on N GPUs. The reference runs its N chains in ONE process (physicalplan.go:22, :337-347); both deployment shapes exist here:
 *   one process, N devices:  fdb_comm_init_all(devices, n, comms)   → ncclCommInitAll; each chain's goroutine / thread drives
 *                                                                     its own handle (collective calls block until all joined)
 *   one process per GPU:     rank 0 calls fdb_comm_unique_id, the host ships the 128 bytes over whatever it already has
 *                            (FrostDB: its own control plane; bench.py: the launcher's rendezvous), every rank calls
 *                            fdb_comm_init_rank.
 *   fdb_comm_init_local is the same interface over direct peer-to-peer loads / copies between the ranks' buffers inside one
 *   process — no RCCL: the ranks may share a device (RCCL refuses two ranks on one GPU), which is how the multi-rank paths are
 *   tested on a 1-GPU box, and on an xGMI node it is the low-latency route for the 8 KiB tables of low-cardinality queries.
 * librccl is bound at run time (dlopen: the copy already in the process, else librccl.so.1, else $FDB_RCCL_LIB), so a host
 * without RCCL can still load this library; fdb_comm_unique_id / _init_rank / _init_all then fail with FDB_ERR_UNSUPPORTED. */
typedef struct fdb_comm fdb_comm;
#define FDB_COMM_ID_BYTES 128
FDB_API int fdb_comm_unique_id(uint8_t id[FDB_COMM_ID_BYTES]);
FDB_API int fdb_comm_init_rank(const uint8_t id[FDB_COMM_ID_BYTES], int32_t n_ranks, int32_t rank, int device, fdb_comm** out);
FDB_API int fdb_comm_init_all(const int* devices, int32_t n, fdb_comm** out /* [n] */);
FDB_API int fdb_comm_init_local(const int* devices, int32_t n, fdb_comm** out /* [n] */);
FDB_API int32_t fdb_comm_rank(const fdb_comm* comm);
FDB_API int32_t fdb_comm_size(const fdb_comm* comm);
/* The size the TRANSPORT itself reports for this communicator (RCCL: ncclCommCount) — evidence that an N-GPU merge really ran
 * over N ranks; -1 if the bound library has no such entry point. */
FDB_API int32_t fdb_comm_transport_ranks(fdb_comm* comm);
FDB_API const char* fdb_comm_last_error(const fdb_comm* comm);
FDB_API void fdb_comm_destroy(fdb_comm* comm);
/* Low-cardinality merge (cfgs 2-4): when every rank's dense table has the same slot layout (fdb_plan_state_signature — parts of
 * one table share their dictionaries), slot i means the same group everywhere and the tables are merged on the plan's own
 * stream. Small tables (≤ 32 MiB over all ranks: every configuration of the benchmark's low-cardinality queries): the packed
 * table of every rank is all-gathered — ONE collective — and folded locally in RANK ORDER into the table and its host copy, so
 * float64 sums come out bit-identical on every rank and in every run whatever order the ranks arrived in. Bigger dense tables:
 * the arrays are all-reduced in place, one grouped launch (SUM for counts and sums, integer MIN / MAX for MIN / MAX — float64
 * MIN / MAX live as order-preserving int64 keys). The layout check is one tiny MAX all-reduce issued on the communicator's own
 * stream, so it overlaps the scan kernel. *aligned = 1: every rank now holds the
 * merged table (call fdb_plan_finish on the rank that emits; the others just close). *aligned = 0: layouts differ (or the plan
 * is in hash mode) — nothing was changed, use fdb_plan_exchange. Collective: every rank of `comm` must call it. */
FDB_API int fdb_plan_allreduce(fdb_plan* plan, fdb_comm* comm, int32_t* aligned);
/* General merge (any table mode, any key sets; cfg 5): ranks agree on one group schema (all-gather of column names + distinct key
 * values, union in rank order), every table is re-keyed and hash-partitioned on the device (fdb_plan_hash_export), partitions
 * travel point-to-point to their owners (grouped send / recv: all 7 xGMI links of a GPU busy at once; slices of ≤ 128 MiB per
 * peer), owners merge on the device. The result STAYS SHARDED: *shard is a new plan of the same descriptor holding this rank's
 * share of the final groups (fingerprint % n_ranks == rank) — fdb_plan_finish + fdb_plan_close it. Collective. */
FDB_API int fdb_plan_exchange(fdb_plan* plan, fdb_comm* comm, fdb_plan** shard);

/* Device memory owned by the library right now (≙ the reference's leak-checked allocator, memory.CheckedAllocator.AssertSize(0),
 * logictest/logic_test.go:169-177): blocks handed out by the per-device caching allocator and not yet returned, plus the arenas
 * of live resident batches; `pinned` counts result blocks whose Arrow release callback has not run. Cached-but-idle blocks are
 * not counted. All three are 0 once every plan, batch, communicator and result record has been closed / released. */
FDB_API int fdb_live_allocations(int64_t* device_blocks, int64_t* device_bytes, int64_t* pinned_blocks);

/* ---- resident batches (a part kept in HBM between queries; 288 GB per GPU) ---------------------- */
FDB_API int fdb_batch_import(struct ArrowArray* batch, struct ArrowSchema* schema, int device, fdb_batch** out);
FDB_API int64_t fdb_batch_num_rows(const fdb_batch* batch);
/* Bytes this batch occupies in HBM (values/indices + validity bitmaps; dictionaries stay on the host). */
FDB_API int64_t fdb_batch_device_bytes(const fdb_batch* batch);
FDB_API void fdb_batch_release(fdb_batch* batch);
/* The record of a resident batch as Arrow in host memory (the caller owns `out` / `out_schema` and calls their release()). */
FDB_API int fdb_batch_export(const fdb_batch* batch, struct ArrowArray* out, struct ArrowSchema* out_schema);

/* ---- Parquet column chunks decoded on the device ------------------------------------------------------------
 * ≙ pqarrow/arrow.go:711-823 (writeColumnToArray) + pqarrow/writer/writer.go:391-405: instead of decoding a row group into an
 * arrow.Record on the CPU (one dictionary-builder probe per row) and pushing that, the host hands over the column chunks'
 * BYTES as they sit in the file and gets a resident batch with the layout fdb_batch_import produces — usable with
 * fdb_plan_push_batch(es), fdb_plan_filter_batch, fdb_batch_export. The host side of this call reads page headers, the dictionary
 * page and the headers of RLE / bit-packed runs; definition levels → validity bitmaps, value ranks and the per-row dictionary
 * indices / values are computed in HBM. Covered — the types pqarrow/convert/convert.go:28-102 maps to Arrow, in the encodings and
 * codecs a FrostDB schema can ask for (schema.proto:54-86): flat schemas; BOOLEAN (PLAIN, RLE → bool); INT64 (PLAIN,
 * DELTA_BINARY_PACKED → int64, or uint64 for the logical type Int(64, unsigned)); DOUBLE (PLAIN); BYTE_ARRAY with a dictionary page +
 * RLE_DICTIONARY data pages and / or PLAIN, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY data pages (their values are dictionary-encoded
 * on the host) (→ dictionary<uint32, binary>, convert.go:64-70; utf8 = 1 → dictionary<uint32, utf8>); required or optional; data
 * pages V1 / V2; codecs as listed at `codec`. Repeated (list) columns and everything else: FDB_ERR_UNSUPPORTED (the caller falls
 * back to its Arrow path for that row group). */
typedef struct fdb_parquet_chunk {
  const char* name;        /* field name of the column in the record */
  int32_t physical_type;   /* parquet Type: 0 BOOLEAN, 2 INT64, 5 DOUBLE, 6 BYTE_ARRAY */
  int32_t optional;        /* max definition level: 0 required, 1 optional (> 1: nested / repeated columns — refused) */
  int32_t utf8;            /* BYTE_ARRAY: logical type String; INT64: 1 = logical type Int(64, unsigned) → uint64 column */
  int32_t codec;           /* parquet CompressionCodec of the chunk's pages: 0 UNCOMPRESSED, 1 SNAPPY, 2 GZIP, 4 BROTLI, 6 ZSTD, 7 LZ4_RAW
                              (5, the deprecated LZ4, is read as raw blocks or Hadoop-framed blocks). The compressed pages of a row group
                              are inflated on host threads, page by page in parallel — SNAPPY pages of PLAIN INT64 / DOUBLE values that did
                              not compress (≥ 32 KiB, compressed ≥ 0.9 × plain) on the device instead; the device decodes the values. */
  const uint8_t* data;     /* [dictionary page] data pages …, each preceded by its thrift PageHeader, exactly as in the file */
  int64_t n_bytes;         /* ColumnMetaData.total_compressed_size */
} fdb_parquet_chunk;
FDB_API int fdb_batch_from_parquet(const fdb_parquet_chunk* chunks, int32_t n_chunks, int64_t n_rows, int device, fdb_batch** out);
/* Several row groups in ONE call (≙ the row groups a ParquetConverter is handed one after the other, pqarrow/arrow.go:264-405, and the
 * parts table.go:740-868 iterates): out[g] = the batch of groups[g], each exactly what fdb_batch_from_parquet returns for it. The call
 * walks the page headers of every chunk first, queues every chunk that crosses PCIe as it is on one copy queue — row group after row
 * group, so the link does not idle while a later row group is still on the host —, inflates the compressed pages of all row groups and
 * parses all chunks on host threads side by side, and launches a row group's decode kernels as soon as ITS chunks are parsed. The
 * chunks (or inflated images) of all the call's row groups are in device memory at once: bound a call by bytes, not by row groups.
 * On error no batch is returned (out[0 … n_groups) = NULL). */
typedef struct fdb_parquet_row_group {
  const fdb_parquet_chunk* chunks;  /* the row group's column chunks (one per column of the record) */
  int32_t n_chunks;
  int64_t n_rows;                   /* RowGroup.num_rows */
} fdb_parquet_row_group;
FDB_API int fdb_batches_from_parquet(const fdb_parquet_row_group* groups, int32_t n_groups, int device, fdb_batch** out);
/* Snappy pages inflated on the device (one wave per page, fdb_kernels.h snappy_decode_kernel) — the building block for pages that cross
 * PCIe compressed (pqarrow/arrow.go:711-823 inflates them on the host; so does fdb_batch_from_parquet today). This entry
 * point takes HOST buffers, for tests and measurement: `src` holds the compressed pages (pages[i] = {src_off, dst_off, src_len,
 * dst_len}: where page i starts in `src`, where its bytes go in `dst`, its compressed and uncompressed sizes), they are copied to the
 * device, decoded by ONE launch, and `dst` is copied back. status[i]: 0 = ok, 1 length preamble ≠ dst_len, 2 truncated input,
 * 3 output overrun, 4 copy offset outside the output, 5 output shorter than announced, 6 a copy from more than 65 472 bytes back (the
 * format allows it, no compressor emits it: matches stay inside a 64 KiB fragment) (a page that fails leaves its part of `dst`
 * undefined; the others are unaffected). *kernel_ms (may be NULL): device time of the launch. */
typedef struct fdb_snappy_page { uint64_t src_off; uint64_t dst_off; uint32_t src_len; uint32_t dst_len; } fdb_snappy_page;
FDB_API int fdb_snappy_decode_pages(const uint8_t* src, int64_t src_bytes, const fdb_snappy_page* pages, int32_t n_pages, uint8_t* dst, int64_t dst_bytes,
                            int device, uint32_t* status, double* kernel_ms);

/* ---- measurement hooks (bench.py / rocprof correlation; not needed by the Go shim) -------------- */
/* Algorithmic bytes (values-or-indices + validity of every referenced column, once per
 * row) and accumulated device time in ms (hipEvent pairs on the plan's stream around each scan
 * kernel) since the plan was created; `n_launches` scan-kernel launches. */
FDB_API int fdb_plan_stats(fdb_plan* plan, int64_t* algorithmic_bytes, double* kernel_ms, int64_t* n_launches,
                   int64_t* rows_scanned);
/* fdb_batch_from_parquet, accumulated over the process: calls, wall time of the host part (page-header walk, inflating compressed
 * pages, dictionary pages) and of the device part (copies, pq_* kernels, the waits), bytes of column chunks read and of columns
 * produced. */
FDB_API int fdb_parquet_stats(int64_t* calls, double* host_ms, double* device_ms, int64_t* file_bytes, int64_t* out_bytes);
/* Run-time specialisation (hiprtc): kernels this process compiled, the wall time the compiler took (ms), and code objects it
 * loaded from the on-disk cache ($FDB_JIT_CACHE) instead — what the FIRST query of a shape pays on top of its scan. */
FDB_API int fdb_jit_stats(int64_t* n_compiled, double* compile_ms, int64_t* n_disk_loads);
/* Accumulated device time in ms of the cross-GPU merges this plan took part in (hipEvent pairs on the plan's stream around the
 * collectives of fdb_plan_allreduce / the all-to-all of fdb_plan_exchange); 0 unless timing is enabled. Read it after
 * fdb_plan_finish (or anything else that waits for the plan's stream). */
FDB_API int fdb_plan_merge_ms(fdb_plan* plan, double* merge_ms);
/* Enables/disables the per-launch hipEvent timing above (off by default; costs two events per launch). */
FDB_API int fdb_plan_set_timing(fdb_plan* plan, int32_t enabled);
/* The hipStream_t the plan launches on, as an opaque pointer. */
FDB_API int fdb_plan_stream(fdb_plan* plan, void** stream_out);
/* Kernel geometry knobs for bench.py's variant sweeps: rows_per_thread 0 = load-hoisting slot kernel (default),
 * 4 / 8 = sequential kernel with that many rows per lane; grid_blocks bits 0-19 = persistent grid size (0 = default),
 * bits 25-27 = variant (1/2/3: 512/256/1024-thread workgroups, 4: interpreting kernels only, no run-time specialisation). */
FDB_API int fdb_plan_set_tuning(fdb_plan* plan, int32_t rows_per_thread, int32_t grid_blocks);
/* Reproducible float sums. float64 addition is not associative and the default scan accumulates a workgroup's rows with LDS atomics
 * whose interleaving across waves differs from run to run: SUM(float64) (and AVG, which is lowered to it) agrees with the reference
 * to ~1e-12 relative but not bit for bit between two runs. With `enabled` every wave accumulates into an LDS table of its own, the
 * tables are added up in wave order and the workgroups' tables in workgroup order (no atomics anywhere): the same records pushed in
 * the same calls give the same bits on every run (NOT the reference's bits: it adds in row order, aggregate.go:822-840). Costs LDS
 * (4 tables per workgroup: ~10 % on cfg 2). Only the dense path of the run-time specialised kernel has it: a scan that needs the
 * hash table, the combining cache (table too big for LDS), the interpreting kernels or 4 tables that do not fit LDS answers
 * FDB_ERR_UNSUPPORTED — from the call that launches it: the push, or for queued small host records a later push / Finish — when the
 * plan has a float64 SUM. MIN / MAX / COUNT and integer sums are exact in every mode.
 * Across GPUs: fdb_plan_allreduce's merge of small tables folds in rank order (reproducible, see there); the in-place all-reduce
 * of big dense tables and the exchange of hash tables are not covered. An ordered plan (fdb_plan_desc.ordered) with this flag does
 * not collect runs (their Finish folds cut groups with atomics): it keeps the dense kernel. */
FDB_API int fdb_plan_set_deterministic(fdb_plan* plan, int32_t enabled);
/* Name of the scan kernel the latest push launched ("fdb_plan_kernel" = the run-time specialised kernel,
 * "scan_slots_kernel" / "scan_dense_kernel" = the interpreting kernels, "scan_hash_kernel" = the hash-table path);
 * "" before the first push. The string is static. */
FDB_API const char* fdb_plan_last_kernel(fdb_plan* plan);

#ifdef __cplusplus
}
#endif

#endif /* FROSTDB_AMD_H */
