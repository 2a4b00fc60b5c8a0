// Package gpuplan binds libfrostdb_amd.so (include/frostdb_amd.h) into FrostDB's physical plan: one Operator replaces one
// chain's PredicateFilter + HashAggregate(final=false) (query/physicalplan/physicalplan.go:417-474).
//
// This file lives at query/physicalplan/gpuplan/operator.go in a FrostDB checkout. It is real source, not pseudo-code, but this
// repository's build image has no Go toolchain, so it has never been compiled here; what CAN be checked without a compiler is checked
// by tests/test_go_shim_cpu.py (every C.fdb_* identifier and struct field against the header, the import graph, balanced syntax).
//
// Import graph (no cycle): gpuplan imports physicalplan and logicalplan; physicalplan never learns about gpuplan — Build consults an
// OperatorFactory option (integration/go/patches/physicalplan_operator_factory.diff) and the USER passes gpuplan.Factory(device):
//
//	engine := query.NewEngine(pool, provider,
//		query.WithPhysicalplanOptions(physicalplan.WithOperatorFactory(gpuplan.Factory(0))))
//
// Every identifier of the reference this file touches:
//
//	physicalplan.PhysicalPlan {Callback, Finish, SetNext, Draw, Close}   query/physicalplan/physicalplan.go:24-30
//	physicalplan.Diagram {Details, Child}                                query/physicalplan/physicalplan.go:558-561
//	physicalplan.ErrUnsupportedBooleanExpression                         query/physicalplan/filter.go:46
//	physicalplan.FusedStage, OperatorFactory, ErrOperatorNotFused,
//	  WithOperatorFactory                                                (new: the patch above, next to execOptions :260-285)
//	logicalplan.Expr {Name()}                                            query/logicalplan/builder.go:69-101
//	logicalplan.BinaryExpr {Left, Op, Right}                             query/logicalplan/expr.go:105-109
//	logicalplan.Op, OpAnd, OpOr, OpAdd, OpSub, OpMul, OpDiv              query/logicalplan/expr.go:13-35
//	logicalplan.Column {ColumnName}                                      query/logicalplan/expr.go:292-294
//	logicalplan.DynamicColumn {ColumnName}                               query/logicalplan/expr.go:518-520
//	logicalplan.LiteralExpr {Value scalar.Scalar}                        query/logicalplan/expr.go:586-588
//	logicalplan.AggregationFunction {Func AggFunc, Expr}                 query/logicalplan/expr.go:648-651
//	logicalplan.AliasExpr {Expr, Alias}                                  query/logicalplan/expr.go:1000-1003
//	logicalplan.Aggregation {AggExprs, GroupExprs}                       query/logicalplan/logicalplan.go:409-412
//	arrow.Record; array.Concatenate, array.NewRecord; memory.Allocator;
//	  scalar.{Int64, Int32, Uint64, Float64, Boolean, String, Binary};
//	  cdata.{CArrowArray, CArrowSchema, ExportArrowRecordBatch, ImportCRecordBatch,
//	  ReleaseCArrowArray, ReleaseCArrowSchema}                           github.com/apache/arrow-go/v18 v18.2.0 (go.mod:7)
package gpuplan

/*
#cgo CFLAGS: -I${SRCDIR}/../../../third_party/frostdb_amd/include
#cgo LDFLAGS: -L${SRCDIR}/../../../third_party/frostdb_amd -lfrostdb_amd
#include <stdlib.h>
#include "frostdb_amd.h"
extern int32_t fdbRegexMatch(void*, char*, int64_t, uint8_t*, int64_t);
// cgo exports take non-const pointers; this wrapper has fdb_regex_match_fn's exact (const-qualified) signature
static int32_t fdbRegexMatchC(void* user, const char* pat, int64_t pat_len, const uint8_t* val, int64_t val_len) {
	return fdbRegexMatch(user, (char*)pat, pat_len, (uint8_t*)val, val_len);
}
static fdb_regex_match_fn fdbRegexMatchFn(void) { return fdbRegexMatchC; }
*/
import "C"

import (
	"context"
	"errors"
	"fmt"
	"regexp"
	"sync"
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow"
	"github.com/apache/arrow-go/v18/arrow/array"
	"github.com/apache/arrow-go/v18/arrow/cdata"
	"github.com/apache/arrow-go/v18/arrow/memory"
	"github.com/apache/arrow-go/v18/arrow/scalar"

	"github.com/polarsignals/frostdb/query/logicalplan"
	"github.com/polarsignals/frostdb/query/physicalplan"
)

// FactoryOption tunes what Factory's operators do around the C ABI.
type FactoryOption func(*factoryConfig)

type factoryConfig struct {
	device         int
	accountResults bool
	deterministic  bool
}

// AccountResults makes every emitted record a copy built from the engine's memory.Allocator (FusedStage.Pool), so that
// memory.CheckedAllocator sees the operator's output like the Go operators' (logictest/logic_test.go:169-177). Without it the record's
// buffers stay the library's (pinned host memory released through the Arrow release callback): zero-copy, invisible to the allocator.
func AccountResults() FactoryOption { return func(c *factoryConfig) { c.accountResults = true } }

// Deterministic asks every operator for float64 sums that are bit-identical from run to run (fdb_plan_set_deterministic).
func Deterministic() FactoryOption { return func(c *factoryConfig) { c.deterministic = true } }

// Factory is what the user hands to physicalplan.WithOperatorFactory: Build calls the returned func once per chain for every
// [Filter] [Projection] Aggregation|Distinct stage. A stage this library does not cover is declined with ErrOperatorNotFused and Build
// plans the Go operators for it, so installing the factory never makes a query fail that worked before.
func Factory(device int, opts ...FactoryOption) physicalplan.OperatorFactory {
	cfg := factoryConfig{device: device}
	for _, o := range opts {
		o(&cfg)
	}
	return func(st physicalplan.FusedStage) (physicalplan.PhysicalPlan, error) {
		if st.Final && st.Aggregation != nil {
			// A single chain's lone HashAggregate is a FINAL stage reading raw rows (COUNT sums its input there, aggregate.go:965-969);
			// the reference never runs that way in practice (concurrencyHardcoded = GOMAXPROCS, physicalplan.go:22). Keep the Go operator.
			return nil, fmt.Errorf("gpuplan: single-chain plan: %w", physicalplan.ErrOperatorNotFused)
		}
		agg := st.Aggregation
		if agg == nil { // Filter → Distinction (distinct.go:21-170): group matchers without aggregations
			agg = &logicalplan.Aggregation{GroupExprs: st.Distinct}
		}
		if st.Ordered && len(agg.AggExprs) != 1 {
			return nil, fmt.Errorf("gpuplan: ordered aggregate with %d aggregations: %w", len(agg.AggExprs), physicalplan.ErrOperatorNotFused)
		}
		op, err := newOperator(cfg.device, st.Filter, st.Projections, agg, st.Ordered)
		if err != nil {
			// Whatever the library refuses (an expression shape, a type) the Go operators may still take: decline, keep the reason.
			return nil, fmt.Errorf("gpuplan: %v: %w", err, physicalplan.ErrOperatorNotFused)
		}
		if cfg.accountResults {
			op.pool = st.Pool
		}
		if cfg.deterministic {
			op.SetDeterministic(true)
		}
		return op, nil
	}
}

// Operator implements physicalplan.PhysicalPlan (physicalplan.go:24-30) for one chain.
type Operator struct {
	plan *C.fdb_plan
	next physicalplan.PhysicalPlan
	pool memory.Allocator // non-nil: emitted records are copied into it (AccountResults)
}

// cArena owns the C memory a descriptor is built from. The descriptor's arrays hold pointers (column names, literals), and cgo
// forbids passing Go memory that itself contains Go pointers ("Go pointer to unpinned Go pointer"): everything the descriptor
// points at is therefore C memory. fdb_plan_create copies what it keeps (names, literals, patterns), so the arena is freed as
// soon as the call has returned.
type cArena struct{ ptrs []unsafe.Pointer }

func (a *cArena) str(s string) *C.char {
	p := C.CString(s)
	a.ptrs = append(a.ptrs, unsafe.Pointer(p))
	return p
}
func (a *cArena) alloc(n int, size uintptr) unsafe.Pointer {
	if n == 0 {
		return nil
	}
	p := C.calloc(C.size_t(n), C.size_t(size))
	a.ptrs = append(a.ptrs, p)
	return p
}
func (a *cArena) free() {
	for _, p := range a.ptrs {
		C.free(p)
	}
	a.ptrs = nil
}

// New flattens the logical expressions into fdb_plan_desc. Op and AggFunc values are passed through unchanged:
// fdb_op == logicalplan.Op and fdb_agg_func == logicalplan.AggFunc numerically (logicalplan/expr.go:17-35, :718-729).
func New(device int, filter logicalplan.Expr, agg *logicalplan.Aggregation) (*Operator, error) {
	return newOperator(device, filter, nil, agg, false)
}

// NewOrdered is New for the chains of an OrderedAggregate (Build plans one when the scan is ordered by the group columns and there is ONE
// aggregation, physicalplan.go:433-449, :525-528): the operator emits its record sorted by the group columns and names the result after the
// aggregated column (ordered_aggregate.go:551-557), so OrderedSynchronizer and the final OrderedAggregate downstream see what they expect.
// Records need not arrive in key order for the result to be right — out-of-order input costs a device-side sort at Finish.
func NewOrdered(device int, filter logicalplan.Expr, agg *logicalplan.Aggregation) (*Operator, error) {
	if len(agg.AggExprs) != 1 {
		return nil, errors.New("gpuplan: an ordered aggregate takes exactly one aggregation") // ≙ NewOrderedAggregate's signature
	}
	return newOperator(device, filter, nil, agg, true)
}

func newOperator(device int, filter logicalplan.Expr, projections []logicalplan.Expr, agg *logicalplan.Aggregation, ordered bool) (*Operator, error) {
	var mem cArena
	defer mem.free()
	var nodes []C.fdb_expr // Go slice while it grows; its strings are C memory already, the array is copied to C below
	root := C.int32_t(-1)
	if filter != nil {
		r, err := flatten(filter, &nodes, &mem) // BinaryExpr{Column, Op, Literal} | And | Or  →  post-order array
		if err != nil {
			return nil, err // ≙ ErrUnsupportedBooleanExpression (filter.go:46)
		}
		root = C.int32_t(r)
	}
	cNodes := (*C.fdb_expr)(mem.alloc(len(nodes), unsafe.Sizeof(C.fdb_expr{})))
	if len(nodes) > 0 {
		copy(unsafe.Slice(cNodes, len(nodes)), nodes)
	}
	cAggs := (*C.fdb_aggregation)(mem.alloc(len(agg.AggExprs), unsafe.Sizeof(C.fdb_aggregation{})))
	aggs := unsafe.Slice(cAggs, len(agg.AggExprs))
	for i, a := range agg.AggExprs {
		aggs[i]._func = C.int32_t(a.Func)
		aggs[i].column = mem.str(a.Expr.Name())
		if _, dyn := a.Expr.(*logicalplan.DynamicColumn); dyn { // `max(foo)` over every foo.* column (aggregate.go:38-46)
			aggs[i].dynamic = 1
		}
	}
	cGroups := (*C.fdb_group_expr)(mem.alloc(len(agg.GroupExprs), unsafe.Sizeof(C.fdb_group_expr{})))
	groups := unsafe.Slice(cGroups, len(agg.GroupExprs))
	for i, g := range agg.GroupExprs {
		_, dyn := g.(*logicalplan.DynamicColumn)
		groups[i].name = mem.str(g.Name())
		if dyn {
			groups[i].dynamic = 1
		}
	}
	// The computed part of the Projection between filter and aggregate (project.go:73-399): arithmetic over columns and literals is
	// evaluated inside the scan kernel under the name the aggregate looks up; plain columns need nothing (only referenced columns are read).
	var projs []C.fdb_projection
	for _, e := range projections {
		switch e.(type) {
		case *logicalplan.Column, *logicalplan.DynamicColumn:
			continue
		}
		var pn []C.fdb_proj_node
		inner := e
		if al, ok := e.(*logicalplan.AliasExpr); ok {
			inner = al.Expr
		}
		r, err := flattenProjection(inner, &pn, &mem)
		if err != nil {
			return nil, err
		}
		cpn := (*C.fdb_proj_node)(mem.alloc(len(pn), unsafe.Sizeof(C.fdb_proj_node{})))
		copy(unsafe.Slice(cpn, len(pn)), pn)
		projs = append(projs, C.fdb_projection{name: mem.str(e.Name()), nodes: cpn, n_nodes: C.int32_t(len(pn)), root: C.int32_t(r)})
	}
	cProjs := (*C.fdb_projection)(mem.alloc(len(projs), unsafe.Sizeof(C.fdb_projection{})))
	if len(projs) > 0 {
		copy(unsafe.Slice(cProjs, len(projs)), projs)
	}
	// (desc itself is a Go value on this stack holding only C pointers: legal to pass by address)
	desc := C.fdb_plan_desc{filter: cNodes, n_filter: C.int32_t(len(nodes)), filter_root: root,
		aggs: cAggs, n_aggs: C.int32_t(len(aggs)), groups: cGroups, n_groups: C.int32_t(len(groups)),
		projections: cProjs, n_projections: C.int32_t(len(projs)),
		regex_match: C.fdbRegexMatchFn()} // `=~` / `!~` keep Go's regexp semantics, see below
	if ordered {
		desc.ordered = 1
	}
	op := &Operator{}
	if rc := C.fdb_plan_create(&desc, C.int(device), &op.plan); rc != C.FDB_OK {
		return nil, errors.New(C.GoString(C.fdb_last_error()))
	}
	return op, nil
}

// fdbRegexMatch is the library's regex engine: it is called once per DISTINCT value of the filtered column (per dictionary
// entry), never per row, so `labels.x =~ "(?i)foo.*"` selects exactly the rows the reference's RegExpFilter would
// (regexpfilter.go:84-166: unanchored regexp.Match on the value's bytes; patterns compiled once, filter.go:105-124).
// The descriptor gets the const-correct C wrapper from the preamble (fdbRegexMatchFn), not this export directly.
//
//export fdbRegexMatch
func fdbRegexMatch(_ unsafe.Pointer, pat *C.char, patLen C.int64_t, val *C.uint8_t, valLen C.int64_t) C.int32_t {
	re, err := compiled(C.GoStringN(pat, C.int(patLen))) // sync.Map[string]*regexp.Regexp
	if err != nil {
		return -1 // fdb_plan_create fails with FDB_ERR_INVALID, like regexp.Compile failing in physicalplan.Build
	}
	if re.Match(unsafe.Slice((*byte)(unsafe.Pointer(val)), int(valLen))) {
		return 1
	}
	return 0
}

// compiled caches regexp.Compile per pattern text for the life of the process (the reference compiles once per plan,
// filter.go:105-124; the library asks once per distinct dictionary value and pattern).
var regexCache sync.Map // string → *regexp.Regexp

func compiled(pattern string) (*regexp.Regexp, error) {
	if re, ok := regexCache.Load(pattern); ok {
		return re.(*regexp.Regexp), nil
	}
	re, err := regexp.Compile(pattern)
	if err != nil {
		return nil, err
	}
	regexCache.Store(pattern, re)
	return re, nil
}

// setLiteral maps the scalar of a LiteralExpr (logicalplan/expr.go:586-588) onto fdb_literal. Strings and binaries are copied into
// the arena (C memory); anything the filter of the reference does not compare with either (binaryscalarexpr.go:84-117) is an error
// and the caller keeps the Go operators for that plan.
func setLiteral(dst *C.fdb_literal, v scalar.Scalar, mem *cArena) error {
	if v == nil || v == scalar.ScalarNull || !v.IsValid() {
		dst._type = C.FDB_LIT_NULL
		return nil
	}
	bytesLit := func(kind C.int32_t, b []byte) {
		dst._type = kind
		dst.len = C.int64_t(len(b))
		if len(b) > 0 {
			p := mem.alloc(len(b), 1)
			copy(unsafe.Slice((*byte)(p), len(b)), b)
			dst.data = (*C.char)(p)
		}
	}
	switch x := v.(type) {
	case *scalar.Int64:
		dst._type, dst.i64 = C.FDB_LIT_INT64, C.int64_t(x.Value)
	case *scalar.Int32:
		dst._type, dst.i64 = C.FDB_LIT_INT64, C.int64_t(x.Value)
	case *scalar.Uint64:
		dst._type, dst.u64 = C.FDB_LIT_UINT64, C.uint64_t(x.Value)
	case *scalar.Float64:
		dst._type, dst.f64 = C.FDB_LIT_FLOAT64, C.double(x.Value)
	case *scalar.Boolean:
		dst._type = C.FDB_LIT_BOOL
		if x.Value {
			dst.i64 = 1
		}
	case *scalar.String:
		bytesLit(C.FDB_LIT_STRING, x.Data())
	case *scalar.Binary:
		bytesLit(C.FDB_LIT_BINARY, x.Data())
	default:
		return fmt.Errorf("gpuplan: literal of type %s is not supported", v.DataType())
	}
	return nil
}

// Callback ≙ PredicateFilter.Callback + HashAggregate.Callback. The record is only borrowed (table.go:808,:827):
// fdb_plan_push stages what it needs before returning.
func (o *Operator) Callback(ctx context.Context, r arrow.Record) error {
	if err := ctx.Err(); err != nil { // table.go:761-770 cancels a scan through its context: no crossing for a cancelled query
		return err
	}
	var arr cdata.CArrowArray
	var sch cdata.CArrowSchema
	cdata.ExportArrowRecordBatch(r, &arr, &sch)
	defer cdata.ReleaseCArrowArray(&arr)
	defer cdata.ReleaseCArrowSchema(&sch)
	if rc := C.fdb_plan_push(o.plan, (*C.struct_ArrowArray)(unsafe.Pointer(&arr)), (*C.struct_ArrowSchema)(unsafe.Pointer(&sch))); rc != C.FDB_OK {
		return errors.New(C.GoString(C.fdb_plan_last_error(o.plan)))
	}
	return nil
}

// CallbackMany pushes several records of a chain with ONE cgo crossing (fdb_plan_push_many): a crossing costs ≈ 0.1–0.2 µs of Go
// scheduler work plus the pinning of its arguments, which is comparable with what the library itself spends on a 1 024-row record
// (≈ 10 µs) only when records are tiny — but a scan that hands over a row group's records at once (table.go:783-860 collects them per
// granule) can batch them. On error `pushed` records were accepted; the error belongs to record `pushed`.
func (o *Operator) CallbackMany(ctx context.Context, rs []arrow.Record) (pushed int, err error) {
	if len(rs) == 0 {
		return 0, nil
	}
	if err := ctx.Err(); err != nil {
		return 0, err
	}
	// The ArrowArray / ArrowSchema structs AND the pointer tables live in C memory (C.calloc): cgo forbids storing Go pointers in C
	// memory, and its argument check does not look inside C allocations — structs in a Go slice whose addresses sit in a C array happen to
	// work today but break under a moving collector and under GODEBUG=cgocheck=2.
	n := C.size_t(len(rs))
	arrs := (*[1 << 24]cdata.CArrowArray)(C.calloc(n, C.size_t(unsafe.Sizeof(cdata.CArrowArray{}))))
	schs := (*[1 << 24]cdata.CArrowSchema)(C.calloc(n, C.size_t(unsafe.Sizeof(cdata.CArrowSchema{}))))
	pa := (*[1 << 28]*C.struct_ArrowArray)(C.malloc(n * C.size_t(unsafe.Sizeof(uintptr(0)))))
	ps := (*[1 << 28]*C.struct_ArrowSchema)(C.malloc(n * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arrs))
	defer C.free(unsafe.Pointer(schs))
	defer C.free(unsafe.Pointer(pa))
	defer C.free(unsafe.Pointer(ps))
	for i, r := range rs {
		cdata.ExportArrowRecordBatch(r, &arrs[i], &schs[i])
		pa[i] = (*C.struct_ArrowArray)(unsafe.Pointer(&arrs[i]))
		ps[i] = (*C.struct_ArrowSchema)(unsafe.Pointer(&schs[i]))
	}
	defer func() {  // (registered after the frees above, so it runs before them)
		for i := range rs {
			cdata.ReleaseCArrowArray(&arrs[i])
			cdata.ReleaseCArrowSchema(&schs[i])
		}
	}()
	var done C.int32_t
	if rc := C.fdb_plan_push_many(o.plan, &pa[0], &ps[0], C.int32_t(len(rs)), &done); rc != C.FDB_OK {
		return int(done), errors.New(C.GoString(C.fdb_plan_last_error(o.plan)))
	}
	return int(done), nil
}

// SetDeterministic asks for float64 sums that are bit-identical from run to run (fdb_plan_set_deterministic; before the first
// Callback). A scan that cannot be ordered then fails with the library's ErrUnsupported text instead of being answered approximately.
func (o *Operator) SetDeterministic(on bool) {
	v := C.int32_t(0)
	if on {
		v = 1
	}
	C.fdb_plan_set_deterministic(o.plan, v)
}

// Finish ≙ HashAggregate.Finish: emit the partial record(s) downstream, then propagate Finish (aggregate.go:527-541). The library emits
// one record per "aggregate" like finishAggregate does (aggregate.go:617-624): fdb_plan_finish the first, fdb_plan_finish_next the others —
// more than one only when a plain string / binary key column would pass 2 GiB in a single record (aggregate.go:426-468).
func (o *Operator) Finish(ctx context.Context) error {
	if err := ctx.Err(); err != nil {
		return err
	}
	var arr cdata.CArrowArray
	var sch cdata.CArrowSchema
	var n C.int64_t
	if rc := C.fdb_plan_finish(o.plan, (*C.struct_ArrowArray)(unsafe.Pointer(&arr)), (*C.struct_ArrowSchema)(unsafe.Pointer(&sch)), &n); rc != C.FDB_OK {
		return errors.New(C.GoString(C.fdb_plan_last_error(o.plan)))
	}
	for {
		if err := o.emit(ctx, &arr, &sch, int64(n)); err != nil {
			return err
		}
		var emitted C.int32_t
		if rc := C.fdb_plan_finish_next(o.plan, (*C.struct_ArrowArray)(unsafe.Pointer(&arr)), (*C.struct_ArrowSchema)(unsafe.Pointer(&sch)), &n, &emitted); rc != C.FDB_OK {
			return errors.New(C.GoString(C.fdb_plan_last_error(o.plan)))
		}
		if emitted == 0 {
			break
		}
	}
	return o.next.Finish(ctx)
}

// emit imports one record of the library (taking ownership of the C structs) and hands it to the next operator.
func (o *Operator) emit(ctx context.Context, arr *cdata.CArrowArray, sch *cdata.CArrowSchema, n int64) error {
	rec, err := cdata.ImportCRecordBatch(arr, sch) // takes ownership; release() frees the C++ holder
	if err != nil {
		return err
	}
	defer rec.Release()
	if n == 0 { // finishAggregate skips empty aggregates (aggregate.go:547-549)
		return nil
	}
	out := rec
	if o.pool != nil {
		if out, err = copyToPool(rec, o.pool); err != nil {
			return err
		}
		defer out.Release()
	}
	if err := ctx.Err(); err != nil {
		return err
	}
	return o.next.Callback(ctx, out)
}

// copyToPool rebuilds a record from the engine's allocator: array.Concatenate of ONE array allocates its buffers from `pool` and copies
// (the call filter() itself makes per column, filter.go:314), dictionaries included.
func copyToPool(rec arrow.Record, pool memory.Allocator) (arrow.Record, error) {
	cols := make([]arrow.Array, 0, int(rec.NumCols()))
	defer func() {
		for _, c := range cols {
			c.Release()
		}
	}()
	for i := 0; i < int(rec.NumCols()); i++ {
		c, err := array.Concatenate([]arrow.Array{rec.Column(i)}, pool)
		if err != nil {
			return nil, err
		}
		cols = append(cols, c)
	}
	return array.NewRecord(rec.Schema(), cols, rec.NumRows()), nil // NewRecord retains the columns
}

func (o *Operator) SetNext(next physicalplan.PhysicalPlan) { o.next = next }
func (o *Operator) Draw() *physicalplan.Diagram {
	var child *physicalplan.Diagram
	if o.next != nil {
		child = o.next.Draw()
	}
	return &physicalplan.Diagram{Details: C.GoString(C.fdb_plan_draw(o.plan)), Child: child}
}
func (o *Operator) Close() {
	if o.plan != nil {
		C.fdb_plan_close(o.plan)
		o.plan = nil
	}
	if o.next != nil {
		o.next.Close()
	}
}

// flatten turns a boolean logicalplan.Expr into the post-order fdb_expr array the descriptor carries: leaves are
// BinaryExpr{Column, Op, Literal} (filter.go:79-103), branches And / Or (filter.go:129-160). Anything else is
// ErrUnsupportedBooleanExpression and the caller keeps the Go operators for that plan.
func flatten(e logicalplan.Expr, out *[]C.fdb_expr, mem *cArena) (int, error) {
	b, ok := e.(*logicalplan.BinaryExpr)
	if !ok {
		return -1, physicalplan.ErrUnsupportedBooleanExpression
	}
	if b.Op == logicalplan.OpAnd || b.Op == logicalplan.OpOr {
		l, err := flatten(b.Left, out, mem)
		if err != nil {
			return -1, err
		}
		r, err := flatten(b.Right, out, mem)
		if err != nil {
			return -1, err
		}
		*out = append(*out, C.fdb_expr{op: C.int32_t(b.Op), left: C.int32_t(l), right: C.int32_t(r)})
		return len(*out) - 1, nil
	}
	col, ok := b.Left.(*logicalplan.Column)
	if !ok {
		return -1, errors.New("left side of binary expression must be a column") // filter.go:91-93
	}
	lit, ok := b.Right.(*logicalplan.LiteralExpr)
	if !ok {
		return -1, physicalplan.ErrUnsupportedBooleanExpression
	}
	n := C.fdb_expr{op: C.int32_t(b.Op), left: -1, right: -1, column: mem.str(col.ColumnName)}
	if err := setLiteral(&n.literal, lit.Value, mem); err != nil { // scalar.Int64 → FDB_LIT_INT64, scalar.String → FDB_LIT_STRING (data, len), scalar.Null → FDB_LIT_NULL, …
		return -1, err
	}
	*out = append(*out, n)
	return len(*out) - 1, nil
}

// flattenProjection turns arithmetic over columns and literals (binaryExprProjection, project.go:73-161) and boolean expressions
// (boolExprProjection, project.go:397-470: comparisons, And / Or) into the post-order fdb_proj_node array: kind 0 column, 1 literal,
// 2 binary + - * /, 3 comparison / AND / OR. A string / binary / NULL literal may only be the right side of a comparison with a column
// (the library evaluates that comparison the way a filter leaf is evaluated). Anything else declines the stage.
func flattenProjection(e logicalplan.Expr, out *[]C.fdb_proj_node, mem *cArena) (int, error) {
	return flattenProjectionNode(e, out, mem, false)
}

func flattenProjectionNode(e logicalplan.Expr, out *[]C.fdb_proj_node, mem *cArena, rightOfCompare bool) (int, error) {
	switch x := e.(type) {
	case *logicalplan.Column:
		*out = append(*out, C.fdb_proj_node{kind: 0, left: -1, right: -1, column: mem.str(x.ColumnName)})
	case *logicalplan.LiteralExpr:
		n := C.fdb_proj_node{kind: 1, left: -1, right: -1}
		if err := setLiteral(&n.literal, x.Value, mem); err != nil {
			return -1, err
		}
		numeric := n.literal._type == C.FDB_LIT_INT64 || n.literal._type == C.FDB_LIT_FLOAT64 || n.literal._type == C.FDB_LIT_UINT64
		if !numeric && !rightOfCompare {
			return -1, fmt.Errorf("gpuplan: projection literal %s is not int64 / uint64 / float64", x.Value)
		}
		*out = append(*out, n)
	case *logicalplan.BinaryExpr:
		kind := C.int32_t(2)
		switch x.Op {
		case logicalplan.OpAdd, logicalplan.OpSub, logicalplan.OpMul, logicalplan.OpDiv:
		case logicalplan.OpEq, logicalplan.OpNotEq, logicalplan.OpLt, logicalplan.OpLtEq, logicalplan.OpGt, logicalplan.OpGtEq,
			logicalplan.OpAnd, logicalplan.OpOr:
			kind = 3
		default:
			return -1, fmt.Errorf("gpuplan: projection operator %s is not fused", x.Op.String())
		}
		l, err := flattenProjectionNode(x.Left, out, mem, false)
		if err != nil {
			return -1, err
		}
		_, leftIsColumn := x.Left.(*logicalplan.Column)
		compare := kind == 3 && x.Op != logicalplan.OpAnd && x.Op != logicalplan.OpOr
		r, err := flattenProjectionNode(x.Right, out, mem, compare && leftIsColumn)
		if err != nil {
			return -1, err
		}
		*out = append(*out, C.fdb_proj_node{kind: kind, op: C.int32_t(x.Op), left: C.int32_t(l), right: C.int32_t(r)})
	default:
		return -1, fmt.Errorf("gpuplan: projection %s is not fused", e.String())
	}
	return len(*out) - 1, nil
}

// ---- more than one GPU in this process (the reference's N chains live in ONE process, physicalplan.go:22, :337-347) ------------

// Comms is one RCCL communicator over the node's GPUs: ncclCommInitAll behind fdb_comm_init_all; comms[d] belongs to device d.
type Comms []*C.fdb_comm

func NewComms(devices []int) (Comms, error) {
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	out := make([]*C.fdb_comm, len(devices))
	if rc := C.fdb_comm_init_all(&devs[0], C.int32_t(len(devs)), &out[0]); rc != C.FDB_OK {
		return nil, errors.New(C.GoString(C.fdb_last_error()))
	}
	return out, nil
}

// MergeAcrossDevices is ≙ Synchronizer + HashAggregate(final=true) for chains on different GPUs: every chain's goroutine calls it
// with its own endpoint (the calls are collective). With equal table layouts — parts of one table — the per-GPU tables are
// all-reduced in place over xGMI and the chain on comms[0] emits the final record; otherwise the tables are hash-partitioned,
// exchanged, merged, and EVERY chain emits its shard of the groups (OutputPlan's callback takes several records).
func (o *Operator) MergeAcrossDevices(ctx context.Context, comm *C.fdb_comm) error {
	// (no ctx.Err() shortcut here: the merge is collective — a rank that stayed out would leave its peers waiting inside RCCL;
	// a cancelled query still takes part and drops its result afterwards)
	var aligned C.int32_t
	if rc := C.fdb_plan_allreduce(o.plan, comm, &aligned); rc != C.FDB_OK {
		return errors.New(C.GoString(C.fdb_plan_last_error(o.plan)))
	}
	if aligned == 1 {
		if C.fdb_comm_rank(comm) != 0 {
			return o.next.Finish(ctx) // the merged table is emitted by rank 0's chain only
		}
		return o.Finish(ctx)
	}
	var shard *C.fdb_plan
	if rc := C.fdb_plan_exchange(o.plan, comm, &shard); rc != C.FDB_OK {
		return errors.New(C.GoString(C.fdb_plan_last_error(o.plan)))
	}
	C.fdb_plan_close(o.plan)
	o.plan = shard
	return o.Finish(ctx)
}

// ---- parts that never become Arrow on the host (pqarrow/arrow.go:711-823 is today's producer) ---------------------------------

// ResidentRowGroup decodes one Parquet row group on the device (fdb_batch_from_parquet) from the column chunks' bytes as
// parquet-go's file metadata locates them; the batch can be pushed to any number of queries (fdb_plan_push_batches) and stays
// in HBM until released. Each chunk names its pages' codec (fdb_parquet_chunk.codec = format.CompressionCodec of the column's
// metadata: UNCOMPRESSED, SNAPPY, GZIP, ZSTD, LZ4_RAW and BROTLI are inflated by the library). Row groups outside the covered
// layouts (repeated / nested columns, physical types other than BOOLEAN, INT64, DOUBLE, BYTE_ARRAY) return FDB_ERR_UNSUPPORTED:
// convert those with pqarrow as before and use fdb_batch_import. The chunk descriptors hold pointers (column name, page bytes):
// like the plan descriptor they must point at C memory (C.CString / C.CBytes, or an mmap of the file) — cgo refuses Go memory
// that itself contains Go pointers; the library copies what it keeps before it returns.
func ResidentRowGroup(device int, chunks []C.fdb_parquet_chunk, rows int64) (*C.fdb_batch, error) {
	var b *C.fdb_batch
	if rc := C.fdb_batch_from_parquet(&chunks[0], C.int32_t(len(chunks)), C.int64_t(rows), C.int(device), &b); rc != C.FDB_OK {
		return nil, errors.New(C.GoString(C.fdb_last_error()))
	}
	return b, nil
}

// ResidentRowGroups decodes several row groups with ONE call (fdb_batches_from_parquet): one copy queue for all of them, their page
// headers, inflating and run-header walks side by side on the library's host threads, a row group's decode kernels launched while later
// ones are still being parsed (20 M rows in 4 row groups: 7.4 ms against 8.6–10 for four calls from two goroutines; with SNAPPY pages
// and DELTA timestamps 5.7–6.1 against 6.6–9.6). The chunks or inflated images of all the call's row groups are in device memory at
// once: bound a call by bytes. `groups` must be C memory like the chunk descriptors (it holds pointers to them). On error no batch
// is returned.
func ResidentRowGroups(device int, groups []C.fdb_parquet_row_group) ([]*C.fdb_batch, error) {
	if len(groups) == 0 {
		return nil, nil
	}
	out := make([]*C.fdb_batch, len(groups))
	if rc := C.fdb_batches_from_parquet(&groups[0], C.int32_t(len(groups)), C.int(device), &out[0]); rc != C.FDB_OK {
		return nil, errors.New(C.GoString(C.fdb_last_error()))
	}
	return out, nil
}
