// fdb_jit.h — plan-specialised scan kernels (fdb_jit.cpp).
#pragma once
#include <cstdlib>

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "fdb_kernels.h"

namespace fdb {

struct JitSlot { bool has_values = false; int has_validity = 0; };  // has_validity: 0 no record has a bitmap, 1 every record, 2 some (checked per record)
struct JitLeaf { int kind = 0, slot = -1, wide = 0, op = 0; bool lut_in_lds = false; };
struct JitGroup { int slot = -1; bool lut_in_lds = false; };
struct JitAgg { int func = 0, type = 0, slot = -1; int expr = 0; };  // expr: 1 + root node of a computed input, 0 = stored column
struct JitExprNode { int kind = 0, op = 0, left = -1, right = -1, slot = -1, type = 0; };  // literal values stay run-time arguments

// Everything that changes the generated CODE. Pointers, literals, truth-table bits, LUT offsets, strides and row counts
// are run-time arguments and deliberately absent, so that queries of the same shape share one compiled kernel.
struct JitShape {
  int block = 512;
  bool lds_acc = true, need_count = false, two_phase = false;
  int n_c4 = 0, n_c8 = 0, n_l4 = 0, n_l8 = 0;
  JitSlot c4[FDB_ARG_C4], c8[FDB_ARG_C8], l4[FDB_ARG_L4], l8[FDB_ARG_L8];
  std::vector<JitLeaf> leaves;
  std::vector<uint8_t> code;  // postfix program over the leaves
  std::vector<JitGroup> gcols;
  std::vector<JitAgg> aggs;
  std::vector<JitExprNode> exprs;  // pre-aggregate arithmetic (FdbScanArgs.expr)
  // Tiny tables (no group-by, or a handful of slots): every lane keeps the WHOLE table in registers across all its tiles and
  // the table is reduced across the wave once, at the end — per-row LDS atomics on ≤ 8 addresses serialise almost completely
  // (`sum(value)` without group-by ran at 0.7 TB/s with them). 0 = off, else the exact slot count the kernel is built for.
  int reg_slots = 0;
  // Tables too big for LDS (lds_acc = false): a per-workgroup combining cache in LDS (direct-mapped by a hash of the slot, first
  // come first served, no eviction) absorbs the rows of the keys it holds and is flushed with one global atomic per entry and
  // aggregate at the end; rows of other keys update the global table directly. Hot keys — which serialise on their cache
  // line in L2 when every row is a global atomic (sum by (path, instance): 11 ms per 50 M rows) — are first to get a place.
  bool cache = false;
  // Reproducible float sums (fdb_plan_set_deterministic): every WAVE accumulates into an LDS table of its own — rows reach a table in
  // program order, lane order inside an instruction — and the flush adds the waves' tables up in wave order; with shared tables the
  // order in which the waves' LDS atomics interleave differs from run to run, and float64 addition is not associative.
  bool wave_tables = false;
  // LDS tables: when the slots of a wave's selected rows agree (sorted input) the updates go to a wave-uniform address, so that the compiler
  // folds them across the lanes (fdb_jit.cpp, "Sorted input")
  bool uniform_fold = true;  // ($FDB_NO_UNIFORM_FOLD, A/B aid: a plan reads it at create, Plan::Knobs)
  // fdb_select_kernel only (not part of key()): early slots whose values the kernel compacts itself, bit i = slot i of c4 / c8
  int fuse4 = 0, fuse8 = 0;
  std::string key(bool with_validity = true) const;
};

// Shape of a high-cardinality (hash table) scan: fdb_hash_kernel.
struct JitHashCol { int kind = 0; bool has_validity = false, lut_in_lds = false; int expr_root = -1; bool lut_identity = false; };  // kind 2: computed int64 key; lut_identity: key id = dictionary index + 1 (FdbHashCol.lut == nullptr)
struct JitHashShape {
  std::vector<JitHashCol> cols;
  std::vector<JitLeaf> leaves;  // slot / wide describe the leaf's own column (wide = 8-byte values)
  std::vector<bool> leaf_validity;
  std::vector<uint8_t> code;
  std::vector<JitAgg> aggs;
  std::vector<bool> agg_validity;
  std::vector<JitExprNode> exprs;  // column nodes read base.l8[slot]
  int n_expr_cols = 0;
  bool need_count = true;  // some aggregation is COUNT: otherwise the per-entry row count is never read (occupancy = fingerprint ≠ 0) and its atomic is skipped
  int ablate = 0;  // tuning aid (tools/cfg5_ablate.py): 1 = stream + fingerprint only, 2 = no count / aggregate atomics
  // Table-free OrderedAggregate (FdbHashArgs.runs): no probe, no insert — every wave emits the runs of equal keys among its 256 rows
  // (key ids, row count, folded aggregate) and notes them in the launch's directory. Exactly one aggregation. 1: narrow records —
  // dictionary columns with ≤ 255 values, at most FDB_RUN_TUPLE_BYTES of them, ids packed one byte each while the fingerprint is
  // computed; 2: wide records — the table's own key tuple (any cardinality, int64 / computed keys, absent columns), written by the lanes
  // that end a run from re-loaded columns.
  int runs = 0;
  std::string key() const;
};
// `hcols` = host copy of args.hcols.
JitHashShape jit_hash_shape(const FdbHashArgs& args, const FdbHashCol* hcols);
std::string jit_hash_source(const JitHashShape& shape);
hipFunction_t jit_hash_get(const JitHashShape& shape);
// 256-thread workgroups, 1 024-row tiles.
hipError_t jit_hash_launch(hipFunction_t fn, const FdbHashArgs& args, int grid, size_t lds_bytes, hipStream_t stream);

// The shape of one record's slot-assigned argument block.
JitShape jit_shape(const FdbScanArgs& a, bool two_phase, int block);
// Folds `other` into `into` when the two differ at most in which slots carry validity bitmaps; false otherwise.
bool jit_shape_merge(JitShape* into, const JitShape& other);
// The same for the shape of argument block `b`, decided on the blocks themselves (`a`: the block `into` was made from): no JitShape, no key strings.
bool jit_shape_merge_args(JitShape* into, const FdbScanArgs& a, const FdbScanArgs& b, bool two_phase);
// Workgroups of `block` threads with `lds_bytes` of dynamic LDS that fit one CU (≥ 1).
int jit_blocks_per_cu(hipFunction_t fn, int block, size_t lds_bytes);

// Kernels compiled with hiprtc by this process, the time that took, and code objects loaded from the disk cache instead.
void jit_stats(int64_t* n_compiled, double* compile_ms, int64_t* n_disk_loads);

// filter() (Plan::filter_batches): the selection-bitmap kernel of a predicate shape — 256-thread workgroups, one wave per tile of
// 2 048 rows (FDB_COMPACT_TILE): 64 mask words and one count per tile, plain stores. The argument blocks count tile_begin /
// tile_end in units of FOUR tiles (a workgroup's share, all of one record) and carry the record's first tile in out_tile_base.
std::string jit_flags_source(const JitShape& shape);
hipFunction_t jit_flags_get(const JitShape& shape);
hipError_t jit_flags_launch(hipFunction_t fn, const FdbScanArgs* d_parts, int n_parts, int64_t total_super_tiles, const FdbScanArgs& common, int grid, size_t lds_bytes,
                            uint32_t* masks, uint32_t* tile_counts, hipStream_t stream);

// filter() in one pass over the filter columns (fdb_kernels.h, FdbSelectArgs): fdb_flags_kernel's geometry and bitmap, plus the tile
// offsets (decoupled look-back inside each record, tiles handed out by a ticket counter) and the compacted values of the fused slots
// (shape.fuse4 / fuse8), staged in LDS — jit_select_stage_bytes(shape) per wave (jit_select_block() / 64 of them) behind `stage_off`.
std::string jit_select_source(const JitShape& shape);
hipFunction_t jit_select_kernel_get(const JitShape& shape);
size_t jit_select_stage_bytes(const JitShape& shape);
int jit_select_block();  // threads per workgroup (8 waves, half a tile each)
hipError_t jit_select_launch(hipFunction_t fn, const FdbScanArgs* d_parts, int n_parts, int64_t total_super_tiles, const FdbScanArgs& common, int grid, size_t lds_bytes,
                             uint32_t* masks, uint32_t* offsets, const FdbSelectArgs& sel, hipStream_t stream);

std::string jit_source(const JitShape& shape);
// The compiled kernel for `shape` (cached in the process and on disk), or nullptr if specialisation is unavailable.
// While one of these with may_defer = true is alive on a thread, a jit_*get of a kernel that is not built yet may return nullptr at once
// and leave the build to a background thread ($FDB_JIT_ASYNC=1 only): set by callers whose launch the interpreting kernels can serve.
class JitDeferScope {
 public:
  explicit JitDeferScope(bool may_defer);
  ~JitDeferScope();
  JitDeferScope(const JitDeferScope&) = delete;
  JitDeferScope& operator=(const JitDeferScope&) = delete;
 private:
  bool prev_;
};
hipFunction_t jit_get(const JitShape& shape);
// jit_get plus the launch geometry: workgroup size and workgroups per CU such that ≈64 KB of loads are in flight per CU
// (`row_bytes` = bytes of column data the scan reads per row).
hipFunction_t jit_select(JitShape shape, size_t lds_bytes, int row_bytes, int* block_out, int* blocks_per_cu_out);
hipError_t jit_launch(hipFunction_t fn, const FdbScanArgs* d_parts, int n_parts, int64_t total_tiles, const FdbScanArgs& common, int grid, int block,
                      size_t lds_bytes, hipStream_t stream);

}  // namespace fdb
