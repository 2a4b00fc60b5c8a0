// fdb_kernels.h — argument blocks and launch wrappers of the gfx950 scan kernels (host/device shared).
//
// The device sees no strings: every dictionary column is reduced on the host, once per batch and per
// dictionary ENTRY, to small look-up tables (predicate LUT: entry → 0/1; group LUT: entry → per-column
// key id, 0 = NULL). That replaces the reference's per-ROW bytes.Equal / metro.Hash64
// (binaryscalarexpr.go:214-229, dynparquet/hashed.go:201-216).
#pragma once

#ifndef FDB_DEVICE_ONLY  // (the run-time compiled plan kernels include this header through hiprtc, without host headers)
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#else
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t;
#endif

#define FDB_MAX_LEAVES 12
#define FDB_MAX_CODE 32
#define FDB_MAX_DENSE_GCOLS 8
#define FDB_MAX_HASH_GCOLS 64
#define FDB_MAX_AGGS 8
#define FDB_MAX_C4 4            // column slots of the load-hoisting kernel: 4-byte (dictionary index) columns
#define FDB_MAX_C8 2            //                                           8-byte (int64/uint64/float64) columns
#define FDB_MAX_L4 2            // late (post-filter) slots of the two-phase kernel
#define FDB_MAX_L8 3
// Sizes of the slot arrays in the argument block: what the run-time specialised kernels can take (they have no
// register-resident plan); the FDB_MAX_* values above stay the limits of the interpreting slot kernel's template instances.
#define FDB_ARG_C4 6
#define FDB_ARG_C8 4
#define FDB_ARG_L4 6
#define FDB_ARG_L8 4
#define FDB_BLOCK 1024          // 16 waves share one LDS partial table
#define FDB_LDS_BUDGET 65536    // bytes of LDS per workgroup (2 workgroups/CU of the 160 KiB)
#define FDB_NO_LDS 0xFFFFFFFFu

enum FdbLeafKind : int32_t {
  FDB_LEAF_CONST = 0,      // op = 0/1: no row / every row (missing-column rules, binaryscalarexpr.go:47-73)
  FDB_LEAF_DICT_LUT = 1,   // values = uint32 indices; lut[idx] ∈ {0,1}; entry lut_len-1 is the answer for a NULL row
  FDB_LEAF_DICT_BITS = 7,  // same truth table packed in `lit` (≤ 63 dictionary entries + the NULL row): no LDS access
  FDB_LEAF_CMP_I64 = 2,    // values = int64;  lit = int64
  FDB_LEAF_CMP_U64 = 3,    // values = uint64; lit = uint64
  FDB_LEAF_CMP_F64 = 4,    // values = double; lit = double bits
  FDB_LEAF_CMP_I64_F64 = 5,// values = int64 compared as double against a double literal
  FDB_LEAF_VALIDITY = 6    // op = 0: IS NULL, 1: IS NOT NULL (binaryscalarexpr.go:165-172, :205-212)
};

enum FdbCode : uint8_t { FDB_CODE_AND = 0x80, FDB_CODE_OR = 0x81 };  // < 0x80: push leaf i

struct FdbLeaf {
  const void* values;
  const uint8_t* validity;  // Arrow validity bitmap at bit offset 0, or nullptr (no nulls)
  const uint8_t* lut;       // FDB_LEAF_DICT_LUT: global copy of the LUT
  int64_t lit;
  int32_t kind;
  int32_t op;               // fdb_op for compares
  uint32_t lut_len;
  uint32_t lut_lds;         // byte offset of the LUT's LDS copy, or FDB_NO_LDS (too big: gather from L2)
  int32_t slot;             // column slot (c4 for dictionary columns, c8 for 8-byte columns); -1: not slotted
  int32_t wide;             // 1: the slot is in c8
};

struct FdbGroupCol {
  const uint32_t* idx;
  const uint8_t* validity;
  const uint32_t* lut;      // dictionary entry → key id (≥ 1); NULL rows take id 0
  uint32_t lut_len;
  uint32_t lut_lds;
  uint32_t stride;          // dense path: mixed-radix multiplier of this column
  int32_t slot;             // c4 slot
};

enum FdbAggType : int32_t { FDB_T_NONE = 0, FDB_T_I64 = 1, FDB_T_F64 = 2, FDB_T_BOOL = 3 /* expression nodes only: a comparison's 0 / 1 */,
                            FDB_T_U64 = 4 /* expression nodes only: uint64 arithmetic (project.go:138-150); no aggregation takes it (aggregate.go:736) */ };

struct FdbAgg {
  const void* values;       // nullptr for COUNT (row count only) and for computed inputs
  const uint8_t* validity;
  unsigned long long* acc;  // global accumulator array [n_slots] (int64 bits / double bits / ordered-f64 keys)
  int32_t func;             // fdb_agg_func (SUM/MIN/MAX/COUNT)
  int32_t type;             // FdbAggType of the input column / expression
  int32_t slot;             // c8 slot
  int32_t expr;             // 1 + root node (FdbScanArgs.expr) of a computed input (pre-aggregate Projection); 0: stored column
  unsigned long long null_value;  // what a NULL row contributes: 0 (the builder's zeroed slot: SUM/MIN/MAX, aggregate.go:784-935), or the
                                  // value that makes a NULL decisive / neutral for the composite reducers built on MIN and MAX
                                  // (UNIQUE: INT64_MIN for its MIN half, INT64_MAX for its MAX half; AND: 1)
};

// One node of a pre-aggregate arithmetic expression (physicalplan/project.go:73-161), evaluated per row inside the run-time
// specialised kernels only (the interpreting kernels reject such plans). Structure and types are part of the kernel's
// shape; literal VALUES are read from here at run time, so `timestamp / 1000 * 1000` and `timestamp / 5000 * 5000` share a kernel.
#define FDB_MAX_EXPR_NODES 40
struct FdbExprNode {
  int64_t lit;              // literal: int64 value / float64 bits
  int32_t kind;             // 0 column, 1 literal, 2 binary arithmetic, 3 comparison (boolExprProjection: NULL operand ⇒ false),
                            // 4 convert int64 → float64 (left), 5 isnull (left = a column node), 6 if (cond = node `op`) left else right
  int32_t op;               // binary: fdb_op (ADD 11, SUB 12, MUL 13, DIV 14); comparison: EQ 1 … GT_EQ 6; if: index of the condition node
  int32_t left, right;      // binary: child node indices
  int32_t slot;             // column: 8-byte slot in the pool the aggregates use (dense single-phase: c8, two-phase: l8; hash scan: l8)
  int32_t type;             // FdbAggType of the node's value
};

// One distinct referenced column of the batch. The slot kernel issues the loads of ALL slots of a tile before
// it consumes any of them, so a wave keeps every referenced column in flight at once (memory-level parallelism).
struct FdbColSlot {
  const void* values;       // nullptr: only the validity bitmap is needed (IS [NOT] NULL leaves)
  const uint8_t* validity;  // nullptr: no NULLs
};

struct FdbScanArgs {
  int64_t n_rows;
  int64_t tile_begin;       // multi-record launch: this record owns global tiles [tile_begin, tile_end)
  int64_t tile_end;
  unsigned long long* cnt;  // global selected-row count per slot (occupancy + COUNT)
  uint32_t n_slots;
  int32_t n_leaves;
  int32_t n_code;
  int32_t n_gcols;
  int32_t n_aggs;
  int32_t lds_acc;          // 1: stage partial aggregates in LDS, flush once per workgroup; 0: global atomics per row
  uint32_t lds_lut_bytes;   // bytes of LUT copies at the start of dynamic LDS
  int32_t need_count;       // 1: some aggregation is COUNT (exact per-slot row counts needed)
  int32_t n_c4;             // > 0 or n_c8 > 0: slots are assigned (otherwise only the sequential kernel can run)
  int32_t n_c8;
  // Two-phase plans (more columns than the single-phase register class holds): c4/c8 are the columns the FILTER
  // reads (loaded first); l4/l8 the columns only the group-by and the aggregates read — late materialisation:
  // loaded after the filter, and not at all for a lane group whose rows were all filtered out.
  int32_t n_l4;
  int32_t n_l8;
  FdbColSlot l4[FDB_ARG_L4];
  FdbColSlot l8[FDB_ARG_L8];
  FdbColSlot c4[FDB_ARG_C4];
  FdbColSlot c8[FDB_ARG_C8];
  unsigned long long* partials;  // LDS mode: per-workgroup partial tables [grid][1 + n_aggs][n_slots], written with plain
                                 // coalesced stores and folded by fdb_launch_reduce_partials (nullptr: flush with atomics)
  int32_t ablate;           // tuning aid (bench --ablate): 1 skip occupancy, 2 skip aggregate atomics, 4 skip group LUTs, 8 skip filter
  int32_t lut_class;        // records with the same class carry byte-identical LUT sets at the same LDS offsets
  uint8_t code[FDB_MAX_CODE];
  uint32_t ops_after[FDB_MAX_LEAVES];  // slot kernel: the AND/OR ops that follow leaf l in postfix order: [3:0] count, then 2 bits
                                       // per op (1 AND, 2 OR); 0xFFFFFFFF if the program does not fit this form
  FdbLeaf leaves[FDB_MAX_LEAVES];
  FdbGroupCol gcols[FDB_MAX_DENSE_GCOLS];
  FdbAgg aggs[FDB_MAX_AGGS];
  int32_t n_expr;           // nodes of computed aggregate inputs / group keys (0: none)
  int32_t cache_slots;      // lds_acc == 0 (table too big for LDS), specialised kernel only: entries (a power of two) of the workgroup's
                            // LDS combining cache [tag u32 | count u32 | acc u64 × n_aggs] placed after the LUT copies; 0: none
  FdbExprNode expr[FDB_MAX_EXPR_NODES];
  int64_t out_tile_base;    // fdb_flags_kernel only: the record's first tile in the launch-wide mask / count arrays
  // fdb_plan_kernel only, read from the by-value copy: a table that no kernel has touched yet gets its identity fill from the FIRST
  // scan launch itself (the partial tables are folded into it by the next kernel on the stream) instead of a launch of its own in
  // front of the scan — 6 µs + a dispatch gap of a 125 M-row shard's 350 µs step. nullptr: nothing to fill.
  unsigned long long* fill_state;
  uint32_t fill_words;      // slots allocated × arrays
  uint32_t fill_alloc;      // slots allocated per array: word i belongs to array i / fill_alloc
  unsigned long long fill_idents[1 + FDB_MAX_AGGS];
};

// ---- high-cardinality path: global open-addressing hash table -------------------------------------------------
// Key tuple layout (32-bit words): [valid mask lo | valid mask hi | column words …]; bit g of the mask = group column
// g (plan-level index, stable for the life of the plan) is non-NULL in this group. A dictionary column owns one word
// (its key id), an int64 column two (low, high). The fingerprint is a SUM of per-column mixes of the non-NULL columns
// only, so it does not depend on column order and a column that shows up later (all earlier groups NULL in it)
// leaves earlier fingerprints unchanged.
struct FdbHashCol {
  const void* values;       // uint32 indices (dictionary) or int64 values
  const uint8_t* validity;
  const uint32_t* lut;      // dictionary entry → key id (scan) / source key id → destination key id (merge; nullptr = identity)
  uint32_t lut_len;
  uint32_t lut_lds;         // byte offset of the LDS copy or FDB_NO_LDS
  int32_t kind;             // 0 dictionary, 1 int64, 2 computed int64 (src_word = root node of its expression; specialised scan only)
  int32_t word;             // first word of this column inside the key tuple
  int32_t gi;               // plan-level group column index (fingerprint salt, valid-mask bit)
  int32_t src_word;         // merge: first word of this column inside the INCOMING key tuple (-1: absent ⇒ NULL)
  unsigned long long k1;    // fdb_fp_k1(gi), fdb_fp_k2(gi): this column's multipliers of the multilinear fingerprint,
  unsigned long long k2;    // computed once on the host (recomputing them per row cost ≈1 000 scalar instructions per wave tile)
};

static inline unsigned long long fdb_fmix64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}
static inline unsigned long long fdb_fp_k1(int gi) { return fdb_fmix64(0x9E3779B97F4A7C15ULL * (unsigned long long)(gi + 1)) | 1ull; }
static inline unsigned long long fdb_fp_k2(int gi) {
  return fdb_fmix64(0xD6E8FEB86659FD93ULL * (unsigned long long)(gi + 1) + 0x632BE59BD9B4E019ULL) | 1ull;
}

// Table entry = entry_words × 8 bytes: [fingerprint lo (0 = empty) | fingerprint hi | selected-row count | acc 0 … ]
// padded to a multiple of 32 bytes so that probe, count and accumulators of a group share one 64-byte sector.
// Run store of the table-free OrderedAggregate (Plan::push_runs, fdb_hash_kernel in runs mode): a scan whose rows arrive ordered
// by the group columns needs no table — a group is one RUN of consecutive rows (ordered_aggregate.go:163-409: group ranges, the
// last group carried into the next record). Every wave emits the runs of its 256 rows — packed key ids (one byte per group
// column, plan order; the mode is only chosen while every column has ≤ 255 distinct values), row count and the folded aggregate
// of each — into chunks of FDB_RUN_CHUNK runs handed out by one global atomic per chunk, and notes (first run, number of runs) in
// the directory entry of its (tile, wave). Runs cut by a wave / record boundary and keys that come back later are merged at Finish.
#define FDB_RUN_CHUNK 4096
#define FDB_RUN_TUPLE_BYTES 32
#define FDB_RUN_BYTES 48                              // one run: [key ids: 32 bytes | rows of the run: 8 | its aggregate, the accumulator's own representation: 8]
#define FDB_RUN_STAGE 256                             // runs a wave can hold back in LDS (≥ the runs of one tile of 256 rows)
#define FDB_RUN_WAVE_LDS (FDB_RUN_STAGE * FDB_RUN_BYTES)  // a wave's runs wait in LDS over several tiles and leave as ONE contiguous copy: stores issued every tile
                                                      // made the next tile's loads wait for their acknowledgement (loads and stores share a counter): 3.06 → 2.3 ms per 100 M rows
// The WIDE run record (round 5): [the run's key tuple in the hash table's own layout — valid mask in words 0-1, two words of padding,
// one 32-bit word per dictionary column (key id, 0 = NULL / column absent), two per int64 column: key_words words | rows of the run: 2
// words | its aggregate: 2 words] = run_words 32-bit words (a multiple of 4). Any cardinality, int64 / computed keys, records that lack
// a group column. The scan does not carry key ids for it: a lane re-loads the columns (one 16-byte load per column for its 4 rows)
// only when one of its rows ENDS a run, the way an inserting lane of the table path does.
// The MEDIUM run record (round 5): the narrow record with TWO bytes per key id — [key ids: 64 bytes | rows: 8 | aggregate: 8] = 20 words —
// for dictionary columns of ≤ 65 534 distinct values (still ≤ FDB_RUN_TUPLE_BYTES columns, all present, ids packed in registers while the
// fingerprint is computed: no second pass over the columns). In FdbRunsOut.run_words / FdbRunSegs.run_words it is tagged with the value
// FDB_RUN_MEDIUM_WORDS = 1 (a wide record has ≥ 12 words; 20 itself is a legal wide stride).
#define FDB_RUN_MEDIUM_WORDS 1                        // the tag; the record itself has FDB_RUN_MEDIUM_BYTES
#define FDB_RUN_MEDIUM_BYTES 80
struct FdbRunsOut {
  unsigned char* tuples;         // [capacity][FDB_RUN_BYTES or run_words × 4]; nullptr: not a runs launch
  unsigned int* dir;             // [4 × tiles of the launch][2]
  unsigned int* chunk_cursor;    // next free chunk
  int32_t run_words;             // 0: the narrow record (FDB_RUN_BYTES); FDB_RUN_MEDIUM_WORDS: the medium record; else the wide record's words
  int32_t stage_cap;             // runs a wave's LDS stage holds: FDB_RUN_STAGE (narrow), FDB_RUN_WAVE_LDS / (run_words × 4) (wide)
};

struct FdbHashArgs {
  FdbScanArgs base;         // n_rows, filter program, aggregates (values / validity / func / type); dense fields unused
  FdbRunsOut runs;          // runs mode only
  const FdbHashCol* hcols;  // device array [n_hcols]
  unsigned long long* table;
  uint32_t* keys;           // [capacity][key_words]: the key tuple of each occupied slot (written once, by the inserter)
  unsigned long long* n_groups;  // device counter of occupied slots
  uint64_t mask;            // capacity - 1 (capacity is a power of two)
  int64_t row_begin;        // the launch covers rows [row_begin, row_end) of the record (multiples of 8)
  int64_t row_end;
  int32_t n_hcols;
  int32_t key_words;
  int32_t entry_words;
  int32_t canonical;        // 1: hcols[c].word == 4 + Σ widths of the columns before c and every column of the table is present
};

// ---- device helpers of the hash path (shared by fdb_kernels.hip and the kernels fdb_jit.cpp generates) ---------------
#ifdef FDB_DEVICE_HELPERS  // defined by translation units that hold device code
__device__ __forceinline__ unsigned long long fmix64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}
// Per-column contribution to the 128-bit fingerprint, summed over the non-NULL columns (order independent; a column
// that appears later leaves older fingerprints unchanged). Multilinear hashing: Σ_c x_c · K_c mod 2^64 with one odd
// 64-bit constant per column and per half — strongly universal, and one v_mad_u64_u32 pair per half for a 32-bit key
// id. int64 keys go through fmix64 first. fp_final adds the avalanche.
__device__ __forceinline__ void fp_add32(unsigned long long& h1, unsigned long long& h2, unsigned long long k1, unsigned long long k2, uint32_t id) {
  h1 += (unsigned long long)id * k1;
  h2 += (unsigned long long)id * k2;
}
__device__ __forceinline__ void fp_add(unsigned long long& h1, unsigned long long& h2, unsigned long long k1, unsigned long long k2, unsigned long long v) {
  const unsigned long long x = fmix64(v ^ 0x9FB21C651E98DF25ULL) | 1ull;
  h1 += x * k1;
  h2 += fmix64(x) * k2;
}
__device__ __forceinline__ void fp_final(unsigned long long& h1, unsigned long long& h2) {
  h1 = fmix64(h1 + 0x243F6A8885A308D3ULL); h2 = fmix64(h2 ^ 0xA5A5A5A5A5A5A5A5ULL);
  if (h1 == 0) h1 = 1;
  if (h2 == 0) h2 = 1;
}

// Finds the entry of fingerprint (h1, h2), inserting it if absent. Returns the slot and whether THIS lane inserted.
__device__ __forceinline__ uint64_t hash_find_or_insert(unsigned long long* table, uint64_t mask, int ew, unsigned long long h1,
                                                        unsigned long long h2, bool& inserted) {
  uint64_t slot = h1 & mask;
  inserted = false;
  for (;;) {
    unsigned long long* e = table + slot * (uint64_t)ew;
    unsigned long long prev = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == 0) prev = atomicCAS(e, 0ull, h1);
    if (prev == 0) {  // this lane owns the slot: publish the high half right away
      __hip_atomic_store(e + 1, h2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      inserted = true;
      return slot;
    }
    if (prev == h1) {
      const unsigned long long v = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v == h2) return slot;
      if (v == 0) { __builtin_amdgcn_s_sleep(1); continue; }  // the owner has not published yet: look again
    }
    slot = (slot + 1) & mask;
  }
}

#endif  // FDB_DEVICE_HELPERS

#define FDB_HASH_BLOCK 256
#ifndef FDB_DEVICE_ONLY
hipError_t fdb_launch_scan_hash(const FdbHashArgs& args, int grid_blocks, size_t lds_bytes, hipStream_t stream);
// table[i] = {0, 0, 0, idents…, 0 pad} for i < capacity
hipError_t fdb_launch_hash_init(unsigned long long* table, uint64_t capacity, int entry_words, int n_aggs, const unsigned long long* idents,
                                hipStream_t stream);
// Moves every occupied entry of (old_table, old_keys) into the (empty-initialised) new table; the first `old_used_words` words of
// each key tuple (valid mask + columns, without the tail padding of its stride `old_key_words`) are copied word for word into the
// (possibly wider) new key store, the other words zeroed. The key store itself needs NO initialisation: a
// tuple is only ever read where the entry is occupied, and every insert writes all of its words.
hipError_t fdb_launch_hash_rehash(const unsigned long long* old_table, const uint32_t* old_keys, uint64_t old_capacity, int old_key_words, int old_used_words,
                                  unsigned long long* new_table, uint32_t* new_keys, uint64_t new_mask, int entry_words, int new_key_words,
                                  hipStream_t stream);
// First output row of every 64-slot chunk of the table (bases[(capacity + 63) / 64], exclusive prefix sums of the occupied
// entries per chunk) and *n_out = occupied entries: what makes the two compactions below deterministic — SLOT ORDER, no atomics.
hipError_t fdb_launch_hash_chunk_bases(const unsigned long long* table, uint64_t capacity, int entry_words, uint32_t* bases, uint32_t* block_sums,
                                       unsigned long long* n_out, hipStream_t stream);
// In-place exclusive prefix sums of n 32-bit counts and their total. `block_sums`: scratch of (n + 1023) / 1024 + 1 words (counts are
// summed per 1 024, the sums scanned by one workgroup, then every block scans its own part: three launches); nullptr or
// n ≤ 4 096: one workgroup does everything.
hipError_t fdb_launch_exclusive_scan(uint32_t* counts, int64_t n, uint32_t* block_sums, unsigned long long* total, hipStream_t stream);
// Compacts the occupied entries in slot order: out_entries[i][0..entry_words-2] = {count, acc…} (fingerprints dropped),
// out_keys[i][key_words].
hipError_t fdb_launch_hash_compact(const unsigned long long* table, const uint32_t* keys, uint64_t capacity, int entry_words, int key_words,
                                   unsigned long long* out_entries, uint32_t* out_keys, const uint32_t* bases, hipStream_t stream);
// Finish for big tables: the occupied entries go straight into COLUMN buffers on the device, so the host only copies (and, for
// narrow dictionary columns, widens) finished buffers. cols[c].word / kind / gi describe the key tuple, cols[c].src_word the
// TRANSPORT width of the column in bytes (negative: BITS) — cols[c].lut, when set, maps a key id to the value that travels instead of id − 1 (fdb_launch_rank_ids) —; dictionary indices (= key id - 1) travel as uint8 / uint16 when the dictionary has
// ≤ 256 / ≤ 65 536 entries and are widened to Arrow's uint32 on the host (PCIe is the narrowest link of Finish), int64 keys are
// 8 bytes — and cols[c].lut_len != 0 marks a SLICED column: out_key[c] + slice · slice_stride + (row in slice) · width with
// 2^slice_shift rows per slice, so that one slice of all narrow columns is a contiguous run for the copy engine and the host can
// widen slice k while slice k + 1 is in flight. Other columns: out_key[c] + row · width. out_vals[0] = counts, out_vals[1 + j] =
// accumulator j (8 bytes per row); out_bits[c] = the column's Arrow validity bitmap, padded to a multiple of 8 bytes (whole
// 64-row words are stored); out_nulls[c] = its NULL count, ZEROED by the caller (n_cols ≤ 64). dense_keys: scratch of n_rows × key_words words (the occupied entries' key tuples as dense rows in
// slot order: pass 1 gathers them from the slot-indexed key store, pass 2 transposes 64 rows at a time into the columns).
struct FdbHashColumnsArgs {
  const unsigned long long* table; const uint32_t* keys; uint64_t capacity;
  const FdbHashCol* cols; void* const* out_key; uint8_t* const* out_bits; unsigned long long* const* out_vals;
  const uint32_t* bases;  // fdb_launch_hash_chunk_bases
  uint32_t* dense_keys; uint64_t n_rows; unsigned long long* out_nulls;
  uint64_t row_begin, row_end;  // pass 2 only: the output rows of this launch (row_begin a multiple of 64)
  uint64_t slice_stride;
  int32_t n_cols, entry_words, key_words, n_vals, slice_shift;
  int32_t any_lut;  // some column of `cols` leaves through a table (FdbHashCol::lut): pass 2 translates those in its LDS tile first
};
// Which key ids does the RESULT hold? A result column usually uses a small part of its dictionary (a query filters, a table's parts
// share big dictionaries), and what crosses PCIe is priced per row — so Finish can ship the ids at the width of the values PRESENT
// instead of the dictionary's (cfg 5 with 65 532-entry dictionaries of which 4 entries occur: 2 bits per row instead of 16).
// fdb_launch_present_ids: one pass over the dense key rows sets bit `id` of candidate k's bitmap (bitmaps + bm_off[k], 32-bit words;
// bit 0 = NULL is not recorded) — per-workgroup bitmaps in LDS, OR-ed into the global ones at the end; candidates are taken in groups
// whose bitmaps fit LDS. fdb_launch_rank_ids: per candidate, remap[id] = rank of id among the present ids (remap[0] = 0),
// present[rank] = id − 1 (the dictionary index the host widens to), counts[k] = number of present ids.
struct FdbPresentArgs {
  const uint32_t* dense_keys; uint64_t n_rows; int32_t key_words, n_cand;
  int32_t no_wave_set;        // A/B aid ($FDB_PRESENT_NO_SET): ask the global bitmap for every (column, id) instead of the wave's LDS set first
  uint32_t* bitmaps;          // zeroed by the caller
  uint32_t* remap;            // [Σ (dict_len[k] + 1)] at remap_off[k]
  uint32_t* present;          // same offsets
  unsigned long long* counts; // [n_cand]
  int32_t word[FDB_MAX_HASH_GCOLS];      // the candidate's word in a key row
  uint32_t dict_len[FDB_MAX_HASH_GCOLS]; // ids are 1 … dict_len
  uint32_t bm_off[FDB_MAX_HASH_GCOLS], remap_off[FDB_MAX_HASH_GCOLS];
};
hipError_t fdb_launch_present_ids(const FdbPresentArgs& args, int device, hipStream_t stream);
hipError_t fdb_launch_rank_ids(const FdbPresentArgs& args, hipStream_t stream);
hipError_t fdb_launch_hash_gather_rows(const FdbHashColumnsArgs& args, int device, hipStream_t stream);      // pass 1: all rows
hipError_t fdb_launch_hash_rows_to_columns(const FdbHashColumnsArgs& args, int device, hipStream_t stream);  // pass 2: rows [row_begin, row_end)

// Merges pre-aggregated groups into the table (fdb_merge.hip). Two kinds of source:
//   packed rows — `n` entries ({count, acc…}, in_entry_words × 8 bytes apart) + key tuples (in_key_words × 4 bytes apart);
//   a table     — src_table / src_keys / src_capacity, scanned in place (≙ the Synchronizer feeding a final-stage aggregate from a
//                 second chain's table: no packed intermediate).
// cols[c] describes DESTINATION column c: where its words sit in the incoming tuple (src_word, -1 = absent ⇒ NULL), a LUT
// translating incoming dictionary key ids (nullptr = identity) and, for int64 columns, its bit in the incoming valid mask (lut_len).
// funcs[j]: 1 add u64, 2 add f64, 3 min i64, 4 max i64, 0 skip.
struct FdbHashMergeArgs {
  const unsigned long long* entries; const uint32_t* in_keys; int64_t n;
  const unsigned long long* src_table; const uint32_t* src_keys; uint64_t src_capacity;
  unsigned long long* table; uint32_t* keys; unsigned long long* n_groups; uint64_t mask;
  const FdbHashCol* cols;   // device array [n_cols]
  int32_t n_cols, in_key_words, in_entry_words, entry_words, key_words, n_aggs;
  int32_t src_key_words, src_entry_words;
  int32_t in_words;       // words of an incoming tuple that the columns reach (0: its whole stride)
  int32_t same_layout;    // every destination column sits at the same word of the incoming tuple, none is absent: translated in place
  int32_t unique_source;  // every incoming group occurs once (a table, rows exported from ONE table): a new slot takes plain stores
  int32_t funcs[FDB_MAX_AGGS];
};
hipError_t fdb_launch_hash_merge(const FdbHashMergeArgs& args, int device, hipStream_t stream);

// ---- Finish of a run store (fdb_kernels.hip: runs_*_kernel) ---------------------------------------------------------------------
// A plan's runs live in SEGMENTS (one per launch), each with its own directory. Logical order = segments in launch order, within
// a segment directory entries in order, within an entry the runs in order — i.e. row order of the scan.
#define FDB_MAX_RUN_SEGMENTS 64
struct FdbRunSegs {
  const unsigned char* tuples[FDB_MAX_RUN_SEGMENTS];   // [run][FDB_RUN_BYTES or run_words × 4]
  uint32_t first_entry[FDB_MAX_RUN_SEGMENTS + 1];  // directory entries of segment s = [first_entry[s], first_entry[s + 1]) of the concatenated directory
  int32_t run_words[FDB_MAX_RUN_SEGMENTS];         // FdbRunsOut.run_words of the segment's launch: 0 narrow, FDB_RUN_MEDIUM_WORDS medium, else the wide record's stride in words
  int32_t n_segs;
};
// A group column as the run store's Finish sees it (the plan's column order): kind 0 dictionary (rank table of its key ids at
// rank + rank_off: rank[0] = NULL = 0xFFFFFFFF = last), 1 int64 / bool, 3 uint64 (NULL last, then by value); `word` = its place in a
// wide record / a dense key row, `gi` = its bit in the valid mask.
struct FdbRunCol { int32_t kind, word, gi; uint32_t rank_off; };
// Exclusive prefix sums of in[i * stride] (i < n) into out[i]; *total = the sum. `scratch`: ≥ (n / 1024 + 2) × 8 bytes.
hipError_t fdb_launch_scan_u32(const uint32_t* in, int stride, uint32_t* out, int64_t n, unsigned long long* scratch, unsigned long long* total, hipStream_t stream);
// phys[logical run] = (segment << 32) | index inside the segment, from the concatenated directory `dir` ([n_entries][2]) and the
// exclusive prefix sums `starts` of its counts.
hipError_t fdb_launch_runs_map(const uint32_t* dir, const uint32_t* starts, int64_t n_entries, const FdbRunSegs& segs, unsigned long long* phys, hipStream_t stream);
// flags[i] = 1 if run i starts a new key (its tuple differs from run i − 1's), else 0. *violation is set when a new key does not
// sort AFTER its predecessor — columns compared in plan order through `rank` ([n_cols][256]: rank of a key id among the column's
// values, NULL = 255 = last): then the input was not ordered and the caller falls back to the hash table.
hipError_t fdb_launch_runs_flags(const unsigned long long* phys, int64_t n_runs, const FdbRunSegs& segs, const unsigned char* rank, int n_cols, uint32_t* flags,
                                 unsigned int* violation, hipStream_t stream);
// The same for a run store with wide segments (any mix of narrow and wide): columns compared one by one in plan order through `cols`
// (device array [n_cols]) and the 32-bit rank tables `rank32`.
hipError_t fdb_launch_runs_flags_wide(const unsigned long long* phys, int64_t n_runs, const FdbRunSegs& segs, const FdbRunCol* cols, const uint32_t* rank32, int n_cols,
                                      uint32_t* flags, unsigned int* violation, hipStream_t stream);
// Runs whose keys did NOT arrive in order (several ordered sets pushed one after the other, a record out of place) are brought into key
// order by a stable LSD radix sort of (key, phys) pairs, one pass per group of columns from the LAST group column to the first — after it
// the flags / expand kernels above see one ordered set. A pass's 64-bit key:
//   mode 0: the ranks of `n` dictionary columns packed side by side (column col[k]'s rank << shift[k]; NULL = null_rank[k] = the column's
//           number of values, i.e. last), earlier columns in higher bits;
//   mode 1: int64 column col[0]'s value (sign bit flipped unless uint64; 0 for NULL);   mode 2: 1 where int64 column col[0] is NULL.
struct FdbRunKeyPass {
  int32_t mode, n;
  int32_t col[FDB_MAX_HASH_GCOLS], shift[FDB_MAX_HASH_GCOLS];
  uint32_t null_rank[FDB_MAX_HASH_GCOLS];
};
// Exactly one of `segs` (the things sorted are runs, phys[i] = (segment << 32) | index) and `rows` (they are dense key rows of `row_kw`
// words — an ordered plan's groups out of the hash table; a row has the layout of a wide run's key tuple — and phys[i] is a row number).
hipError_t fdb_launch_runs_sort_keys(const unsigned long long* phys, int64_t n_runs, const FdbRunSegs* segs, const uint32_t* rows, int row_kw, const FdbRunCol* cols,
                                     const uint32_t* rank32, const FdbRunKeyPass& pass, unsigned long long* keys, hipStream_t stream);
hipError_t fdb_launch_iota_u64(unsigned long long* p, int64_t n, hipStream_t stream);  // p[i] = i
hipError_t fdb_launch_gather_rows_u32(const uint32_t* src, int words, const unsigned long long* order, int64_t n, uint32_t* dst, hipStream_t stream);  // dst row i = src row order[i]
hipError_t fdb_launch_gather_u64(const unsigned long long* src, const unsigned long long* order, int64_t n, unsigned long long* dst, hipStream_t stream);  // dst[i] = src[order[i]]
// Stable sort of (key, value) pairs by the low `bits` bits of the key (fdb_sort.hip: rocPRIM's device radix sort — the one library primitive
// of the kernel set; it sits on the ordered Finish's fallback path, not on the scan). temp == nullptr: *temp_bytes = the scratch it needs.
hipError_t fdb_sort_pairs_u64(void* temp, size_t* temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned long long* vals_in,
                              unsigned long long* vals_out, int64_t n, int bits, hipStream_t stream);
// Groups out: for every run i, group g = out_idx[i] (exclusive prefix sums of flags) if flags[i] else out_idx[i] − 1 (flags == nullptr:
// every run is its own group, g = i). A group's first run writes its key tuple as one row of `dense_keys` ([n_groups][key_words]
// u32: valid mask in words 0-1, column c's id in word col_word[c]); counts and aggregates are folded into vals_cnt / vals_acc
// (func: 1 add u64, 2 add f64, 3 min i64, 4 max i64, 0 none) — plain stores for groups made of ONE run, atomics otherwise, so both
// arrays must hold the identity beforehand when flags != nullptr.
struct FdbRunsExpandArgs {
  const unsigned long long* phys; const uint32_t* flags; const uint32_t* out_idx; int64_t n_runs;
  uint32_t* dense_keys; unsigned long long* vals_cnt; unsigned long long* vals_acc;
  int32_t n_cols, key_words, func;
  int32_t val_stride;  // vals_cnt[g × val_stride], vals_acc[g × val_stride]; 0 = 1
  int32_t col_word[FDB_RUN_TUPLE_BYTES];
};
hipError_t fdb_launch_runs_expand(const FdbRunsExpandArgs& args, const FdbRunSegs& segs, hipStream_t stream);
// The runs of ANOTHER plan's run store, in their logical order, re-keyed into this plan's key ids and column order and written as one new
// segment of this plan (the merge of two ordered plans AS RUNS: ≙ OrderedSynchronizer's merge of the chains' sorted records,
// ordered_synchronizer.go:59-116, and the ordered aggregate's merge of its ordered sets, ordered_aggregate.go:449-470 — the sets meet
// at Finish, where the run store's sort brings them into one key order). cols[c] describes DESTINATION column c: kind, word (its place
// in a destination wide tuple), gi (= c), src_word (its place in a SOURCE wide tuple, −1: the source plan lacks the column), lut_len (the
// SOURCE column index: byte / half-word of a narrow / medium source run, valid-mask bit of an int64 column), lut (source id →
// destination id, nullptr = identity). The segment's directory is written too: entry k = (256 k, runs in block k).
struct FdbRunsTranslateArgs {
  const unsigned long long* phys; int64_t n_runs;
  const FdbHashCol* cols;
  unsigned char* out_tuples; uint32_t* out_dir;
  int32_t n_cols, out_run_words;  // out_run_words: 0 narrow, FDB_RUN_MEDIUM_WORDS medium, else the wide stride (key words + 4)
};
hipError_t fdb_launch_runs_translate(const FdbRunsTranslateArgs& args, const FdbRunSegs& segs, hipStream_t stream);
hipError_t fdb_launch_fill_u64(unsigned long long* p, int64_t n, unsigned long long v, hipStream_t stream);

// Export of a hash table for a merge elsewhere (another plan on this device, or — hash-partitioned — other ranks over RCCL):
// every occupied entry is re-keyed into the DESTINATION layout (per-column id translation, destination word positions and
// column indices), its fingerprint is recomputed there, and the entry is written as one packed row
//   [destination key tuple: dst_key_words × u32 | count | accumulators … | padding to a multiple of 16 bytes ]   (row_words32 words)
// into the region of the partition that owns it: partition = (fingerprint hi >> 32) % n_parts. The table slot uses the low
// bits of the OTHER fingerprint half, so partitions stay uniformly spread inside each receiver's table.
// cols[c] describes SOURCE column c: kind, word (source), src_word (DESTINATION word), gi (DESTINATION column index),
// lut (source id → destination id, nullptr = identity), lut_len (SOURCE column index: bit of the source valid mask), k1/k2 (of gi).
// Three launches without a host round trip: per-(wave, partition) counts, their prefix sums (region bases: partitions back to back,
// waves in order inside a partition — no atomics, so the rows of a partition are in SLOT order), the scatter. `counts` receives the
// rows per partition.
#define FDB_MAX_PARTS 64
static inline int fdb_packed_row_words(int dst_key_words, int n_vals) { return (dst_key_words + 2 * n_vals + 3) & ~3; }  // (dst_key_words is a multiple of 4: values are 8-byte aligned, rows 16-byte)
struct FdbHashPartArgs {
  const unsigned long long* table; const uint32_t* keys; uint64_t capacity;
  const FdbHashCol* cols;       // device array [n_cols]
  uint32_t* out;                // packed rows, 16-byte aligned
  unsigned long long* counts;   // [n_parts] rows per partition (written by the launch)
  unsigned long long* wave_bases; uint32_t* wave_counts;  // (set by fdb_launch_hash_partition from its scratch)
  int32_t n_cols, entry_words, key_words, dst_key_words, row_words32, n_vals, n_parts;
  int32_t in_words;     // words of a source tuple that the columns reach (0: the whole stride)
  int32_t same_layout;  // source and destination tuples have one layout
  int32_t same_ids;     // … and every column keeps its key ids and its index: the fingerprint stored in the table IS the destination's (the counting pass reads entries only)
};
size_t fdb_hash_partition_scratch_bytes(int device, const FdbHashPartArgs& args);
hipError_t fdb_launch_hash_partition(const FdbHashPartArgs& args, int device, void* scratch, hipStream_t stream);
#endif  // FDB_DEVICE_ONLY

// Identity elements stored in accumulators. MIN/MAX over float64 run on order-preserving int64 keys
// (fdb_f64_to_ordered) so one integer atomic serves both types.
#define FDB_I64_MAX 0x7FFFFFFFFFFFFFFFLL
#define FDB_I64_MIN (-0x7FFFFFFFFFFFFFFFLL - 1)

static inline int64_t fdb_f64_to_ordered_host(double d) {
  int64_t b;
  __builtin_memcpy(&b, &d, 8);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}
static inline double fdb_ordered_to_f64_host(int64_t k) {
  int64_t b = k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFFLL);
  double d;
  __builtin_memcpy(&d, &b, 8);
  return d;
}

// filter() in ONE pass over the filter columns (fdb_select_kernel, generated per predicate shape in fdb_jit.cpp; Plan::filter_batches):
// the workgroup that evaluates a share of four tiles also learns the share's place in its record's output and writes the compacted
// values of the filter columns it holds (≤ 8 bytes per row, columns without a validity bitmap) itself; the other columns follow in
// compact_multi_kernel from the bitmap and the tile offsets it leaves behind. Places come from the launch's scanner (the workgroup
// that arrives first): workers publish their share's count, the scanner sums them in order.
// `ctl`: a zero-initialised block the context keeps across launches —
//   word 0 the ticket counter shares are handed out from (in the order workgroups ask: a share's predecessors are then always held
//   by RUNNING workgroups, whatever else occupies the GPU), word 1 an error flag (a wait that did not end), word 2 the arrival counter
//   (who is the scanner), words [FDB_SELECT_CTL_WORDS, status_off) the selected rows of every record (FdbSelectPart::total points
//   there: the host fetches error word and totals with one copy), words [status_off …) one status word per share:
//   (epoch << 40) | (1 << 38) | count, words [place_off …) every share's place, sixteen words (128 bytes) apart: (epoch << 40) | prefix.
// A word of an earlier launch carries an earlier epoch and reads as "not there yet", so nothing is cleared between launches.
#define FDB_SELECT_CTL_WORDS 8
#define FDB_SELECT_MAX_FUSED 2
struct FdbSelectPart { void* dst[FDB_SELECT_MAX_FUSED]; unsigned long long* total; };  // per record: outputs of the fused columns (worst-case sized), selected rows
struct FdbSelectArgs {
  unsigned long long* ctl;
  unsigned long long ticket_base;   // value of the ticket counter when the launch starts
  unsigned long long arrival_base;  // value of the arrival counter when the launch starts
  const FdbSelectPart* sparts;
  uint32_t epoch;                   // 1 … 2^24 − 1
  uint32_t stage_off;               // LDS: the waves' staging regions start here (behind the predicate's LUTs)
  uint32_t status_off;              // first status word, in words from ctl
  uint32_t place_off;               // first place word, in words from ctl
  // Regions the launch zeroes on its way in — the validity bitmaps and NULL counters the compaction launch behind it ORs / adds into
  // (allocated at their worst-case size before this launch: the exact one is only known after it) — instead of a launch of their own:
  // pairs (address, bytes), 16-byte aligned, multiples of 16 bytes.
  const unsigned long long* zero;
  int32_t n_zero;
};

#ifndef FDB_DEVICE_ONLY
// ---- launch wrappers (fdb_kernels.hip) ---------------------------------------------------------------
// All launches are asynchronous on `stream`.
// rows_per_thread: 4 or 8 → the sequential (one column at a time) kernel; 0 → the slot kernel (needs args.n_c4/n_c8).
hipError_t fdb_launch_scan_dense(const FdbScanArgs& args, int grid_blocks, size_t lds_bytes, int rows_per_thread,
                                 hipStream_t stream);
// Number of workgroups fdb_launch_scan_dense will actually launch for `grid_blocks` requested (clamped to the tile count).
int fdb_scan_grid(const FdbScanArgs& args, int grid_blocks, int rows_per_thread);
// Folds the per-workgroup partial tables into the global table: state[arr * state_stride + slot] (op)= Σ_b partials[b][arr][slot].
// funcs[arr]: 0 skip, 1 add u64, 2 add f64, 3 min i64, 4 max i64. host_out (nullptr: none): pinned host memory laid out like `state`
// that receives every updated slot (and the untouched contents of the skipped arrays) — the table's host copy without a copy command.
hipError_t fdb_launch_reduce_partials(const unsigned long long* partials, int n_blocks, int n_arrays, uint32_t n_slots,
                                      unsigned long long* state, uint64_t state_stride, const int32_t* funcs, unsigned long long* host_out, hipStream_t stream);
// The slot kernel over `n_parts` records in ONE launch. `d_parts` is the device copy of the per-record argument
// blocks (global tile ranges filled in, in units of fdb_slot_geometry's tile_rows); `common` = any of them (table
// pointers, aggregation functions, LDS layout are identical across records). sub_tiles: 1 or 2.
// Measurement only: reads `bytes` (multiple of 16) once with the best streaming pattern of this box.
hipError_t fdb_launch_stream_read(const void* src, int64_t bytes, unsigned long long* out, hipStream_t stream);
hipError_t fdb_launch_scan_slots(const FdbScanArgs* d_parts, int n_parts, const FdbScanArgs& common, int64_t total_tiles, int grid_blocks,
                                 size_t lds_bytes, int two_phase, int mode, hipStream_t stream);
// Rows per tile and resident workgroups per CU (occupancy query on the instance that will run) of the slot kernel.
int fdb_slot_geometry(int two_phase, int mode, int lds_acc, size_t lds_bytes, int device, int* tile_rows, int* blocks_per_cu);
int fdb_slot_kernel_block(void);      // threads per workgroup of the slot kernel
hipError_t fdb_launch_fill_u64(unsigned long long* dst, unsigned long long value, int64_t n, hipStream_t stream);
// v[i] ← bits of the float64 whose order-preserving integer key (fdb_f64_to_ordered) v[i] holds, in place.
hipError_t fdb_launch_ordered_to_f64(unsigned long long* v, int64_t n, hipStream_t stream);
// base[a * n + i] = idents[a] for a < n_arrays (≤ 1 + FDB_MAX_AGGS), i < n: the whole partial table in one launch.
hipError_t fdb_launch_fill_state(unsigned long long* base, int64_t n, int n_arrays, const unsigned long long* idents, hipStream_t stream);
// host_out[a * stride + s] = state[a * stride + s] for s < n_slots, a < n_arrays (host_out: pinned host memory).
hipError_t fdb_launch_state_to_host(const unsigned long long* state, unsigned long long* host_out, uint32_t n_slots, uint64_t stride, int n_arrays, hipStream_t stream);
// Rank-ordered cross-GPU merge of a small dense table (fdb_kernels.hip, state_fold_ranks_kernel): packed[array][slot] = the first n_slots
// entries of every array; gathered = the ranks' packed tables back to back in rank order; ops[array]: 0 keep this rank's, 1 int64 sum,
// 2 float64 sum, 3 int64 min, 4 int64 max (Plan::state_array_op). The fold writes the table and, if given, its pinned host copy.
hipError_t fdb_launch_state_pack(const unsigned long long* state, unsigned long long* packed, uint32_t n_slots, uint64_t stride, int n_arrays, hipStream_t stream);
hipError_t fdb_launch_state_fold_ranks(const unsigned long long* gathered, int n_ranks, uint32_t n_slots, int n_arrays, const int32_t* ops, unsigned long long* state,
                                       uint64_t stride, unsigned long long* host_out, hipStream_t stream);
// dst[map[i]] (op)= src[i] for i < n; op: fdb_agg_func (SUM/COUNT add, MIN/MAX signed 64-bit, f64 SUM when is_f64).
hipError_t fdb_launch_merge_u64(unsigned long long* dst, const unsigned long long* src, const uint32_t* map,
                                int64_t n, int32_t func, int32_t is_f64, hipStream_t stream, const unsigned long long* src_cnt = nullptr);
// ---- filter() on the device (≙ filter.go:276-323): selection mask → tile offsets → compacted columns ----------------------------
// 1. fdb_launch_filter_flags evaluates the filter of `args` over rows [0, args.n_rows): one mask BIT per row (`masks`, bit i of the
//    bitmap = row i selected) and the number of selected rows per tile of FDB_COMPACT_TILE (2 048) rows (`tile_counts`, zeroed by the
//    caller). No barriers, no LDS beyond the filter's LUTs: it runs at full occupancy, which is what hides the latency of the
//    predicate interpreter's dependent loads.
// 2. fdb_launch_exclusive_scan turns the counts into exclusive prefix sums in place and writes the total.
// 3. fdb_launch_compact_col streams one column per launch. A tile belongs to ONE WAVE (no workgroup barriers anywhere: 32 independent
//    waves per CU overlap each other's load → stage → store phases): its lanes read their mask bits, find their output positions
//    with wave prefix sums, scatter the selected values into the wave's LDS staging buffer and write the buffer to its place in the
//    output with consecutive lanes writing consecutive elements. Rows keep their input order; nothing is gathered through an index
//    vector. (Round 4: for predicates whose filter columns it can hold, steps 1-2 and the filter columns' own compaction are ONE generated
//    kernel — FdbSelectArgs above. Round 2: a one-pass version with a decoupled look-back over tile totals was built and measured first: with an INTERPRETED
//    predicate every tile pays its 8 dependent load round trips inside the look-back chain — 0.13 ms per 25 M rows for the selection
//    vector alone, whatever the tile size, window or occupancy)
#define FDB_COMPACT_BLOCK 256
#define FDB_COMPACT_SUB 8                                  // a tile is 8 sub-tiles of (64 lanes × 4 consecutive rows)
#define FDB_COMPACT_SUBTILE (64 * 4)
#define FDB_COMPACT_TILE (FDB_COMPACT_SUBTILE * FDB_COMPACT_SUB)  // 2 048 rows, owned by one wave
#define FDB_COMPACT_STREAM_WAVE_LDS 6144                   // compact_multi_kernel: 4 KiB of values (512 × 8 or 1 024 × 4 bytes) + 512 B of dump slots, then
                                                           // ≤ 1 024 + 64 validity bytes — 24 KiB per workgroup, 6 workgroups (24 waves) per CU
#define FDB_COMPACT_WAVE_LDS 5120                          // staging bytes per wave: 4 KiB of values + 1 KiB of validity bytes
hipError_t fdb_launch_filter_flags(const FdbScanArgs& args, uint8_t* masks, uint32_t* tile_counts, int device, hipStream_t stream);
// One column per launch. width 4 / 8: values of that many bytes (`src` → `dst`, validity bitmap `src_valid` (nullptr: no NULLs) →
// validity BITMAP of the output in `dst_valid` (nullptr: not wanted; 8-byte aligned and ZEROED by the caller), null_count[0 … 63] (zeroed) receive the
// NULLs among the selected rows — the caller adds the 64 partial counts); width 0: the selection vector — ascending row numbers — into `dst` (uint32).
hipError_t fdb_launch_compact_col(int width, const void* src, const uint8_t* src_valid, void* dst, uint8_t* dst_valid, const uint8_t* masks,
                                  const uint32_t* tile_offsets, int64_t n_rows, unsigned long long* null_count, int device, hipStream_t stream);
// filter() over every record of a scan at once (Plan::filter_batches): global tiles of FDB_COMPACT_TILE rows, record r owns
// [recs[r].tile_begin, recs[r + 1].tile_begin) (the last one up to total_tiles); mask words [tile × 64, tile × 64 + 64).
struct FdbCompactRec { int64_t tile_begin; int64_t n_rows; };
struct FdbCompactCol { const void* src; const uint8_t* src_valid; void* dst; uint8_t* dst_valid; int32_t width; int32_t nullable; };  // width 4 / 8; nullable: the
// same for the column in EVERY record of the launch — then src_valid / dst_valid are never null (a record without NULLs passes an all-ones bitmap)
struct FdbZeroRegion { void* ptr; int64_t bytes; };  // 16-byte aligned, a multiple of 16 bytes
// offsets[t] = Σ tile_counts[< t] (mod 2^32), rec_base[r] = the same sum in full at record r's first tile, rec_base[n_recs] = the
// grand total. block_sums[b] = Σ tile_counts[1024 b …) (fdb_launch_sel_block_sums; nullptr: every workgroup adds up the counts in
// front of its block itself — cheaper than a launch while there are only a few dozen blocks).
hipError_t fdb_launch_sel_scan(const uint32_t* tile_counts, const unsigned long long* block_sums, int64_t total_tiles, uint32_t* offsets, const FdbCompactRec* recs,
                               int n_recs, unsigned long long* rec_base, hipStream_t stream);
hipError_t fdb_launch_sel_block_sums(const uint32_t* tile_counts, int64_t total_tiles, unsigned long long* block_sums, hipStream_t stream);
hipError_t fdb_launch_zero_regions(const FdbZeroRegion* regions, int n_regions, int64_t max_bytes, hipStream_t stream);
// Every column (cols[rec × n_cols + col]) of every record compacted in ONE launch; null_counts[(rec × n_cols + col) × 64 …] (zeroed).
// Wave g of the launch works on the column c with col_wave_begin[c] ≤ g < col_wave_begin[c + 1] (device array of n_cols + 1 entries,
// n_waves = its last one): the host deals the waves to the columns in proportion to their bytes.
int fdb_compact_multi_blocks_per_cu(int any_nullable);
hipError_t fdb_launch_compact_multi(const FdbCompactRec* recs, int n_recs, const FdbCompactCol* cols, int n_cols, int any_nullable, const int32_t* col_wave_begin, int n_waves,
                                    const uint32_t* masks, const uint32_t* tile_offsets, const unsigned long long* rec_base, int64_t total_tiles,
                                    unsigned long long* null_counts, hipStream_t stream);
int fdb_scan_default_grid(int device);
// ---- Parquet pages → columns in HBM (≙ pqarrow/arrow.go:711-823 writeColumnToArray + parquet-go's page decoders) ----
// The host parses page headers and run headers only (fdb_parquet.cpp); every per-row step runs here, gather-style: a thread owns
// output rows, finds the run that holds its level / value by binary search in the run tables and extracts it from the chunk's bytes.
// A run of the RLE / bit-packed hybrid encoding (definition levels, dictionary indices), in chunk-global numbering.
struct FdbPqRun {
  int64_t start;        // first row (definition-level runs) / first value rank (index runs) the run covers
  uint64_t payload;     // RLE: the repeated value; bit-packed: BIT offset of the run's first value in the chunk
  uint32_t bit_width;   // bits per bit-packed value (0 for RLE)
  uint32_t kind;        // 0 RLE, 1 bit-packed
};
// A data page with PLAIN fixed-width values: value ranks [rank_start, next page's rank_start) live at byte_off + 8·(rank − rank_start).
struct FdbPqPlainPage { int64_t rank_start; int64_t byte_off; };
// DELTA_BINARY_PACKED (INT64): a page's values are first + prefix sums of (min_delta of the block + bit-packed delta). The host
// reads the page / block headers only; a miniblock = `vpm` deltas packed at one bit width.
struct FdbPqDeltaMini { uint64_t bit_off; uint64_t min_delta; uint32_t width; uint32_t _pad; };  // bit_off: of its first delta, in the chunk
struct FdbPqDeltaPage {
  int64_t rank_start;    // rank (index among the column's non-NULL values) of the page's first value
  int64_t n_values;      // values in the page (≥ 1)
  uint64_t first_value;
  int32_t mini_begin;    // the page's miniblocks are minis[mini_begin …]; delta d of the page lives in miniblock d / vpm
  int32_t vpm;           // deltas per miniblock
};
// dense[rank] = value for every non-NULL value of the column (one workgroup per page: unpack, add min_delta, block-wide inclusive
// scan with the running total of the page carried from tile to tile; int64 arithmetic wraps like the reference's decoder).
// Snappy block format (format_description.txt) → bytes, on the device: one wave per page. Elements are byte-serial inside a page
// (every tag says where the next one starts and copies refer to output already produced), pages are independent: the wave parses the
// tags out of an LDS window of the compressed stream — every lane the same bytes, LDS latency instead of a global round trip per
// element — and all 64 lanes move the bytes of the element: literals from the stream, copies from the page's own output (a copy's
// source lies wholly before its destination, byte i of an overlapping pattern comes from source byte i mod offset: no lane waits for
// another). The page's most recent 64 KiB of output live in an LDS ring (what copies read and write: LDS latency per element instead
// of a trip to HBM) and leave for HBM in 16 KiB segments. status[page]: 0 = ok, else what failed first (1 length preamble, 2 truncated
// input, 3 output overrun, 4 bad offset, 5 output short of the announced length, 6 a copy from further back than the ring holds). fdb_batch_from_parquet inflates pages of literals with it; match-heavy pages stay on the host threads.
struct FdbSnappyPage { uint64_t src_off; uint64_t dst_off; uint32_t src_len; uint32_t dst_len; };
hipError_t fdb_launch_snappy_decode(const uint8_t* src, const FdbSnappyPage* pages, int32_t n_pages, uint8_t* dst, uint32_t* status, hipStream_t stream);
hipError_t fdb_launch_pq_delta(const uint8_t* chunk, const FdbPqDeltaPage* pages, int32_t n_pages, const FdbPqDeltaMini* minis, unsigned long long* dense,
                               hipStream_t stream);
// validity[w] = the 32 definition levels (max level 1) of rows 32w … 32w+31, counts[w] = popcount; rows ≥ n_rows are 0.
hipError_t fdb_launch_pq_validity(const uint8_t* chunk, const FdbPqRun* def_runs, int32_t n_runs, int64_t n_rows, uint32_t* validity, uint32_t* counts,
                                  hipStream_t stream);
// out[r] = the value of row r (0 for NULL rows). rank(r) = prefix[r / 32] + popcount(validity word below r) (prefix = exclusive
// scan of the counts); validity == nullptr: required column, rank(r) = r. kind 0: PLAIN 8-byte values (pages[]); kind 1: dictionary
// indices through idx_runs[] (uint32 out); kind 2: BOOLEAN bits through idx_runs[] (int64 out: 1 false / 2 true, 0 for NULL rows).
hipError_t fdb_launch_pq_decode(int kind, const uint8_t* chunk, const uint32_t* validity, const uint32_t* prefix, const FdbPqPlainPage* pages, int32_t n_pages,
                                const FdbPqRun* idx_runs, int32_t n_idx_runs, int64_t n_rows, void* out, hipStream_t stream);

// *flag |= 1 if a row i < n has idx[i] >= limit while its validity bit (validity == nullptr: every row) is set.
hipError_t fdb_launch_validate_indices(const uint32_t* idx, const uint8_t* validity, int64_t n, uint32_t limit, uint32_t* flag, hipStream_t stream);
// Local (in-process) communicator: dst[i] = reduce over r < n_srcs, in rank order, of srcs[r][i] for lo ≤ i < hi (8-byte elements;
// op 1 int64 sum, 2 float64 sum, 3 int64 min, 4 int64 max). `srcs` are device pointers of this or of peer devices; dst may be
// one of them (each element is read from every source before it is written).
hipError_t fdb_launch_peer_reduce(unsigned long long* dst, const void* const* srcs, int n_srcs, int64_t lo, int64_t hi, int op, hipStream_t stream);
#endif  // FDB_DEVICE_ONLY
