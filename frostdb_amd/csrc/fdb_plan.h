// fdb_plan.h — host side of the MI355X operator chain: the C++ mirror of the reference's push operators
// (PhysicalPlan: Callback / Finish / Draw / Close, query/physicalplan/physicalplan.go:24-30) for the fused
// PredicateFilter → HashAggregate(final=false) chain, plus the HBM-resident record (`DeviceBatch`).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "fdb_arrow.h"
#include "fdb_regex.h"
#include "fdb_kernels.h"

namespace fdb {

void hip_check(hipError_t e, const char* what);
// Counts and array pointers of a plan descriptor agree (no negative count, no missing array, no projection without nodes): what every
// reader of the descriptor — the dynamic-aggregation copy, Draw, the plan itself — may then rely on. Throws FDB_ERR_INVALID.
void check_desc_shape(const fdb_plan_desc* d);

// One column of a record resident in HBM.
struct DevColumn {
  std::string name;
  std::string format;           // Arrow format of the column as received (index format for DICT)
  ColKind kind = ColKind::OTHER;
  int64_t length = 0;
  int64_t null_count = 0;
  void* d_values = nullptr;     // int64/uint64/double values or uint32 dictionary indices; nullptr for STR/OTHER/BOOL
  uint8_t* d_validity = nullptr;  // validity bitmap at bit offset 0; nullptr ⇔ null_count == 0
  std::shared_ptr<HostDict> dict;
  int64_t value_bytes = 0;      // algorithmic bytes: values/indices
  int64_t validity_bytes = 0;   // algorithmic bytes: bitmap (0 when the column has no nulls)
};

struct DeviceBatch {
  int device = 0;
  int64_t rows = 0;
  std::vector<DevColumn> cols;
  void* arena = nullptr;        // one allocation per record
  size_t arena_bytes = 0;
  std::vector<void*> extra_arenas;  // filter() results: columns the one-pass kernel wrote before the row count was known (worst-case sized pool blocks)
  bool arena_borrowed = false;  // the arena is a piece of a plan's record slab (small pushed records): nothing to free here
  class Context* arena_ctx = nullptr;  // transient batches (fdb_plan_push): the arena is borrowed from the plan's block cache
  int64_t payload_bytes = 0;    // Σ value_bytes + validity_bytes
  // Plans scan resident records asynchronously on their own streams and the caller may release a record as soon as the push
  // call has returned: every launch that reads the arena notes its stream here, and the destructor waits for those streams
  // before the block goes back to the pool (where another import, on another stream or thread, would overwrite it). Streams
  // belong to pooled Contexts, which live as long as the process. Thread-safe (N chains push one record concurrently).
  void note_reader(hipStream_t s) const;
  ~DeviceBatch();
  // Exactly-one-field lookup like ArrayRef.ArrowArray (binaryscalarexpr.go:22-29): -1 if absent or ambiguous.
  int find(const std::string& name) const;

 private:
  mutable std::mutex readers_mu_;
  mutable std::vector<hipStream_t> readers_;
};

// Stages the columns of `view` accepted by `want(name)` (nullptr ⇒ all) to the device.
// With `ctx`, the arena comes from (and returns to) that context's block cache and the copies are asynchronous on
// `stream` (one synchronisation at the end) — the per-record cost of fdb_plan_push; without, a private hipMalloc
// (resident batches, which outlive any plan).
// `via_ring` (with ctx): every buffer is first copied into the context's pinned ring (the source is fully read when the call
// returns, nothing is waited for) — the deferred, coalescing mode of fdb_plan_push for small records.
// `sink` (small pushed records): the record's device bytes are a piece of a slab the caller owns — sink(bytes, &device, &pinned) names
// where they go and where they are assembled on the host; the CALLER ships the pinned piece (one DMA for many records).
typedef std::function<void(size_t bytes, void** dev, unsigned char** pinned)> RecordSink;
std::unique_ptr<DeviceBatch> import_batch(const HostRecordView& view, int device,
                                          const std::function<bool(const std::string&)>* want, hipStream_t stream, class Context* ctx = nullptr,
                                          bool via_ring = false, const RecordSink* sink = nullptr);

// Parquet column chunks of one row group → a resident batch (fdb_parquet.cpp).
std::unique_ptr<DeviceBatch> batch_from_parquet(const fdb_parquet_chunk* chunks, int32_t n_chunks, int64_t n_rows, int device);
// … of several row groups in one call: one copy queue, the host work of all of them side by side (fdb_batches_from_parquet).
std::vector<std::unique_ptr<DeviceBatch>> batches_from_parquet(const fdb_parquet_row_group* groups, int32_t n_groups, int device);
void parquet_stats(int64_t* calls, double* host_ms, double* device_ms, int64_t* file_bytes, int64_t* out_bytes);
// The record of a resident batch as Arrow in host memory.
void export_batch(const DeviceBatch& b, ArrowArray* out, ArrowSchema* out_schema);

struct Literal {
  int32_t type = FDB_LIT_NULL;
  int64_t i64 = 0;
  uint64_t u64 = 0;
  double f64 = 0;
  std::string bytes;
  bool valid() const { return type != FDB_LIT_NULL; }
  std::string str() const;
};

// Truth tables of filter leaves (one answer per dictionary entry + one for NULL) are kept per (filter node, dictionary):
// dictionaries with equal content are ONE object (read_dictionary / encode_plain intern them), the parts of a table mostly share
// theirs, and a regex leaf costs a regex match per entry — with the host's engine (fdb_plan_desc.regex_match) a call back into
// the application per entry.
struct TruthCache {
  struct Entry { std::shared_ptr<HostDict> dict; std::shared_ptr<const std::vector<uint8_t>> truth; };
  std::unordered_map<uint64_t, std::vector<Entry>> by_key;  // key = node index mixed with the dictionary's address
  size_t entries = 0;
  static uint64_t key(int node, const HostDict* d) { return (uint64_t)(uintptr_t)d * 0x9E3779B97F4A7C15ull + (uint64_t)node; }
  std::shared_ptr<const std::vector<uint8_t>> find(int node, const HostDict* d) const {
    auto it = by_key.find(key(node, d));
    if (it == by_key.end()) return nullptr;
    for (const Entry& e : it->second) if (e.dict.get() == d) return e.truth;
    return nullptr;
  }
  void put(int node, const std::shared_ptr<HostDict>& d, std::shared_ptr<const std::vector<uint8_t>> t) {
    if (entries >= 4096) { by_key.clear(); entries = 0; }  // (bounded: a scan over parts with ever-changing dictionaries)
    by_key[key(node, d.get())].push_back(Entry{d, std::move(t)});
    entries++;
  }
};

struct ExprNode {
  int32_t op = 0, left = -1, right = -1;
  std::string column;
  Literal lit;
  std::shared_ptr<const Regex> re;                          // the built-in RE2-syntax engine (fdb_regex.h; no host matcher given)
  fdb_regex_match_fn re_fn = nullptr;                       // the host application's engine (fdb_plan_desc.regex_match)
  void* re_user = nullptr;
  bool regex_matches(const std::string& v) const;           // unanchored match of this leaf's pattern against `v`
};

struct AggState {
  int32_t func = 0;
  std::string column;
  std::string result_name;        // "sum(value)" — AggregationFunction.Name() (logicalplan/expr.go:700-702)
  std::string emit_name;          // the result column's name: result_name, except a partial-stage OrderedAggregate's (the column's own)
  int32_t type = FDB_T_NONE;      // FDB_T_I64 / FDB_T_F64 once a batch has shown the column; COUNT keeps NONE
  unsigned long long* d_acc = nullptr;
  // Composite reducers ride on MIN / MAX accumulators (aggs_ is the PHYSICAL list, one accumulator array each):
  //   UNIQUE(x) (aggregate.go:677-732) = MIN(x) with NULL ↦ INT64_MIN  +  MAX(x) with NULL ↦ INT64_MAX: the group has one
  //     non-NULL value iff min == max (a NULL row drives the two apart); role 1 = the MIN half (emits the column), 2 = the MAX half
  //   AND(x)    (aggregate.go:635-675) = MIN over the bool column widened to 1 (false) / 2 (true) with NULL ↦ 2 (NULLs are skipped); role 3
  int32_t role = 0;
  unsigned long long null_value = 0;  // what a NULL row contributes (FdbAgg::null_value)
};

// One concrete group-by column, in first-seen order (≙ hashAggregate.colOrdering, aggregate.go:155).
// Key ids are assigned per distinct dictionary VALUE, first seen first (id 0 is NULL / column absent), and
// never change, so the entry→id table of a given dictionary is computed once and cached by content.
struct GroupColState {
  std::string name;
  int kind = 0;                                             // 0 dictionary column, 1 int64 column (hash table only)
  bool plain = false;                                       // kind 0 fed by a plain string / binary column: emitted as one, not as a dictionary
  bool is_u64 = false;                                      // kind 1 fed by a uint64 column: emitted as one
  bool is_bool = false;                                     // kind 1 holding a bool (1 = false, 2 = true; a stored column or a boolean projection): emitted as an Arrow bool column
  int word = -1;                                            // hash table: first word of this column in the key tuple
  std::string value_format = "z";
  std::vector<std::string_view> values;                     // id - 1 → value (views into `owners`)
  std::vector<std::shared_ptr<const HostDict>> owners;      // keep the viewed strings alive
  std::shared_ptr<const HostDict> adopted;                  // the first dictionary, when its entries became ids 1 … n wholesale (lut_for); still the whole value list while values.size() == its size
  const HostDict* whole_dictionary() const { return adopted && !adopted->plain && values.size() == adopted->values.size() ? adopted.get() : nullptr; }
  uint32_t cap = 1;                                         // ids live in [0, cap)
  uint32_t stride = 1;

  // entry → id table for dictionary `d` (assigns ids to values not seen before).
  std::shared_ptr<const std::vector<uint32_t>> lut_for(const std::shared_ptr<HostDict>& d);
  uint32_t intern(std::string_view v);                      // id of `v`, assigning the next id if new (caller keeps v alive via owners)

 private:
  struct Cached { std::shared_ptr<const HostDict> dict; std::shared_ptr<const std::vector<uint32_t>> lut; };
  std::unordered_map<std::string_view, uint32_t> ids_;      // built lazily: values[0 .. ids_built_) are present
  size_t ids_built_ = 0;
  std::unordered_multimap<uint64_t, Cached> lut_cache_;     // by HostDict::hash
  void build_ids();
};

struct GroupMatcher { std::string name; bool dynamic; };

// A computed column of the Projection between filter and aggregate (fdb_projection; project.go:58-161).
struct ProjNode { int32_t kind = 0, op = 0, left = -1, right = -1; std::string column; int32_t lit_type = 0; int64_t i64 = 0; double f64 = 0; Literal lit; };  // (lit: the literal as a filter leaf takes it — a comparison of a column with a string / binary / NULL literal is evaluated as one, project.go:409-470)
struct Projection { std::string name; std::vector<ProjNode> nodes; int32_t root = -1; };

enum class TableMode { DENSE, HASH };

// Occupied groups of a plan's table in host memory, independent of the table's device representation.
struct CompactState {
  int64_t n = 0;
  std::vector<unsigned long long> cnt;                  // [n]
  std::vector<std::vector<unsigned long long>> acc;     // [aggregation][n]
  std::vector<std::vector<uint32_t>> ids;               // [group column][n] dictionary key ids (0 = NULL)
  std::vector<std::vector<int64_t>> ivals;              // [group column][n] int64 keys
  std::vector<std::vector<uint8_t>> ivalid;
};

class Context;  // per-device stream + cached device/pinned memory (fdb_context.h)
class Comm;     // one rank's endpoint of a cross-GPU communicator (fdb_comm.h)

// The group columns of a plan — names, kinds, distinct key values in id order — and the value types of its aggregations: what
// ranks exchange to agree on one key-id space before a hash-partitioned merge (fdb_plan_exchange).
struct GroupSchemaCol {
  std::string name;
  int kind = 0;
  bool is_bool = false, is_u64 = false, plain = false;
  std::string value_format = "z";
  std::vector<std::string> values;
};
struct GroupSchema { std::vector<GroupSchemaCol> cols; std::vector<int32_t> agg_types; };

class Plan {
 public:
  Plan(const fdb_plan_desc* desc, int device, bool explain_only = false);
  struct CloneTag {};
  Plan(const Plan& proto, CloneTag);  // a fresh plan of the same descriptor on the same device (no state)
  Plan(const Plan&) = delete;
  Plan& operator=(const Plan&) = delete;
  ~Plan();

  void push(const ArrowArray* array, const ArrowSchema* schema);        // ≙ Callback
  // Small host records are not scanned one by one: push validates them, copies what the plan references into pinned staging
  // (the caller's buffers are only borrowed) and queues the copy; ONE launch covers the queued records once enough rows are
  // pending. settle() launches whatever is queued — every entry point that reads or merges state calls it first.
  void settle();
  void push_batch(const DeviceBatch& batch);
  void push_batches(const DeviceBatch* const* batches, int n);         // one fused launch over n resident records
  void finish(ArrowArray* out, ArrowSchema* out_schema, int64_t* n_rows);  // ≙ Finish
  // The records a Finish still owes: the reference starts a new output record when a plain string / binary key builder would pass
  // 2 GiB (aggregate.go:426-468, optbuilders.go:221-224); finish() emits the first, finish_next() the others (false: none left).
  bool finish_next(ArrowArray* out, ArrowSchema* out_schema, int64_t* n_rows);
  std::vector<std::pair<std::vector<OutColumn>, int64_t>> pending_out_;
  // ≙ Finish for a consumer on the device: the result record as a resident batch (group columns as dictionary / int64 / bool columns,
  // one column per aggregation). Big hash tables are materialised in HBM without crossing PCIe; small tables take the host route.
  std::unique_ptr<DeviceBatch> finish_batch(int64_t* n_rows);
  // The result as column descriptors (group-key columns first — n_key_columns() of them — then one column per aggregation).
  int64_t finish_columns(std::vector<OutColumn>* cols);
  size_t n_key_columns() const { return gcols_.size(); }
  void merge_from(Plan& src);                                          // ≙ Synchronizer + final stage
  void select(const ArrowArray* array, const ArrowSchema* schema, uint32_t* indices, int64_t capacity, int64_t* n_selected);
  void filter(const ArrowArray* array, const ArrowSchema* schema, ArrowArray* out, ArrowSchema* out_schema, int64_t* n_selected);
  // The same for a record resident in HBM, results staying in HBM: the compacted record as a new resident batch / the
  // selection vector in a DEVICE buffer of `capacity` ≥ rows entries.
  std::unique_ptr<DeviceBatch> filter_batch(const DeviceBatch& in, int64_t* n_selected);
  // The same for every record of a scan at once: one launch sequence, two host round trips in total (sizes, NULL counts).
  std::unique_ptr<DeviceBatch> filter_batch_interp(const DeviceBatch& in, int64_t* n_selected);
  std::vector<std::unique_ptr<DeviceBatch>> filter_batches(const DeviceBatch* const* in, int n, int64_t* n_selected);
  struct SelectStall {};  // the one-pass select kernel ran into its poll bound: filter_batches retries through the three-launch path
  std::vector<std::unique_ptr<DeviceBatch>> filter_batches_impl(const DeviceBatch* const* in, int n, int64_t* n_selected, bool force_two_pass);
  int64_t select_batch(const DeviceBatch& in, uint32_t* d_indices, int64_t capacity);
  const char* draw();                                                  // ≙ Draw
  int64_t num_groups();
  void partial_keys(ArrowArray* out, ArrowSchema* out_schema);
  void partial_state(int32_t agg, void* dst, int64_t capacity_bytes);
  char agg_format(int32_t agg) const;
  uint64_t state_signature(int64_t* n_slots);
  void state_read(int32_t array, void* dst, int64_t capacity_bytes);
  void state_pointers(void** base, int64_t* array_stride, int64_t* n_slots);
  void state_write(int32_t array, const void* src, int64_t bytes);
  // Reduction that merges table array `array` (0 = row counts, 1 + j = physical accumulator j) across plans / ranks:
  // 0 none (unused array), 1 integer sum, 2 float64 sum, 3 integer min, 4 integer max.
  int32_t state_array_op(int32_t array) const;
  int32_t num_state_arrays() const { return (int32_t)(1 + aggs_.size()); }
  bool has_composite_aggs() const { for (const AggState& a : aggs_) if (a.role != 0) return true; return false; }

  // Hash-table exchange (fdb_hash.cpp): merge of high-cardinality partial tables without a host round trip — between two plans
  // of one device (merge_from) and, hash-partitioned, between ranks (frostdb_amd/distributed.py: merge_plan_alltoall).
  void group_schema(ArrowArray* out, ArrowSchema* out_schema);  // zero-row record: one column per group column, carrying its dictionary
  void seed_groups(const ArrowArray* array, const ArrowSchema* schema);  // adopt those columns and dictionary values, in that order
  // Re-keys every occupied entry into `layout`'s columns / key ids and packs it into partition (fingerprint % n_parts);
  // *dev_rows (owned by this plan until its next push / close) holds the partitions back to back, counts[p] rows each.
  void hash_export(Plan& layout, int n_parts, void** dev_rows, int64_t* counts, int32_t* row_words32);
  // rows packed for THIS plan's layout. `unique_rows`: every group occurs at most once among them (the rows one rank exported from its
  // one table) — a group they create takes plain stores
  void hash_import(const void* dev_rows, int64_t n_rows, bool unique_rows = false);
  // Cross-GPU merges over a communicator (fdb_comm.cpp; ≙ Synchronizer + final stage, synchronize.go:31-53). Collective calls.
  bool comm_allreduce(Comm& comm);               // aligned dense tables: in-place all-reduce on this plan's stream; false = layouts differ, nothing changed
  void comm_exchange(Comm& comm, Plan& shard);   // any tables: schema agreement + hash-partitioned exchange into `shard` (a fresh clone)
  GroupSchema export_schema() const;
  void adopt_schema(const GroupSchema& s);       // adopt columns / key values in that order (switches to the hash table)

  std::string error;
  int device() const { return device_; }
  hipStream_t stream() const { return stream_; }
  bool timing = false;
  int64_t stat_bytes = 0, stat_launches = 0, stat_rows = 0;
  double stat_ms = 0;
  double stat_merge_ms = 0;      // device time of cross-GPU merges (hipEvent pairs around the collectives of comm_allreduce / comm_exchange)
  int rows_per_thread = 0;  // 0: slot (load-hoisting) kernel; 4 / 8: sequential kernel
  int grid_override = 0;
  int ablate = 0;
  int sub_tiles = 0;  // kernel variant mode: 0 default, 1: 512 thr, 2: 256 thr, 3: 1024 thr, 4: interpreting kernel only (no plan specialisation)
  const char* last_kernel() const { return last_kernel_; }
  bool deterministic = false;  // fdb_plan_set_deterministic: wave-private LDS tables, fixed-order folds; scans that cannot have them are refused
  bool use_partials = true;  // LDS mode: flush workgroup tables with plain stores + a fold kernel instead of atomics
  struct Resolved;  // per-batch kernel arguments (fdb_plan.cpp)
  void sync();      // waits for the plan's stream; timing events are read, scratch and consumed records go back to their caches

 private:
  const char* last_kernel_ = "";  // name of the scan kernel of the latest push
  bool references(const std::string& column) const;
  TruthCache truth_cache_;
  std::vector<Projection> projs_;
  const Projection* find_projection(const std::string& name) const;
  // Appends the nodes of `p` to R->args.expr (columns resolved against `b`, types checked); returns the root's index.
  int resolve_projection(const Projection& p, const DeviceBatch& b, Resolved* R);
  void resolve_batch(const DeviceBatch& b, Resolved* R, std::vector<int>* batch_gcols);
  void resolve_filter_only(const DeviceBatch& b, Resolved* R);  // predicate program + LUTs (staged), nothing else
  int64_t count_subtree(const DeviceBatch& b, int node);        // rows of `b` selected by the filter sub-tree rooted at `node` (lazy AND)
  int64_t run_flags(const FdbScanArgs& a, uint8_t** d_masks, uint32_t** d_offsets);  // selection bitmap + tile offsets; returns the number selected
  void ensure_layout(const std::vector<uint32_t>& new_caps);
  void collect_timing();
  void fetch_state(std::vector<unsigned long long>* cnt, std::vector<std::vector<unsigned long long>>* acc);
  void fetch_compact(CompactState* cs);
  void build_key_columns(const CompactState& cs, std::vector<OutColumn>* cols) const;
  void build_agg_columns(const CompactState& cs, std::vector<OutColumn>* cols) const;
  void sort_compact(CompactState* cs) const;  // rows in ascending group-key order, NULLs last (OrderedAggregate's output order)
  // high-cardinality path (fdb_hash.cpp)
  void switch_to_hash();
  void hash_layout();                                   // (re)assign key-tuple words; widen the key store if columns were added
  void hash_reserve(uint64_t extra_groups, uint64_t expected_groups = 0);  // capacity ≥ 2 × (max(groups, expected) + extra): grow + rehash on the device
  void push_hash(const DeviceBatch* const* bs, std::vector<Resolved>& Rs, const std::vector<int>& live, bool runs = false);
  // Table-free OrderedAggregate (ordered_aggregate.go:163-551; fdb_kernels.h "run store"): while an ordered plan's records keep
  // the shape the run kernel needs, their runs of equal keys are collected instead of probing a table. Finish merges what wave /
  // record boundaries cut and, if the keys came in order, emits them as they are; anything else that wants the plan's state
  // (a merge, an export, the raw accessors, input that was NOT ordered) first inserts the runs into the hash table.
  struct RunSegment {
    void* block = nullptr;  // one device allocation: [runs of FDB_RUN_BYTES (narrow) or run_words × 4 bytes (wide) | directory | chunk cursor]
    unsigned char* tuples = nullptr; uint32_t* dir = nullptr; uint32_t* cursor = nullptr;
    int64_t n_entries = 0, capacity = 0;
    int32_t run_words = 0;  // 0: narrow records; else the wide record's stride in 32-bit words (fdb_kernels.h FdbRunsOut)
  };
  struct RunsView {  // the runs in logical (row) order, prepared for Finish
    FdbRunSegs segs;
    unsigned long long* phys = nullptr; uint32_t* flags = nullptr; uint32_t* out_idx = nullptr;
    int64_t n_runs = 0, n_groups = 0;
    const FdbRunCol* d_cols = nullptr; const uint32_t* d_rank32 = nullptr;  // the columns' places and 32-bit rank tables on the device (runs_rank_tables), once somebody needed them
  };
  // Small pushed records are not allocated one by one (a record held until the next sync would cost a hipMalloc each: ≈25 µs, and
  // a process-wide lock that N chains fight over): their bytes are pieces of a SLAB — a device block and a pinned block of the same
  // size, filled at the same offsets — shipped with one DMA per ≈2 MiB and recycled when the stream is next idle.
  struct RecordSlab { void* d = nullptr; unsigned char* h = nullptr; size_t cap = 0, used = 0, shipped = 0; };
  RecordSlab slab_;
  std::vector<RecordSlab> inflight_slabs_;
  void slab_reserve(size_t bytes, void** dev, unsigned char** pinned);
  void slab_ship();
  double prof_push_[4] = {0, 0, 0, 0};  // $FDB_PROFILE_PUSH
  int64_t prof_push_n_ = 0;
  std::vector<RunSegment> runs_;
  bool runs_wanted(const DeviceBatch* const* bs, const std::vector<Resolved>& Rs, const std::vector<int>& live) const;
  int runs_format(const Resolved& R) const;  // the run record this record's launch writes: 0 narrow (a byte per key id), 1 medium (two bytes), 2 wide (the table's key tuple)
  void runs_free();
  void runs_to_table();
  // false: the keys did not arrive in order (the caller falls back to runs_to_table + the ordinary ordered Finish). Device blocks it
  // allocates are appended to `owned` (freed by the caller through ctx_->dev_free).
  bool runs_prepare(RunsView* v, bool check_order, std::vector<void*>* owned);
  void runs_rank_tables(RunsView* v, std::vector<void*>* owned);  // fills v->d_cols / d_rank32 (no-op when they are there)
  // The runs of `v` brought into key order by a device sort (several ordered sets, a record out of place): phys / flags / out_idx / n_groups
  // are replaced. false: not attempted (too many runs, $FDB_RUNS_NO_SORT) — the caller falls back to the table.
  bool runs_sort(RunsView* v, std::vector<void*>* owned);
  // A/B and test switches of the environment, read ONCE per plan (at create): getenv on the per-record path costs a scan of the environment
  // and is not safe against a concurrent setenv. A test sets them before it creates its plan.
  struct Knobs {
    bool no_jit, runs_always, no_identity_lut, runs_no_sort, no_uniform_fold, no_present_ids;
    char runs_wide;             // 0 unset, '1' wide records every launch, 'm' medium where narrow would do
    long long ordered_sort_min; // groups from which an ordered Finish out of the table sorts on the device
    long long present_ids_min_bytes;  // index bytes a Finish must stand to save before it ranks the ids present ($FDB_PRESENT_IDS_MIN_BYTES), 32 MiB
    long long max_key_bytes;    // bytes one plain string / binary key column of ONE output record may hold ($FDB_TEST_MAX_KEY_BYTES; math.MaxInt32 like optbuilders.go:221-224)
    int finish_slice_shift;     // log2 of the rows per Finish slice ($FDB_FINISH_SLICE_SHIFT: tests reach several slices with a small result), 20
    Knobs();
  } knobs_;
  int64_t fresh_groups_ = -1;   // hash_groups() as just fetched by ordered_finish_on_device(), for the finish_columns_hash that follows at once
  bool ordered_finish_on_device();  // an ordered plan's groups out of the hash table: sorted on the device (big results) or on the host
  // The stable LSD sort both ordered Finishes use (fdb_hash.cpp): runs of a run store, or the groups of the hash table as dense key rows.
  unsigned long long* sort_by_group_columns(unsigned long long* order, int64_t n_things, const FdbRunSegs* segs, const uint32_t* rows, int row_kw, RunsView* tables,
                                            std::vector<void*>* owned);
  int32_t runs_func() const;  // how two runs' aggregates fold (FdbRunsExpandArgs.func)
  void hash_merge_device(const unsigned long long* d_entries, const uint32_t* d_keys, int64_t n, int in_kw, const std::vector<FdbHashCol>& cols, bool unique_source);
  void hash_merge_args(FdbHashMergeArgs* m, const std::vector<FdbHashCol>& cols, int in_stride_words);
  void merge_hash_tables(Plan& src);
  bool merge_runs(Plan& src);  // two ordered plans that hold runs only: the source's runs become one more ordered set of ours
  int64_t finish_columns_runs(std::vector<OutColumn>* cols, DeviceBatch* resident, bool* ok);
  void fetch_compact_hash(CompactState* cs);
  int64_t finish_columns_hash(std::vector<OutColumn>* cols, DeviceBatch* resident = nullptr, const RunsView* runs = nullptr);  // device-side column materialisation (big result sets)
  void merge_hash(Plan& src);
  uint64_t hash_groups();                               // occupied slots (reads the device counter; waits for the stream)
  void hash_insert_entries(const std::vector<unsigned long long>& entries, const std::vector<uint32_t>& keys, int64_t n, int in_kw,
                           const std::vector<FdbHashCol>& cols);
  void* upload(const void* host, size_t bytes);  // async H2D through the pinned pool; returns device address

  int device_;
  hipStream_t stream_ = nullptr;
  std::vector<ExprNode> filter_;
  int32_t filter_root_ = -1;
  std::vector<AggState> aggs_;
  std::vector<GroupMatcher> matchers_;
  bool final_stage_ = false;
  bool ordered_ = false;            // OrderedAggregate: one aggregation, result sorted by the group columns
  bool finished_ = false;

  std::vector<GroupColState> gcols_;
  uint32_t n_slots_ = 1;            // Π cap
  uint64_t slots_alloc_ = 0;        // allocated accumulator length
  unsigned long long* d_state_ = nullptr;  // one block: [cnt | acc 0 | acc 1 | …], slots_alloc_ entries each
  unsigned long long* d_cnt_ = nullptr;
  bool state_dirty_ = false;        // any kernel has accumulated into the table
  bool state_virgin_ = false;       // d_state_ is allocated but NOT identity-filled yet: the first specialised scan launch fills it in its
                                    // prologue (push_batches); anything else that reads or accumulates into it calls materialize_state() first
  void materialize_state();
  // Host copy of a small dense table, written by reduce_partials_kernel itself (pinned memory): valid from a push whose fold wrote
  // it until anything else changes the table (another push, a merge, an all-reduce, a raw write, a re-layout). Finish reads it
  // after the stream's sync instead of queueing a device→host copy first.
  unsigned long long* h_mirror_ = nullptr;
  size_t mirror_bytes_ = 0;
  bool mirror_valid_ = false;
  unsigned long long* mirror_target();  // the buffer for this table's layout (nullptr: table too big for the direct path)
  void state_idents(unsigned long long* idents) const;  // [1 + aggs]: 0 / INT64_MAX (MIN) / INT64_MIN (MAX)
  TableMode mode_ = TableMode::DENSE;
  // hash table: entries [capacity][entry_words] u64, key tuples [capacity][key_words] u32
  unsigned long long* h_table_ = nullptr;
  uint32_t* h_keys_ = nullptr;
  unsigned long long* h_count_dev_ = nullptr;
  uint64_t h_capacity_ = 0;
  int h_entry_words_ = 0, h_key_words_ = 4, h_key_used_ = 4;  // key tuple stride (multiple of 4 words) and its used prefix (valid mask + padding + columns)
  uint64_t h_groups_bound_ = 0;     // upper bound of occupied slots known on the host
  bool h_bound_stale_ = false;      // rows were scanned since the bound was last read from the device (it is groups + those rows)
  uint64_t h_rows_seen_ = 0;        // rows scanned into the hash table so far (input of the cardinality estimate)

  Context* ctx_ = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events_;
  std::vector<std::pair<const char*, hipEvent_t>> trace_;  // FDB_PROFILE=1: points on the stream's own timeline, printed at the next sync (tuning aid)
  void trace(const char* what);
  std::vector<std::pair<hipEvent_t, hipEvent_t>> merge_events_;
  std::vector<void*> scratch_;  // device blocks in use by in-flight kernels; returned to the context at the next sync
  std::vector<std::unique_ptr<DeviceBatch>> pending_;   // queued small records (copies in flight or done), not scanned yet
  std::vector<std::unique_ptr<DeviceBatch>> inflight_;  // records a launched kernel may still be reading; dropped at the next sync
  int64_t pending_rows_ = 0;
  size_t pending_bytes_ = 0;
  bool jit_possible() const;
  std::string draw_;
};

}  // namespace fdb
