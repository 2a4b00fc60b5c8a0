// fdb_hash.cpp — the high-cardinality table of the fused filter + aggregate chain (cfg 5, int64 keys).
//
// When the key space outgrows the dense mixed-radix table (more than 8 key columns, more than 2^22 slots, or an int64
// key column such as a time bucket) the plan moves its state into a global open-addressing hash table keyed by a
// 128-bit fingerprint of the key tuple (kernels and layout: fdb_kernels.hip / fdb_kernels.h "high-cardinality path").
// This is the device replacement of the reference's `map[uint64]hashtuple` + per-group builders
// (aggregate.go:130, :398-486), which allocates one builder per aggregation per new group.
//
// Sizing is conservative instead of transactional: a scan is cut into chunks of ≤ kChunkRows rows and, before each
// chunk, the table is grown (device-side re-hash) so that capacity ≥ 2 × (groups so far + rows in the chunk). An insert
// can therefore never fail and no row is ever applied twice.
#include "fdb_context.h"
#include "fdb_plan_internal.h"
#include "fdb_jit.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <thread>

namespace fdb {

namespace {
constexpr int64_t kChunkRows = 4 << 20;   // a decent launch: what the table makes room for when there is no estimate
constexpr int64_t kFirstChunkRows = 1 << 20;  // the first chunk of a fresh table: just enough rows to estimate the cardinality from (what
                                              // it inserts is re-hashed once, when the table takes its final size)
inline uint64_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }
}  // namespace

// Key-tuple words: [valid mask lo, valid mask hi, 2 words of padding, then 1 word per dictionary column / 2 per int64 column],
// rounded up to a multiple of 4 words: tuples and the first column start on 16-byte boundaries, so the inserting lane writes a
// tuple with 16-byte stores (9 for cfg 5's 32 label columns instead of 34 four-byte ones — every store of a scattered tuple is
// its own write request, and those were most of what an insert cost).
void Plan::hash_layout() {
  int kw = 4;
  for (GroupColState& g : gcols_) { g.word = kw; kw += g.kind == 0 ? 1 : 2; }
  const int used = kw;
  kw = (kw + 3) & ~3;
  const int ew = (int)((3 + aggs_.size() + 3) / 4 * 4);
  if (h_table_ == nullptr) { h_key_words_ = kw; h_key_used_ = used; h_entry_words_ = ew; return; }
  if (kw != h_key_words_) {  // columns were added: widen the key store (same capacity, same fingerprints)
    uint32_t* nk = (uint32_t*)ctx_->dev_alloc((size_t)h_capacity_ * kw * 4);  // (no initialisation needed, see fdb_launch_hash_rehash)
    unsigned long long* nt = (unsigned long long*)ctx_->dev_alloc((size_t)h_capacity_ * ew * 8);
    unsigned long long idents[FDB_MAX_AGGS] = {0};
    for (size_t j = 0; j < aggs_.size(); j++)
      idents[j] = aggs_[j].func == FDB_AGG_MIN ? (unsigned long long)FDB_I64_MAX : aggs_[j].func == FDB_AGG_MAX ? (unsigned long long)FDB_I64_MIN : 0ull;
    hip_check(fdb_launch_hash_init(nt, h_capacity_, ew, (int)aggs_.size(), idents, stream_), "hash init");
    hip_check(fdb_launch_hash_rehash(h_table_, h_keys_, h_capacity_, h_key_words_, h_key_used_, nt, nk, h_capacity_ - 1, ew, kw, stream_), "hash rehash");
    hip_check(hipStreamSynchronize(stream_), "sync(rehash)");
    ctx_->dev_free(h_table_); ctx_->dev_free(h_keys_);
    h_table_ = nt; h_keys_ = nk; h_key_words_ = kw;
  }
  h_key_used_ = used;
}

uint64_t Plan::hash_groups() {
  unsigned long long n = 0;
  hip_check(hipMemcpyAsync(&n, h_count_dev_, 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(group count)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  h_groups_bound_ = n;
  return n;
}

void Plan::hash_reserve(uint64_t extra, uint64_t expected_groups) {
  // capacity ≥ 2 × (groups + extra) is what makes an insert unable to fail; `expected_groups` (the cardinality estimate of
  // push_hash) only ever raises the target, so that a table that is still filling up is grown ONCE to its final size instead of
  // doubling its way there (cfg 5: 15 re-hashes per scan became 1)
  uint64_t need = next_pow2(std::max<uint64_t>(2 * (std::max(h_groups_bound_, expected_groups) + extra), 1 << 16));
  if (h_table_ != nullptr && need <= h_capacity_) return;
  if (h_table_ != nullptr && h_groups_bound_ > 0 && expected_groups == 0) need *= 2;  // no estimate: grow by two steps at once
  const int ew = h_entry_words_, kw = h_key_words_;
  unsigned long long* nt = (unsigned long long*)ctx_->dev_alloc((size_t)need * ew * 8);
  uint32_t* nk = (uint32_t*)ctx_->dev_alloc((size_t)need * kw * 4);  // (no initialisation needed, see fdb_launch_hash_rehash)
  unsigned long long idents[FDB_MAX_AGGS] = {0};
  for (size_t j = 0; j < aggs_.size(); j++)
    idents[j] = aggs_[j].func == FDB_AGG_MIN ? (unsigned long long)FDB_I64_MAX : aggs_[j].func == FDB_AGG_MAX ? (unsigned long long)FDB_I64_MIN : 0ull;
  hip_check(fdb_launch_hash_init(nt, need, ew, (int)aggs_.size(), idents, stream_), "hash init");
  if (h_table_ != nullptr) {
    hip_check(fdb_launch_hash_rehash(h_table_, h_keys_, h_capacity_, kw, h_key_used_, nt, nk, need - 1, ew, kw, stream_), "hash rehash");
    hip_check(hipStreamSynchronize(stream_), "sync(rehash)");
    ctx_->dev_free(h_table_); ctx_->dev_free(h_keys_);
  }
  if (h_count_dev_ == nullptr) {
    h_count_dev_ = (unsigned long long*)ctx_->dev_alloc(256);
    hip_check(hipMemsetAsync(h_count_dev_, 0, 256, stream_), "hipMemsetAsync(counter)");
  }
  h_table_ = nt; h_keys_ = nk; h_capacity_ = need;
}

// Groups the scan will end with, estimated from what it has seen: `d` distinct groups after `n` rows. Under a uniform draw from G
// groups d = G·(1 − e^(−n/G)); G is found by bisection and pushed forward to n_total rows. A skewed key distribution makes this
// an UNDER-estimate (then the table simply grows again, the ≥ 2 × invariant is kept by hash_reserve regardless); nearly-all-new
// rows (d ≥ 0.98 n) mean "no saturation in sight": every remaining row may be a new group.
static uint64_t estimate_final_groups(uint64_t d, uint64_t n, uint64_t n_total) {
  if (d == 0 || n == 0 || n_total <= n) return d;
  if ((double)d >= 0.98 * (double)n) return d + (n_total - n);
  auto f = [&](double G) { return G * (1.0 - std::exp(-(double)n / G)); };
  double lo = (double)d, hi = (double)d * 1e4;
  for (int it = 0; it < 60; it++) { const double mid = std::sqrt(lo * hi); if (f(mid) < (double)d) lo = mid; else hi = mid; }
  const double G = hi;
  return (uint64_t)std::min((double)(d + (n_total - n)), G * (1.0 - std::exp(-(double)n_total / G)) + 1.0);
}

// Inserts pre-aggregated entries ({count, acc…} + key tuples of `in_kw` words, described by `cols`) into the table.
void Plan::hash_insert_entries(const std::vector<unsigned long long>& entries, const std::vector<uint32_t>& keys, int64_t n, int in_kw,
                               const std::vector<FdbHashCol>& cols) {
  if (n == 0) return;
  if (h_count_dev_ != nullptr) hash_groups();  // refresh the bound (waits for the stream)
  hash_reserve((uint64_t)n);
  void* d_entries = ctx_->dev_alloc(entries.size() * 8);
  void* d_keys = ctx_->dev_alloc(keys.size() * 4);
  scratch_.push_back(d_entries); scratch_.push_back(d_keys);
  hip_check(hipMemcpyAsync(d_entries, entries.data(), entries.size() * 8, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(entries)");
  hip_check(hipMemcpyAsync(d_keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(keys)");
  hash_merge_device((const unsigned long long*)d_entries, (const uint32_t*)d_keys, n, in_kw, cols, /*unique_source=*/true);  // (both callers compact ONE table)
  hip_check(hipStreamSynchronize(stream_), "sync(hash merge)");  // the host vectors must outlive the copies
}

// What every merge launch shares: the destination side of the argument block and the reducers.
void Plan::hash_merge_args(FdbHashMergeArgs* m, const std::vector<FdbHashCol>& cols, int in_stride_words) {
  std::memset(m, 0, sizeof(*m));
  m->table = h_table_; m->keys = h_keys_; m->n_groups = h_count_dev_; m->mask = h_capacity_ - 1;
  static const FdbHashCol kNoCol = {};
  m->cols = (const FdbHashCol*)(cols.empty() ? upload(&kNoCol, sizeof(FdbHashCol)) : upload(cols.data(), cols.size() * sizeof(FdbHashCol)));
  m->n_cols = (int)gcols_.size(); m->entry_words = h_entry_words_; m->key_words = h_key_words_;
  m->n_aggs = (int)aggs_.size();
  for (size_t j = 0; j < aggs_.size(); j++) {
    const int32_t f = aggs_[j].func;
    m->funcs[j] = f == FDB_AGG_COUNT ? (final_stage_ ? 1 : 0) : f == FDB_AGG_SUM ? (aggs_[j].type == FDB_T_F64 ? 2 : 1) : f == FDB_AGG_MIN ? 3 : 4;
  }
  // how much of an incoming tuple the columns reach, and whether it already has our layout
  int reach = 2;
  bool same = true;
  for (size_t c = 0; c < gcols_.size(); c++) {
    if (cols[c].src_word != cols[c].word) same = false;
    if (cols[c].src_word >= 0) reach = std::max(reach, cols[c].src_word + (cols[c].kind == 0 ? 1 : 2));
  }
  m->in_words = std::min((reach + 3) & ~3, in_stride_words);
  m->same_layout = same && m->in_words <= h_key_words_;
}

// The merge launch itself: `n` pre-aggregated entries ({count, acc…} + key tuples of `in_kw` words) already on the device. The
// table must have room (hash_reserve).
void Plan::hash_merge_device(const unsigned long long* d_entries, const uint32_t* d_keys, int64_t n, int in_kw, const std::vector<FdbHashCol>& cols, bool unique_source) {
  FdbHashMergeArgs m;
  hash_merge_args(&m, cols, in_kw);
  m.entries = d_entries; m.in_keys = d_keys; m.n = n;
  m.in_key_words = in_kw; m.in_entry_words = (int)(1 + aggs_.size());
  m.unique_source = unique_source;
  hip_check(fdb_launch_hash_merge(m, device_, stream_), "hash merge");
  state_dirty_ = true;
}

// Dense → hash migration: the occupied dense slots become pre-aggregated entries keyed by their per-column key ids.
void Plan::switch_to_hash() {
  CompactState cs;
  const bool had_state = state_dirty_ && d_state_ != nullptr;
  if (had_state) fetch_compact(&cs); else sync();
  mode_ = TableMode::HASH;
  hash_layout();
  h_groups_bound_ = 0;
  hash_reserve((uint64_t)std::max<int64_t>(cs.n, 1));
  if (had_state && cs.n > 0) {
    const int in_kw = (int)(2 + gcols_.size());  // dense tables only hold dictionary columns: one word each
    std::vector<uint32_t> keys((size_t)cs.n * in_kw, 0);
    std::vector<unsigned long long> entries((size_t)cs.n * (1 + aggs_.size()));
    for (int64_t i = 0; i < cs.n; i++) {
      for (size_t c = 0; c < gcols_.size(); c++) keys[(size_t)i * in_kw + 2 + c] = cs.ids[c].empty() ? 0u : cs.ids[c][(size_t)i];
      entries[(size_t)i * (1 + aggs_.size())] = cs.cnt[(size_t)i];
      for (size_t j = 0; j < aggs_.size(); j++) entries[(size_t)i * (1 + aggs_.size()) + 1 + j] = cs.acc[j][(size_t)i];
    }
    std::vector<FdbHashCol> cols(gcols_.size());
    for (size_t c = 0; c < gcols_.size(); c++) {
      std::memset(&cols[c], 0, sizeof(FdbHashCol));
      cols[c].kind = gcols_[c].kind; cols[c].word = gcols_[c].word; cols[c].gi = (int)c; cols[c].lut_lds = FDB_NO_LDS;
      cols[c].k1 = fdb_fp_k1((int)c); cols[c].k2 = fdb_fp_k2((int)c);
      cols[c].src_word = gcols_[c].kind == 0 ? (int)(2 + c) : -1;  // a dense table never held int64 key columns
    }
    // COUNT accumulators of a dense table live in its count array: carry them as counts (funcs → skip) — the entry's
    // first word already is the count.
    hash_insert_entries(entries, keys, cs.n, in_kw, cols);
  }
  ctx_->dev_free(d_state_);
  d_state_ = nullptr; d_cnt_ = nullptr;
  state_virgin_ = false;
  mirror_valid_ = false;
  for (AggState& a : aggs_) a.d_acc = nullptr;
  slots_alloc_ = 0; n_slots_ = 1;
}

void Plan::push_hash(const DeviceBatch* const* bs, std::vector<Resolved>& Rs, const std::vector<int>& live, bool runs) {
  hash_layout();
  uint64_t rows_left_total = 0;  // rows this call will still put into the table (what the cardinality estimate extrapolates to)
  for (int i : live) rows_left_total += (uint64_t)bs[i]->rows;
  for (int i : live) {
    Resolved& R = Rs[(size_t)i];
    const DeviceBatch& b = *bs[i];
    FdbHashArgs h;
    std::memset(&h, 0, sizeof(h));
    h.base = R.args;
    FdbScanArgs& a = h.base;
    a.need_count = 0;  // the row count of an entry is only read back for COUNT aggregations (occupancy is "fingerprint ≠ 0")
    for (size_t j = 0; j < aggs_.size(); j++) if (aggs_[j].func == FDB_AGG_COUNT && !final_stage_) a.need_count = 1;
    a.ablate = ablate;
    // columns read by computed aggregate inputs / keys go into base.l8 (the hash scan has no other use for the slot pools)
    bool has_expr = a.n_expr > 0;
    a.n_c4 = a.n_c8 = a.n_l4 = a.n_l8 = 0;
    {
      int pool_col[FDB_ARG_L8];
      for (int k = 0; k < a.n_expr; k++) {
        if (a.expr[k].kind != 0) continue;
        const int ci = R.expr_col[k];
        int slot = -1;
        for (int x = 0; x < a.n_l8; x++) if (pool_col[x] == ci) slot = x;
        if (slot < 0) {
          if (a.n_l8 >= FDB_ARG_L8) throw Error(FDB_ERR_UNSUPPORTED, "computed columns read more than 4 distinct stored columns");
          slot = a.n_l8++;
          pool_col[slot] = ci;
          a.l8[slot].values = b.cols[(size_t)ci].d_values;
          a.l8[slot].validity = b.cols[(size_t)ci].d_validity;
        }
        a.expr[k].slot = slot;
      }
    }
    // LUTs: predicate LUTs from the record's blob, key-id LUTs appended
    std::vector<FdbHashCol> hcols(R.groups.size());
    std::vector<size_t> lut_off(R.groups.size(), 0);
    std::vector<char> lut_identity(R.groups.size(), 0);
    for (size_t g = 0; g < R.groups.size(); g++) {
      const GroupRes& gr = R.groups[g];
      FdbHashCol& C = hcols[g];
      std::memset(&C, 0, sizeof(C));
      C.kind = gr.kind; C.gi = gr.gi; C.word = gcols_[(size_t)gr.gi].word;
      C.lut_lds = FDB_NO_LDS; C.src_word = -1;
      C.k1 = fdb_fp_k1(gr.gi); C.k2 = fdb_fp_k2(gr.gi);
      if (gr.kind == 2) { C.src_word = gr.expr_root; has_expr = true; continue; }  // computed int64 key: no stored column
      const DevColumn& c = b.cols[(size_t)gr.ci];
      C.values = c.d_values; C.validity = c.d_validity;
      if (gr.kind == 0) {
        // the record's dictionary IS the plan's value list of the column, in order (the usual case: the parts of a table share their
        // dictionaries, and the plan interned the first one it saw entry by entry): key id = index + 1 — no table to ship or to read.
        // ($FDB_NO_IDENTITY_LUT: A/B aid)
        const std::vector<uint32_t>& L = *gr.lut;
        bool identity = !L.empty() && L.front() == 1u && L.back() == (uint32_t)L.size() && !knobs_.no_identity_lut;
        for (size_t i = 0; identity && i < L.size(); i++) identity = L[i] == (uint32_t)i + 1u;
        C.lut_len = (uint32_t)L.size();
        if (identity) lut_identity[g] = 1; else lut_off[g] = R.blob.add(L.data(), L.size() * 4);
      }
    }
    // canonical: the record carries every group column of the plan, in the plan's order — then column c's word is a
    // compile-time constant of the specialised kernel (4 + the widths before it) and aligned quads of dictionary columns are
    // written with one 16-byte store
    bool canonical = hcols.size() == gcols_.size();
    { int w = 4; for (size_t g = 0; g < hcols.size() && canonical; g++) { canonical = hcols[g].word == w; w += hcols[g].kind == 0 ? 1 : 2; } }
    h.canonical = canonical ? 1 : 0;
    unsigned char* d_blob = R.blob.bytes.empty() ? nullptr : (unsigned char*)upload(R.blob.bytes.data(), R.blob.bytes.size());
    size_t lds_off = 0;
    for (const PendingLut& p : R.luts) {  // predicate LUTs (kind 0 only: dense group LUTs were never added in this mode)
      const bool in_lds = p.len_bytes <= 16384 && lds_off + p.len_bytes <= 32768;
      uint32_t lds = FDB_NO_LDS;
      if (in_lds) { lds = (uint32_t)lds_off; lds_off = align_up_sz(lds_off + p.len_bytes, 16); }
      a.leaves[p.index].lut = d_blob + p.blob_off;
      a.leaves[p.index].lut_lds = lds;
    }
    for (size_t g = 0; g < hcols.size(); g++) {
      if (hcols[g].kind != 0 || lut_identity[g]) continue;  // (identity: lut stays nullptr)
      const size_t bytes = (size_t)hcols[g].lut_len * 4;
      hcols[g].lut = (const uint32_t*)(d_blob + lut_off[g]);
      if (bytes <= 8192 && lds_off + bytes <= 60 * 1024) { hcols[g].lut_lds = (uint32_t)lds_off; lds_off = align_up_sz(lds_off + bytes, 16); }
    }
    a.lds_lut_bytes = (uint32_t)align_up_sz(lds_off, 16);
    h.hcols = (const FdbHashCol*)upload(hcols.data(), std::max<size_t>(hcols.size(), 1) * sizeof(FdbHashCol));
    h.n_hcols = (int)hcols.size();
    h.key_words = h_key_words_;
    h.entry_words = h_entry_words_;
    const int grid = fdb_scan_default_grid(device_) * 4;  // 256-thread workgroups, 8 per CU
    // The run-time specialised kernel for this record's shape (fdb_jit.cpp), or the interpreting scan_hash_kernel
    hipFunction_t jit_fn = nullptr;
    int jit_grid = 0;
    if (runs) {  // (the shape is that of a runs launch; the real pointers follow below)
      h.runs.tuples = (unsigned char*)(uintptr_t)1;
      const int fmt = runs_format(R);
      h.runs.run_words = fmt == 0 ? 0 : fmt == 1 ? FDB_RUN_MEDIUM_WORDS : h_key_words_ + 4;
      h.runs.stage_cap = fmt == 0 ? FDB_RUN_STAGE : fmt == 1 ? FDB_RUN_WAVE_LDS / FDB_RUN_MEDIUM_BYTES : FDB_RUN_WAVE_LDS / (h.runs.run_words * 4);
    }
    const size_t run_lds = runs ? (size_t)4 * FDB_RUN_WAVE_LDS : 0;
    if (sub_tiles != 4 && a.lds_lut_bytes <= FDB_LDS_BUDGET) {
      JitHashShape shape = jit_hash_shape(h, hcols.data());
      shape.ablate = runs ? 0 : ablate & 3;
      JitDeferScope defer(!runs && a.n_expr == 0);  // ($FDB_JIT_ASYNC=1: the probing scan can be interpreted while its kernel is built; a run store cannot)
      jit_fn = jit_hash_get(shape);
      if (jit_fn != nullptr)
        jit_grid = grid_override > 0 ? grid_override : (fdb_scan_default_grid(device_) / 2) * std::min(4, jit_blocks_per_cu(jit_fn, 256, a.lds_lut_bytes + run_lds));
      if (std::getenv("FDB_JIT_DEBUG")) std::fprintf(stderr, "[frostdb_amd] hash kernel%s: %d workgroups (%d fit a CU with %zu B of LDS)\n", runs ? " (runs)" : "", jit_grid,
                                                   jit_fn ? jit_blocks_per_cu(jit_fn, 256, a.lds_lut_bytes + run_lds) : 0, (size_t)a.lds_lut_bytes + run_lds);
    }
    if (runs && jit_fn == nullptr) {  // no specialised kernel after all (hiprtc failed): what was collected goes into the table, this record and the rest take the probing path
      runs_to_table();
      if (mode_ == TableMode::DENSE) switch_to_hash();
      runs = false;
      std::memset(&h.runs, 0, sizeof(h.runs));
      h.key_words = h_key_words_; h.entry_words = h_entry_words_;
    }
    if (runs) {
      // ---- table-free: ONE launch over the whole record; its runs land in a segment sized for the worst case (every row a run) ----
      if (runs_.size() >= FDB_MAX_RUN_SEGMENTS) throw Error(FDB_ERR_STATE, "internal: run segments exhausted");  // (runs_wanted keeps this from happening)
      const int64_t n_tiles = (b.rows + 1023) / 1024;
      const int64_t launch_grid = std::min<int64_t>(jit_grid, n_tiles);
      RunSegment seg;
      seg.n_entries = n_tiles * 4;
      const int64_t n_chunks = b.rows / (FDB_RUN_CHUNK - 256) + launch_grid * 4 + 2;  // a wave abandons < 256 slots when it changes chunks and keeps one chunk open
      seg.capacity = n_chunks * FDB_RUN_CHUNK;
      seg.run_words = h.runs.run_words;
      const size_t tuples_bytes = align_up_sz((size_t)seg.capacity * (seg.run_words == 0 ? (size_t)FDB_RUN_BYTES : seg.run_words == FDB_RUN_MEDIUM_WORDS ? (size_t)FDB_RUN_MEDIUM_BYTES : (size_t)seg.run_words * 4), 256);
      const size_t dir_bytes = align_up_sz((size_t)seg.n_entries * 8, 256);
      seg.block = ctx_->dev_alloc(tuples_bytes + dir_bytes + 256);
      unsigned char* base = (unsigned char*)seg.block;
      seg.tuples = base;
      seg.dir = (uint32_t*)(base + tuples_bytes); seg.cursor = (uint32_t*)(base + tuples_bytes + dir_bytes);
      runs_.push_back(seg);
      hip_check(hipMemsetAsync(seg.dir, 0, dir_bytes + 256, stream_), "hipMemsetAsync(run directory)");
      h.runs.tuples = seg.tuples; h.runs.dir = seg.dir; h.runs.chunk_cursor = seg.cursor;
      h.table = nullptr; h.keys = nullptr; h.n_groups = nullptr; h.mask = 0;
      h.row_begin = 0; h.row_end = b.rows;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
      hip_check(jit_hash_launch(jit_fn, h, (int)launch_grid, a.lds_lut_bytes + run_lds, stream_), "run scan launch");
      last_kernel_ = seg.run_words == 0 ? "fdb_hash_kernel(runs)" : seg.run_words == FDB_RUN_MEDIUM_WORDS ? "fdb_hash_kernel(runs, medium)" : "fdb_hash_kernel(runs, wide)";
      if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); pending_events_.emplace_back(e0, e1); }
      state_dirty_ = true;
      stat_launches += 1;
      stat_bytes += R.bytes;
      stat_rows += b.rows;
      continue;
    }
    for (int64_t r0 = 0; r0 < b.rows;) {
      // How many rows may go into the table before the next look at its group count: capacity / 2 − groups (every one of them
      // could be a new group). The first chunk of a fresh table is small; from what it finds the final cardinality is
      // estimated, the table grown ONCE to hold it, and the rest of the scan runs in a few big chunks.
      const uint64_t left_here = (uint64_t)(b.rows - r0);
      const uint64_t min_chunk = std::min<uint64_t>(left_here, (uint64_t)(h_rows_seen_ == 0 && h_groups_bound_ == 0 ? kFirstChunkRows : kChunkRows));
      auto room_now = [&]() -> uint64_t { return h_table_ != nullptr && h_capacity_ / 2 > h_groups_bound_ ? h_capacity_ / 2 - h_groups_bound_ : 0; };
      uint64_t room = room_now();  // with the pessimistic bound (groups at the last look + every row scanned since): no wait
      if (room < min_chunk && h_table_ != nullptr && h_bound_stale_) {
        hash_groups();  // the exact count (waits for the previous chunk)
        h_bound_stale_ = false;
        room = room_now();
      }
      if (room < min_chunk) {
        uint64_t expected = 0;
        static const bool no_estimate = std::getenv("FDB_HASH_NO_ESTIMATE") != nullptr;  // (tuning / debugging aid)
        if (!no_estimate && !h_bound_stale_ && h_rows_seen_ >= (uint64_t)(1 << 20) && h_groups_bound_ > 0) {
          expected = estimate_final_groups(h_groups_bound_, h_rows_seen_, h_rows_seen_ + rows_left_total);
          expected += expected / 4;  // head-room for the estimate's error
          const uint64_t bytes_per_slot = (uint64_t)h_entry_words_ * 8 + (uint64_t)h_key_words_ * 4;
          if (next_pow2(2 * (expected + (uint64_t)kChunkRows)) * bytes_per_slot > ((uint64_t)64 << 30)) expected = 0;  // (beyond a sane budget: grow step by step)
        }
        // room for a decent next chunk on top of the expected groups: 1/8 of what is left, 4 M … 32 M rows
        const uint64_t want_chunk = expected == 0 ? min_chunk
                                                  : std::min<uint64_t>(left_here, std::max<uint64_t>((uint64_t)kChunkRows, std::min<uint64_t>(rows_left_total / 8, (uint64_t)32 << 20)));
        hash_reserve(want_chunk, expected);
        room = room_now();
      }
      static const char* cap_env = std::getenv("FDB_HASH_MAX_CHUNK");  // (tuning / debugging aid)
      if (cap_env != nullptr && std::atoll(cap_env) > 0) room = std::min<uint64_t>(room, (uint64_t)std::atoll(cap_env));
      // chunk boundaries inside a record stay on tile boundaries (1 024 rows): the kernels address 4-row lane groups with 16-byte
      // loads and read validity bitmaps bytewise from the chunk's first row
      if (room < left_here) room &= ~(uint64_t)1023;
      const int64_t r1 = r0 + (int64_t)std::min<uint64_t>(left_here, room);
      if (std::getenv("FDB_PROFILE")) std::fprintf(stderr, "[fdb] hash chunk rows [%lld, %lld) capacity %llu bound %llu seen %llu\n", (long long)r0, (long long)r1,
                                                  (unsigned long long)h_capacity_, (unsigned long long)h_groups_bound_, (unsigned long long)h_rows_seen_);
      h.table = h_table_; h.keys = h_keys_; h.n_groups = h_count_dev_; h.mask = h_capacity_ - 1;
      h.row_begin = r0; h.row_end = r1;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
      if (jit_fn == nullptr && has_expr)
        throw Error(FDB_ERR_UNSUPPORTED, "computed (projected) columns need the run-time specialised kernel (hiprtc unavailable or disabled)");
      if (jit_fn != nullptr) {
        const int64_t n_tiles = (r1 - r0 + 1023) / 1024;
        hip_check(jit_hash_launch(jit_fn, h, (int)std::min<int64_t>(jit_grid, n_tiles), a.lds_lut_bytes, stream_), "hash scan launch");
        last_kernel_ = "fdb_hash_kernel";
      } else {
        hip_check(fdb_launch_scan_hash(h, grid, a.lds_lut_bytes, stream_), "hash scan launch");
        last_kernel_ = "scan_hash_kernel";
      }
      if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); pending_events_.emplace_back(e0, e1); }
      if (std::getenv("FDB_PROFILE_CHUNKS")) {  // tuning aid: per-launch time (serialises the scan)
        const auto t0 = std::chrono::steady_clock::now();
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        std::fprintf(stderr, "[fdb] hash chunk of %lld rows: %.1f us\n", (long long)(r1 - r0), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      h_groups_bound_ += (uint64_t)(r1 - r0);
      h_bound_stale_ = true;
      h_rows_seen_ += (uint64_t)(r1 - r0);
      rows_left_total -= (uint64_t)(r1 - r0);
      state_dirty_ = true;
      stat_launches += 1;
      r0 = r1;
    }
    stat_bytes += R.bytes;
    stat_rows += b.rows;
  }
}

void Plan::fetch_compact_hash(CompactState* cs) {
  hip_check(hipSetDevice(device_), "hipSetDevice");
  if (h_table_ == nullptr) { sync(); return; }
  const uint64_t n = hash_groups();
  cs->n = (int64_t)n;
  if (n == 0) { sync(); return; }
  const int ew = h_entry_words_, kw = h_key_words_, oew = ew - 2;
  unsigned long long* d_entries = (unsigned long long*)ctx_->dev_alloc((size_t)n * oew * 8);
  uint32_t* d_keys = (uint32_t*)ctx_->dev_alloc((size_t)n * kw * 4);
  unsigned long long* d_n = (unsigned long long*)ctx_->dev_alloc(256);
  const size_t n_chunks = (size_t)((h_capacity_ + 63) / 64);
  uint32_t* d_bases = (uint32_t*)ctx_->dev_alloc((n_chunks + 4 + n_chunks / 1024 + 8) * 4);  // (+ the scan's per-1024 sums)
  // slot order: repeated calls on an unchanged table (partial_keys, then partial_state per aggregation) line up row by row
  hip_check(fdb_launch_hash_chunk_bases(h_table_, h_capacity_, ew, d_bases, d_bases + n_chunks + 4, d_n, stream_), "hash chunk bases");
  hip_check(fdb_launch_hash_compact(h_table_, h_keys_, h_capacity_, ew, kw, d_entries, d_keys, d_bases, stream_), "hash compact");
  // pinned staging (cached by the context): pageable destinations would cap the copy at a few GB/s
  unsigned long long* entries = (unsigned long long*)ctx_->host_alloc((size_t)n * oew * 8);
  uint32_t* keys = (uint32_t*)ctx_->host_alloc((size_t)n * kw * 4);
  hip_check(hipMemcpyAsync(entries, d_entries, (size_t)n * oew * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(entries)");
  hip_check(hipMemcpyAsync(keys, d_keys, (size_t)n * kw * 4, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(keys)");
  sync();
  ctx_->dev_free(d_entries); ctx_->dev_free(d_keys); ctx_->dev_free(d_n); ctx_->dev_free(d_bases);
  cs->cnt.resize(n);
  for (size_t j = 0; j < aggs_.size(); j++) cs->acc[j].resize(n);
  for (size_t c = 0; c < gcols_.size(); c++) {
    if (gcols_[c].kind == 0) cs->ids[c].resize(n);
    else { cs->ivals[c].resize(n); cs->ivalid[c].resize(n); }
  }
  const size_t n_aggs = aggs_.size(), n_cols = gcols_.size();
  for (uint64_t i = 0; i < n; i++) {  // one pass, row-major over the compacted entries
    const unsigned long long* e = entries + i * oew;
    cs->cnt[i] = e[0];
    for (size_t j = 0; j < n_aggs; j++) cs->acc[j][i] = e[1 + j];
    const uint32_t* k = keys + i * kw;
    const uint64_t vm = (uint64_t)k[0] | ((uint64_t)k[1] << 32);
    for (size_t c = 0; c < n_cols; c++) {
      const GroupColState& g = gcols_[c];
      if (g.kind == 0) cs->ids[c][i] = k[g.word];
      else { cs->ivalid[c][i] = (vm >> c) & 1; cs->ivals[c][i] = (int64_t)((uint64_t)k[g.word] | ((uint64_t)k[g.word + 1] << 32)); }
    }
  }
  ctx_->host_free(entries); ctx_->host_free(keys);
}

// Finish on the device: occupied entries → column buffers (dictionary indices at their transport width, int64 keys, validity
// bitmaps, aggregate columns) → device-to-host copies → Arrow buffers. PCIe is the narrowest link of a big Finish (cfg 5:
// 10 M groups × 32 label columns = 1.28 GB of uint32 indices at ≈56 GB/s), so indices of small dictionaries cross it as uint8 /
// uint16, in slices of 2^20 rows, and host threads widen slice k into the record's uint32 buffers while slice k + 1 is in flight.
// Returns the number of groups.
void widen_indices(const void* src, int width, uint32_t* dst, size_t n);  // fdb_widen.cc
void widen_indices_mapped(const void* src, int width, const uint32_t* table, size_t table_len, uint32_t* dst, size_t n);  // … through a rank → index table

namespace {
// CPUs the container may use at a time (cgroup v2 cpu.max / v1 cfs quota), 0 = no limit known. The bench boxes show 256 CPUs and grant 16:
// widening with 48 or 64 threads there ran a Finish into the quota (8.2 → 13 / 15.5 ms per cfg 5 sorted step), 16 threads do what 32 do.
unsigned cpu_quota() {
  static const unsigned q = [] {
    double quota = 0, period = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char a[32] = {0};
      if (std::fscanf(f, "%31s %lf", a, &period) == 2 && std::strcmp(a, "max") != 0) quota = std::atof(a);
      std::fclose(f);
    } else {
      if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lf", &quota) != 1) quota = 0; std::fclose(g); }
      if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lf", &period) != 1) period = 0; std::fclose(g); }
    }
    return quota > 0 && period > 0 ? (unsigned)std::max(1.0, quota / period + 0.5) : 0u;
  }();
  return q;
}
int host_threads_for(size_t elements) {
  if (elements < ((size_t)4 << 20)) return 0;  // small results: widened inline
  if (const char* e = std::getenv("FDB_HOST_THREADS")) return std::max(0, std::min(64, std::atoi(e)));
  const unsigned hw = std::thread::hardware_concurrency();
  unsigned n = std::max(1u, std::min(32u, hw / 4));
  if (const unsigned q = cpu_quota(); q != 0) n = std::min(n, std::max(4u, q));
  return (int)n;
}
}  // namespace

// `resident` (fdb_plan_finish_batch): the result STAYS in HBM — the columns are written once, at the width a resident record uses
// (uint32 indices), into an arena the batch owns; nothing crosses PCIe but the per-column NULL counts.
int64_t Plan::finish_columns_hash(std::vector<OutColumn>* out, DeviceBatch* resident, const RunsView* runs) {
  hip_check(hipSetDevice(device_), "hipSetDevice");
  PhaseTimer pt;
  // (`runs`: the groups come from an ordered plan's run store instead of the table — pass 1 below is runs_expand, everything
  // after it — the column pass, transport widths, slices, widening, the resident form — is shared)
  const uint64_t n = runs != nullptr ? (uint64_t)runs->n_groups : fresh_groups_ >= 0 ? (uint64_t)fresh_groups_ : hash_groups();
  fresh_groups_ = -1;
  if (pt.on) pt.mark("finish: group count");
  const size_t n_cols = gcols_.size(), n_vals = 1 + aggs_.size();
  const size_t np = (size_t)((n + 63) & ~(uint64_t)63) + 64;  // padded row count of the buffers
  int kSliceShift = 6;  // 2^20 rows per slice; a smaller result is one slice of the next power of two
  while (kSliceShift < knobs_.finish_slice_shift && ((uint64_t)1 << kSliceShift) < n) kSliceShift++;
  const size_t kSliceRows = (size_t)1 << kSliceShift;
  const size_t n_slices = (size_t)((n + kSliceRows - 1) >> kSliceShift);
  // transport width per group column: 1 / 2 bytes for dictionaries that fit, else the column's own width
  std::vector<int> width(n_cols);
  std::vector<bool> narrow(n_cols);
  size_t n_narrow = 0, last_narrow = 0;
  for (size_t c = 0; c < n_cols; c++) {
    const GroupColState& g = gcols_[c];
    // (negative: BITS per index, packed low bits first — a dictionary of ≤ 4 / ≤ 16 entries travels as 2 / 4 bits per row)
    width[c] = g.kind != 0 ? 8 : resident != nullptr ? 4 : g.values.size() <= 4 ? -2 : g.values.size() <= 16 ? -4 : g.values.size() <= 256 ? 1 : g.values.size() <= 65536 ? 2 : 4;
    narrow[c] = g.kind == 0 && width[c] < 4;
    if (narrow[c]) { n_narrow++; last_narrow = c; }
  }
  // Host block (pinned, handed to the caller as the record's buffers): [uint32 arrays of the narrow columns — written by the
  // widening threads][direct part: wide key columns, one validity bitmap per column, one 8-byte column per value array].
  // Device block: [direct part, same layout][narrow columns, slice-major: slice s = rows [s·2^20, (s+1)·2^20) of every narrow
  // column back to back]; row masks on the side.
  std::vector<size_t> off_key(n_cols), off_bits(n_cols), off_val(n_vals), off_narrow(n_cols, 0);
  size_t total = 0;
  auto place = [&](size_t bytes) { const size_t o = total; total = align_up_sz(total + bytes, 256); return o; };
  for (size_t c = 0; c < n_cols; c++) if (narrow[c]) off_key[c] = place(np * 4);
  const size_t direct_begin = total;
  for (size_t c = 0; c < n_cols; c++) if (!narrow[c]) off_key[c] = place(np * (size_t)width[c]);
  for (size_t v = 1; v < n_vals; v++) off_val[v] = place(np * 8);
  const size_t off_nulls = place(std::max<size_t>(n_cols, 1) * 8);  // NULLs per group column, counted by the kernel
  // the per-group row counts only cross PCIe when a COUNT aggregation reads them (cfg 5: 80 MB that nobody asked for), and a column's
  // validity bitmap only when the column has a NULL (cfg 5: 12 of 32 label columns have none — known once the kernel's NULL counts
  // are on the host, so the bitmaps go LAST and are copied one by one)
  bool counts_wanted = false;
  for (const AggState& A : aggs_) if (A.func == FDB_AGG_COUNT && !final_stage_) counts_wanted = true;
  const size_t direct_copy_bytes = (counts_wanted ? total + align_up_sz(np * 8, 256) : total) - direct_begin;
  off_val[0] = place(np * 8);
  const size_t bitmap_bytes = np / 8 + 64;
  for (size_t c = 0; c < n_cols; c++) off_bits[c] = place(bitmap_bytes);
  const size_t direct_bytes = total - direct_begin;
  size_t slice_stride = 0, narrow_bytes = 0;  // (laid out once the transport widths are final: after the present-id pass below)
  auto col_bytes = [&](size_t c, size_t rows) { return width[c] > 0 ? rows * (size_t)width[c] : ((rows * (size_t)(-width[c]) + 31) / 32) * 4; };  // (sub-byte columns are written as whole 32-bit words)
  unsigned char* d_block = nullptr;
  std::vector<void*> owned;
  if (resident != nullptr) {  // the batch owns the block (process-wide pool: it outlives this plan)
    d_block = (unsigned char*)device_pool_alloc(device_, direct_bytes + 256);
    resident->arena = d_block;
    resident->arena_bytes = direct_bytes + 256;
    resident->note_reader(stream_);  // (whatever happens below: the block does not go back to the pool while a kernel of ours still writes it)
  } else {
    d_block = (unsigned char*)ctx_->dev_alloc(direct_bytes + 256);
    owned.push_back(d_block);
  }
  unsigned char* d_narrow = nullptr;
  // (device scratch goes back to the context's cache when this function leaves, also by exception — after the Quiesce guard
  // below has waited for both queues)
  struct FreeOwned { Context* c; std::vector<void*>* v; ~FreeOwned() { for (void* p : *v) c->dev_free(p); } } free_owned{ctx_, &owned};
  auto alloc = [&](size_t bytes) { void* p = ctx_->dev_alloc(bytes); owned.push_back(p); return p; };
  std::vector<void*> d_key(std::max<size_t>(n_cols, 1));
  std::vector<uint8_t*> d_bits(std::max<size_t>(n_cols, 1));
  std::vector<unsigned long long*> d_vals(n_vals);
  for (size_t c = 0; c < n_cols; c++) {
    d_key[c] = narrow[c] ? nullptr : d_block + (off_key[c] - direct_begin);  // (narrow columns: set with the narrow layout below)
    d_bits[c] = d_block + (off_bits[c] - direct_begin);
  }
  for (size_t v = 0; v < n_vals; v++) d_vals[v] = (unsigned long long*)(d_block + (off_val[v] - direct_begin));
  unsigned long long* d_n = (unsigned long long*)alloc(256);
  const size_t n_chunks = (size_t)((h_capacity_ + 63) / 64);
  uint32_t* d_bases = (uint32_t*)alloc((n_chunks + 4 + n_chunks / 1024 + 8) * 4);
  std::vector<FdbHashCol> cols(std::max<size_t>(n_cols, 1));
  FdbHashColumnsArgs a;
  std::memset(&a, 0, sizeof(a));
  a.table = h_table_; a.keys = h_keys_; a.capacity = h_capacity_;
  a.out_vals = (unsigned long long* const*)upload(d_vals.data(), n_vals * sizeof(void*));
  a.out_bits = (uint8_t* const*)upload(d_bits.data(), d_bits.size() * sizeof(void*));
  a.bases = d_bases;
  a.slice_shift = kSliceShift;
  a.n_rows = n;
  a.out_nulls = (unsigned long long*)(d_block + (off_nulls - direct_begin));
  a.dense_keys = (uint32_t*)alloc(std::max<size_t>((size_t)n, 1) * (size_t)h_key_words_ * 4 + 256);
  a.n_cols = (int)n_cols; a.entry_words = h_entry_words_; a.key_words = h_key_words_; a.n_vals = (int)n_vals;
  std::shared_ptr<void> backing;
  unsigned char* h_block = nullptr;
  unsigned char* h_narrow = nullptr;  // pinned landing area of the narrow slices
  if (n > 0 && runs != nullptr) {
    hip_check(hipMemsetAsync(a.out_nulls, 0, std::max<size_t>(n_cols, 1) * 8, stream_), "hipMemsetAsync(null counts)");
    FdbRunsExpandArgs x;
    std::memset(&x, 0, sizeof(x));
    x.phys = runs->phys; x.flags = runs->flags; x.out_idx = runs->out_idx; x.n_runs = runs->n_runs;
    x.dense_keys = a.dense_keys; x.vals_cnt = d_vals[0]; x.vals_acc = d_vals[1];
    x.n_cols = (int)n_cols; x.key_words = h_key_words_; x.func = runs_func();
    for (size_t c = 0; c < n_cols && c < FDB_RUN_TUPLE_BYTES; c++) x.col_word[c] = gcols_[c].word;
    // groups made of several runs (cut by a wave or record boundary) are folded with atomics: identity first
    hip_check(hipMemsetAsync(d_vals[0], 0, (size_t)n * 8, stream_), "hipMemsetAsync(counts)");
    hip_check(fdb_launch_fill_u64(d_vals[1], (int64_t)n, x.func == 3 ? (unsigned long long)FDB_I64_MAX : x.func == 4 ? (unsigned long long)FDB_I64_MIN : 0ull, stream_), "fill identity");
    hip_check(fdb_launch_runs_expand(x, runs->segs, stream_), "runs expand");
  } else if (n > 1 && ordered_) {
    // An ordered plan's groups out of the TABLE (chains merged, a consumer that asked for the group count, more runs than the run store's
    // sort takes): pass 1 gathers the occupied entries into scratch rows in slot order, the rows are sorted by the group columns on the
    // device (the run store's sort: sort_by_group_columns) and rows and value arrays are gathered into that order — pass 2 below then
    // writes the columns of an ordered record. (Before: every group to the host, a comparison sort over 32 columns there — 14 s for 10 M groups.)
    uint32_t* sorted_keys = a.dense_keys;
    std::vector<unsigned long long*> tmp_vals(n_vals);
    a.dense_keys = (uint32_t*)alloc((size_t)n * (size_t)h_key_words_ * 4 + 256);
    for (size_t v = 0; v < n_vals; v++) tmp_vals[v] = (unsigned long long*)alloc((size_t)n * 8 + 256);
    unsigned long long* const* final_vals = a.out_vals;
    a.out_vals = (unsigned long long* const*)upload(tmp_vals.data(), n_vals * sizeof(void*));
    hip_check(fdb_launch_hash_chunk_bases(h_table_, h_capacity_, h_entry_words_, d_bases, d_bases + n_chunks + 4, d_n, stream_), "hash chunk bases");
    hip_check(hipMemsetAsync(a.out_nulls, 0, std::max<size_t>(n_cols, 1) * 8, stream_), "hipMemsetAsync(null counts)");
    hip_check(fdb_launch_hash_gather_rows(a, device_, stream_), "hash gather rows");
    unsigned long long* order = (unsigned long long*)alloc((size_t)n * 8);
    hip_check(fdb_launch_iota_u64(order, (int64_t)n, stream_), "iota");
    RunsView tables;
    order = sort_by_group_columns(order, (int64_t)n, nullptr, a.dense_keys, h_key_words_, &tables, &owned);
    hip_check(fdb_launch_gather_rows_u32(a.dense_keys, h_key_words_, order, (int64_t)n, sorted_keys, stream_), "gather key rows");
    for (size_t v = 0; v < n_vals; v++) hip_check(fdb_launch_gather_u64(tmp_vals[v], order, (int64_t)n, d_vals[v], stream_), "gather values");
    a.dense_keys = sorted_keys;
    a.out_vals = final_vals;
    last_kernel_ = "hash_gather_rows_kernel + runs_sort_keys_kernel";  // (what an ordered Finish out of the table ran)
  } else if (n > 0) {
    hip_check(fdb_launch_hash_chunk_bases(h_table_, h_capacity_, h_entry_words_, d_bases, d_bases + n_chunks + 4, d_n, stream_), "hash chunk bases");
    hip_check(hipMemsetAsync(a.out_nulls, 0, std::max<size_t>(n_cols, 1) * 8, stream_), "hipMemsetAsync(null counts)");
    hip_check(fdb_launch_hash_gather_rows(a, device_, stream_), "hash gather rows");
  }
  // ---- transport widths by the ids PRESENT (fdb_kernels.h FdbPresentArgs): a narrow column whose dictionary needs 1 or 2 bytes per
  // index but whose result rows use ≤ 256 / 16 / 4 of its entries ships the rank of the id among the present ones; the host widens
  // through the rank → index table. Worth a pass over the key rows (≈ 0.3 ms per 10 M × 144 B) when it can save tens of MB of PCIe.
  std::vector<std::vector<uint32_t>> present_of(n_cols);  // rank → dictionary index (empty: the column ships id − 1)
  std::vector<const uint32_t*> remap_of(n_cols, nullptr);
  if (resident == nullptr && n > 0 && !knobs_.no_present_ids) {
    std::vector<size_t> cand;
    size_t at_stake = 0;
    for (size_t c = 0; c < n_cols && cand.size() < FDB_MAX_HASH_GCOLS; c++)
      if (narrow[c] && width[c] > 0) { cand.push_back(c); at_stake += (size_t)n * (size_t)width[c]; }
    if (!cand.empty() && at_stake >= (size_t)knobs_.present_ids_min_bytes) {
      FdbPresentArgs pa;
      std::memset(&pa, 0, sizeof(pa));
      pa.dense_keys = a.dense_keys; pa.n_rows = n; pa.key_words = h_key_words_; pa.n_cand = (int)cand.size();
      static const bool no_set = std::getenv("FDB_PRESENT_NO_SET") != nullptr;
      pa.no_wave_set = no_set ? 1 : 0;
      size_t bm_words = 0, remap_words = 0;
      for (size_t k = 0; k < cand.size(); k++) {
        const size_t len = gcols_[cand[k]].values.size();
        pa.word[k] = gcols_[cand[k]].word; pa.dict_len[k] = (uint32_t)len;
        pa.bm_off[k] = (uint32_t)bm_words; pa.remap_off[k] = (uint32_t)remap_words;
        bm_words += (len + 32) / 32; remap_words += len + 1;
      }
      pa.bitmaps = (uint32_t*)alloc(bm_words * 4);
      pa.remap = (uint32_t*)alloc(remap_words * 4);
      pa.present = (uint32_t*)alloc(remap_words * 4);
      pa.counts = (unsigned long long*)alloc(cand.size() * 8 + 256);
      hip_check(hipMemsetAsync(pa.bitmaps, 0, bm_words * 4, stream_), "hipMemsetAsync(present bitmaps)");
      hip_check(hipMemsetAsync(pa.counts, 0, cand.size() * 8, stream_), "hipMemsetAsync(present counts)");  // (live counts of the marking pass)
      hip_check(fdb_launch_present_ids(pa, device_, stream_), "present ids");
      hip_check(fdb_launch_rank_ids(pa, stream_), "rank ids");
      // (into PINNED memory: out of a pageable vector each of these up to 33 small copies made this thread wait for the stream — a
      // round trip apiece, ≈ 0.5 ms per Finish of 32 columns)
      struct PinnedBack { Context* c; void* p; ~PinnedBack() { c->host_free(p); } } pin{ctx_, ctx_->host_alloc(cand.size() * 8 + cand.size() * 256 * 4)};
      unsigned long long* h_counts = (unsigned long long*)pin.p;
      uint32_t* h_present = (uint32_t*)(h_counts + cand.size());  // (a mapping is only used when ≤ 256 ids are present)
      hip_check(hipMemcpyAsync(h_counts, pa.counts, cand.size() * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(present counts)");
      for (size_t k = 0; k < cand.size(); k++)
        hip_check(hipMemcpyAsync(h_present + k * 256, pa.present + pa.remap_off[k], std::min<size_t>(256, pa.dict_len[k]) * 4, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(present ids)");
      hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
      for (size_t k = 0; k < cand.size(); k++) {
        const size_t c = cand[k];
        const unsigned long long cnt = h_counts[k];
        const int w = cnt <= 4 ? -2 : cnt <= 16 ? -4 : cnt <= 256 ? 1 : width[c];
        if (w == width[c]) continue;  // nothing gained
        width[c] = w;
        remap_of[c] = pa.remap + pa.remap_off[k];
        present_of[c].assign(h_present + k * 256, h_present + k * 256 + std::max<unsigned long long>(cnt, 1));
        if (cnt == 0) present_of[c][0] = 0;  // (a column of NULLs only: every row carries rank 0, masked by its validity bit)
      }
      if (pt.on) pt.mark("finish: present ids");
    }
  }
  for (size_t c = 0; c < n_cols; c++) if (narrow[c]) { off_narrow[c] = slice_stride; slice_stride += col_bytes(c, kSliceRows); }
  narrow_bytes = slice_stride * std::max<size_t>(n_slices, 1);
  if (resident == nullptr && narrow_bytes > 0) d_narrow = (unsigned char*)alloc(narrow_bytes + 256);
  for (size_t c = 0; c < n_cols; c++) {
    if (narrow[c]) d_key[c] = d_narrow + off_narrow[c];
    std::memset(&cols[c], 0, sizeof(FdbHashCol));
    cols[c].kind = gcols_[c].kind; cols[c].word = gcols_[c].word; cols[c].gi = (int)c;
    cols[c].src_word = width[c]; cols[c].lut_len = narrow[c] ? 1u : 0u;
    cols[c].lut = remap_of[c];
    if (remap_of[c] != nullptr) a.any_lut = 1;
  }
  a.cols = (const FdbHashCol*)upload(cols.data(), cols.size() * sizeof(FdbHashCol));
  a.out_key = (void* const*)upload(d_key.data(), d_key.size() * sizeof(void*));
  a.slice_stride = slice_stride;
  if (resident != nullptr) {
    // ---- resident result: one column pass over all rows, the NULL counts to the host, the batch's column table ----------------
    struct Quiesce1 { hipStream_t a; ~Quiesce1() { (void)hipStreamSynchronize(a); } } q1{stream_};
    std::vector<unsigned long long> h_nulls(std::max<size_t>(n_cols, 1), 0);
    if (n > 0) {
      a.row_begin = 0; a.row_end = n;
      hip_check(fdb_launch_hash_rows_to_columns(a, device_, stream_), "hash rows to columns");
      for (size_t j = 0; j < aggs_.size(); j++)  // float64 MIN / MAX live as order-preserving integer keys: back to doubles, in place
        if (aggs_[j].type == FDB_T_F64 && (aggs_[j].func == FDB_AGG_MIN || aggs_[j].func == FDB_AGG_MAX))
          hip_check(fdb_launch_ordered_to_f64(d_vals[1 + j], (int64_t)n, stream_), "decode float keys");
      if (n_cols > 0) hip_check(hipMemcpyAsync(h_nulls.data(), a.out_nulls, n_cols * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(null counts)");
    }
    sync();
    resident->device = device_;
    resident->rows = (int64_t)n;
    for (size_t c = 0; c < n_cols; c++) {
      const GroupColState& g = gcols_[c];
      DevColumn d;
      d.name = g.name; d.length = (int64_t)n; d.null_count = (int64_t)h_nulls[c];
      if (g.kind == 0) {
        d.kind = ColKind::DICT; d.format = "I";
        std::vector<std::string> vals;
        for (const std::string_view& v : g.values) vals.emplace_back(v);
        d.dict = make_dictionary(std::move(vals), g.value_format);
        if (g.plain) {  // a plain string / binary key column: its distinct values, marked as such (not interned with real dictionaries)
          std::shared_ptr<HostDict> pd(new HostDict(*d.dict));
          pd->plain = true;
          d.dict = pd;
        }
        d.value_bytes = (int64_t)n * 4;
      } else {
        d.kind = g.is_bool ? ColKind::BOOL : g.is_u64 ? ColKind::U64 : ColKind::I64;
        d.format = g.is_bool ? "b" : g.is_u64 ? "L" : "l";
        d.value_bytes = g.is_bool ? (int64_t)(n + 7) / 8 : (int64_t)n * 8;
      }
      if (n > 0) {
        d.d_values = d_key[c];
        if (d.null_count > 0) { d.d_validity = d_bits[c]; d.validity_bytes = (int64_t)(n + 7) / 8; }
      }
      resident->payload_bytes += d.value_bytes + d.validity_bytes;
      resident->cols.push_back(std::move(d));
    }
    for (size_t j = 0; j < aggs_.size(); j++) {
      const AggState& A = aggs_[j];
      const bool count_from_cnt = A.func == FDB_AGG_COUNT && !final_stage_;
      DevColumn d;
      d.name = A.emit_name; d.length = (int64_t)n;
      const bool is_f64 = !count_from_cnt && A.type == FDB_T_F64;
      d.kind = is_f64 ? ColKind::F64 : ColKind::I64;
      d.format = is_f64 ? "g" : "l";
      if (n > 0) d.d_values = d_vals[count_from_cnt ? 0 : 1 + j];
      d.value_bytes = (int64_t)n * 8;
      resident->payload_bytes += d.value_bytes;
      resident->cols.push_back(std::move(d));
    }
    finished_ = true;
    return (int64_t)n;
  }
  if (n > 0) {
    h_block = (unsigned char*)pinned_pool_alloc(total);
    backing = std::shared_ptr<void>(h_block, [](void* p) { pinned_pool_free(p); });
    if (n_narrow > 0) h_narrow = (unsigned char*)pinned_pool_alloc(narrow_bytes);
  }
  struct FreeNarrow { unsigned char* p; ~FreeNarrow() { if (p) pinned_pool_free(p); } } free_narrow{h_narrow};
  // copies run on a second queue: slice k crosses PCIe while pass 2 produces slice k + 1. Whatever happens below, nothing is
  // still writing into the host blocks when they go back to their pools.
  hipStream_t copy_stream = n > 0 ? ctx_->aux_stream(0) : nullptr;
  struct Quiesce { hipStream_t a, b; ~Quiesce() { if (a) (void)hipStreamSynchronize(a); if (b) (void)hipStreamSynchronize(b); } } quiesce{copy_stream, stream_};
  pt.mark("finish: launch");
  // column descriptors (dictionaries are rebuilt on the host while the copy is in flight)
  out->clear();
  for (size_t c = 0; c < n_cols; c++) {
    const GroupColState& g = gcols_[c];
    OutColumn oc;
    oc.name = g.name;
    oc.length = (int64_t)n;
    oc.format = g.kind == 0 ? "I" : g.is_bool ? "b" : g.is_u64 ? "L" : "l";
    if (n > 0) { oc.backing = backing; oc.ext_values = g.is_bool ? nullptr : h_block + off_key[c]; oc.ext_validity = h_block + off_bits[c]; }  // (narrow columns: the widened array)
    if (g.kind == 0 && !g.plain) {
      // (the column's values ARE one interned dictionary — the usual case, parts of a table share theirs: its cached Arrow offsets and its
      // bytes are handed out as they are instead of copying every value into a new buffer per Finish)
      if (g.whole_dictionary() != nullptr && g.adopted->value_format == g.value_format && g.adopted->concat.size() <= 0x7FFFFFFFull) {
        oc.is_dict = true; oc.dict_format = g.value_format; oc.dict_ref = g.adopted;
      } else {
        set_dictionary(&oc, g.values, g.value_format);
      }
    }
    out->push_back(std::move(oc));
  }
  std::vector<size_t> out_of_agg(aggs_.size(), (size_t)-1);  // physical aggregate → its output column
  for (size_t j = 0; j < aggs_.size(); j++) {
    const AggState& A = aggs_[j];
    if (A.role == 2) continue;  // MAX half of UNIQUE
    OutColumn oc;
    oc.name = A.emit_name;
    oc.length = (int64_t)n;
    const bool count_from_cnt = A.func == FDB_AGG_COUNT && !final_stage_;
    const bool is_f64 = !count_from_cnt && A.type == FDB_T_F64;
    oc.format = A.role == 3 ? "b" : is_f64 ? "g" : "l";
    if (n > 0 && A.role != 3) { oc.backing = backing; oc.ext_values = h_block + off_val[count_from_cnt ? 0 : 1 + j]; }
    out_of_agg[j] = out->size();
    out->push_back(std::move(oc));
  }
  if (n > 0) {
    // per slice of 2^20 rows: pass 2 on the plan's stream, then — on the copy queue — the slice's narrow columns; the direct
    // part (wide keys, bitmaps, value columns, NULL counts) follows the last slice
    std::vector<hipEvent_t> landed(n_narrow > 0 ? n_slices : 0), produced(n_slices);
    struct PutEvents { Context* c; std::vector<hipEvent_t>* v; ~PutEvents() { for (hipEvent_t e : *v) if (e) c->put_event(e); } } put_landed{ctx_, &landed}, put_produced{ctx_, &produced};
    for (size_t sl = 0; sl < n_slices; sl++) {
      const size_t rows = std::min<size_t>(kSliceRows, (size_t)n - sl * kSliceRows);
      a.row_begin = sl * kSliceRows; a.row_end = a.row_begin + rows;
      hip_check(fdb_launch_hash_rows_to_columns(a, device_, stream_), "hash rows to columns");
      produced[sl] = ctx_->get_event();
      hip_check(hipEventRecord(produced[sl], stream_), "hipEventRecord");
      hip_check(hipStreamWaitEvent(copy_stream, produced[sl], 0), "hipStreamWaitEvent");
      if (n_narrow == 0) continue;
      const size_t bytes = rows == kSliceRows ? slice_stride : off_narrow[last_narrow] + col_bytes(last_narrow, rows);  // (short last slice)
      hip_check(hipMemcpyAsync(h_narrow + sl * slice_stride, d_narrow + sl * slice_stride, bytes, hipMemcpyDeviceToHost, copy_stream), "hipMemcpyAsync(narrow slice)");
      landed[sl] = ctx_->get_event();
      hip_check(hipEventRecord(landed[sl], copy_stream), "hipEventRecord");
    }
    // the NULL counts first (a few bytes, behind the last slice): the host learns from them which bitmaps have to cross at all
    hipEvent_t nulls_landed = ctx_->get_event();
    struct PutEvent { Context* c; hipEvent_t e; ~PutEvent() { c->put_event(e); } } put_nulls{ctx_, nulls_landed};
    hip_check(hipMemcpyAsync(h_block + off_nulls, d_block + (off_nulls - direct_begin), std::max<size_t>(n_cols, 1) * 8, hipMemcpyDeviceToHost, copy_stream), "hipMemcpyAsync(null counts)");
    hip_check(hipEventRecord(nulls_landed, copy_stream), "hipEventRecord");
    hip_check(hipMemcpyAsync(h_block + direct_begin, d_block, direct_copy_bytes, hipMemcpyDeviceToHost, copy_stream), "hipMemcpyAsync(result)");
    auto copy_bitmaps = [&] {  // (queued behind the direct part, which is still crossing when the counts are known)
      hip_check(hipEventSynchronize(nulls_landed), "hipEventSynchronize(null counts)");
      const unsigned long long* nulls = (const unsigned long long*)(h_block + off_nulls);
      for (size_t c = 0; c < n_cols; c++)
        if (nulls[c] != 0)
          hip_check(hipMemcpyAsync(h_block + off_bits[c], d_block + (off_bits[c] - direct_begin), (size_t)(n + 7) / 8, hipMemcpyDeviceToHost, copy_stream), "hipMemcpyAsync(validity bitmap)");
    };
    if (n_narrow == 0) copy_bitmaps();
    if (n_narrow > 0) {
      // one task = one narrow column of one slice
      std::vector<size_t> narrow_cols;
      for (size_t c = 0; c < n_cols; c++) if (narrow[c]) narrow_cols.push_back(c);
      const size_t n_tasks = n_slices * narrow_cols.size();
      auto run_task = [&](size_t t) {
        const size_t sl = t / narrow_cols.size(), c = narrow_cols[t % narrow_cols.size()];
        const size_t rows = std::min<size_t>(kSliceRows, (size_t)n - sl * kSliceRows);
        uint32_t* dst = (uint32_t*)(h_block + off_key[c]) + sl * kSliceRows;
        if (present_of[c].empty()) widen_indices(h_narrow + sl * slice_stride + off_narrow[c], width[c], dst, rows);
        else widen_indices_mapped(h_narrow + sl * slice_stride + off_narrow[c], width[c], present_of[c].data(), present_of[c].size(), dst, rows);
      };
      const int n_threads = host_threads_for((size_t)n * n_narrow);
      if (n_threads <= 0) {
        copy_bitmaps();
        hip_check(hipStreamSynchronize(copy_stream), "hipStreamSynchronize(copy queue)");
        for (size_t t = 0; t < n_tasks; t++) run_task(t);
      } else {
        std::atomic<size_t> next{0}, ready_slices{0};
        std::atomic<bool> failed{false};
        std::vector<std::thread> workers;
        auto work = [&] {
          for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= n_tasks) return;
            const size_t sl = t / narrow_cols.size();
            while (ready_slices.load(std::memory_order_acquire) <= sl) {
              if (failed.load()) return;
              std::this_thread::yield();
            }
            run_task(t);
          }
        };
        hipError_t err = hipSuccess;
        try {
          for (int i = 0; i < n_threads; i++) workers.emplace_back(work);
        } catch (...) { /* fewer threads than asked for: the ones that started (or this thread, below) do the work */ }
        for (size_t sl = 0; sl < n_slices && err == hipSuccess; sl++) {
          err = hipEventSynchronize(landed[sl]);
          if (err == hipSuccess) ready_slices.store(sl + 1, std::memory_order_release);
        }
        if (pt.on) pt.mark("finish: slices landed");
        if (err != hipSuccess) failed.store(true);
        else {
          try { copy_bitmaps(); } catch (...) { failed.store(true); for (std::thread& w : workers) w.join(); throw; }
          work();  // this thread helps with what is left
        }
        for (std::thread& w : workers) w.join();
        hip_check(err, "hipEventSynchronize(narrow slice)");
      }
    }
  }
  if (pt.on) pt.mark("finish: widened");
  if (copy_stream) hip_check(hipStreamSynchronize(copy_stream), "hipStreamSynchronize(copy queue)");
  sync();
  pt.mark("finish: copy");
  // post-processing that needs the data on the host: NULL counts, float64 MIN/MAX key decoding
  for (size_t c = 0; c < n_cols && n > 0; c++) {
    OutColumn& oc = (*out)[c];
    if (gcols_[c].is_bool) {  // bool key: 8-byte 1 (false) / 2 (true) on the device → Arrow's bit-packed bool
      const unsigned long long* v = (const unsigned long long*)(h_block + off_key[c]);
      oc.values.assign((size_t)(n + 7) / 8 + 8, 0);
      for (uint64_t i = 0; i < n; i++) if (v[i] >= 2ull) oc.values[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    if (gcols_[c].kind == 0 && gcols_[c].plain)  // plain string / binary key column: entry indices → offsets + bytes
      set_plain_strings(&oc, (const uint32_t*)(h_block + off_key[c]), ((const unsigned long long*)(h_block + off_nulls))[c] != 0 ? h_block + off_bits[c] : nullptr, (int64_t)n, gcols_[c].values,
                        gcols_[c].value_format);
    oc.null_count = (int64_t)((const unsigned long long*)(h_block + off_nulls))[c];
  }
  for (size_t j = 0; j < aggs_.size() && n > 0; j++) {  // composite reducers: UNIQUE's validity, AND's bits
    const AggState& A = aggs_[j];
    if (A.role == 1) {
      OutColumn& oc = (*out)[out_of_agg[j]];
      const unsigned long long* lo = (const unsigned long long*)(h_block + off_val[1 + j]);
      const unsigned long long* hi = (const unsigned long long*)(h_block + off_val[2 + j]);
      oc.validity.assign((size_t)(n + 7) / 8, 0);
      for (uint64_t i = 0; i < n; i++) {
        if (lo[i] == hi[i]) oc.validity[i >> 3] |= (uint8_t)(1u << (i & 7));
        else { oc.null_count++; std::memset(h_block + off_val[1 + j] + i * 8, 0, 8); }
      }
    } else if (A.role == 3) {
      OutColumn& oc = (*out)[out_of_agg[j]];
      const unsigned long long* v = (const unsigned long long*)(h_block + off_val[1 + j]);
      oc.values.assign((size_t)(n + 7) / 8 + 8, 0);
      for (uint64_t i = 0; i < n; i++) if (v[i] >= 2ull) oc.values[i >> 3] |= (uint8_t)(1u << (i & 7));  // MIN over 1 (false) / 2 (true)
    }
  }
  for (size_t j = 0; j < aggs_.size() && n > 0; j++) {
    const AggState& A = aggs_[j];
    if (A.type == FDB_T_F64 && (A.func == FDB_AGG_MIN || A.func == FDB_AGG_MAX)) {
      unsigned char* v = h_block + off_val[1 + j];
      for (uint64_t i = 0; i < n; i++) {
        int64_t k; std::memcpy(&k, v + i * 8, 8);
        const double d = fdb_ordered_to_f64_host(k);
        std::memcpy(v + i * 8, &d, 8);
      }
    }
  }
  pt.mark("finish: host post");
  return (int64_t)n;
}

// ---- exchange of hash tables ----------------------------------------------------------------------------------------------
void Plan::group_schema(ArrowArray* out, ArrowSchema* out_schema) {
  std::vector<OutColumn> cols;
  for (const GroupColState& g : gcols_) {
    OutColumn oc;
    oc.name = g.name;
    oc.length = 0;
    if (g.kind == 0) {
      // a plain string / binary key column also travels as its value set; the SIGNED index type marks it (columns of real
      // dictionaries are always described with uint32 indices)
      oc.format = g.plain ? "i" : "I";
      set_dictionary(&oc, g.values, g.value_format);
    } else {
      oc.format = g.is_bool ? "b" : g.is_u64 ? "L" : "l";
    }
    cols.push_back(std::move(oc));
  }
  // the value types of the aggregated columns travel with the schema (a rank that saw no record does not know them)
  for (const AggState& A : aggs_) {
    if (A.func == FDB_AGG_COUNT || A.type == FDB_T_NONE || A.role == 2) continue;
    OutColumn oc;
    oc.name = A.result_name;
    oc.length = 0;
    oc.format = A.type == FDB_T_F64 ? "g" : "l";
    cols.push_back(std::move(oc));
  }
  export_record(std::move(cols), 0, out, out_schema);
}

void Plan::seed_groups(const ArrowArray* array, const ArrowSchema* schema) {
  runs_to_table();
  HostRecordView view;
  view_record(array, schema, &view);
  if (mode_ == TableMode::DENSE) switch_to_hash();  // (before the column set changes)
  for (const HostColView& c : view.cols) {
    bool is_agg = false;
    for (AggState& A : aggs_) {
      if (A.func == FDB_AGG_COUNT || c.name != A.result_name) continue;
      const int32_t t = c.kind == ColKind::F64 ? FDB_T_F64 : c.kind == ColKind::I64 ? FDB_T_I64 : FDB_T_NONE;
      if (t == FDB_T_NONE) throw Error(FDB_ERR_INVALID, "aggregate column " + c.name + " has an unsupported type");
      if (A.type != FDB_T_NONE && A.type != t) throw Error(FDB_ERR_INVALID, "aggregation types differ between plans");
      A.type = t;
      is_agg = true;
    }
    if (is_agg) continue;
    if (c.kind != ColKind::DICT && c.kind != ColKind::I64 && c.kind != ColKind::U64 && c.kind != ColKind::BOOL)
      throw Error(FDB_ERR_UNSUPPORTED, "group column " + c.name + ": only dictionary, string, int64, uint64 and bool columns can be group keys");
    const int kind = c.kind == ColKind::DICT ? 0 : 1;
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == c.name) break;
    if (gi == gcols_.size()) {
      GroupColState g;
      g.name = c.name; g.kind = kind; g.is_bool = c.kind == ColKind::BOOL; g.is_u64 = c.kind == ColKind::U64; g.plain = kind == 0 && c.format == "i"; g.cap = 1; g.stride = 0;
      gcols_.push_back(std::move(g));
    }
    GroupColState& g = gcols_[gi];
    if (g.kind != kind || g.is_bool != (c.kind == ColKind::BOOL) || g.is_u64 != (c.kind == ColKind::U64) || g.plain != (kind == 0 && c.format == "i")) throw Error(FDB_ERR_INVALID, "group column " + c.name + " has a different type in this plan");
    if (kind == 0) {
      std::shared_ptr<HostDict> d = read_dictionary(c);
      g.value_format = g.plain ? std::string(c.schema->dictionary->format) : d->value_format;  // (a plain column keeps a large type)
      g.owners.push_back(d);
      for (const std::string& v : d->values) g.intern(std::string_view(v));
    }
  }
  if (gcols_.size() > FDB_MAX_HASH_GCOLS) throw Error(FDB_ERR_UNSUPPORTED, "too many group columns");
  hash_layout();
  hash_reserve(0);
}

void Plan::hash_export(Plan& layout, int n_parts, void** dev_rows, int64_t* counts, int32_t* row_words32) {
  runs_to_table();
  layout.runs_to_table();
  if (n_parts < 1 || n_parts > FDB_MAX_PARTS) throw Error(FDB_ERR_INVALID, "partition count out of range");
  if (layout.device_ != device_) throw Error(FDB_ERR_INVALID, "layout plan lives on another device");
  if (layout.aggs_.size() != aggs_.size()) throw Error(FDB_ERR_INVALID, "plans have different aggregations");
  hip_check(hipSetDevice(device_), "hipSetDevice");
  if (mode_ == TableMode::DENSE) switch_to_hash();
  hash_layout();
  hash_reserve(0);
  if (&layout != this && layout.mode_ == TableMode::DENSE) layout.switch_to_hash();  // (before its column set changes: the dense slot decoding needs the old one)
  // the layout plan adopts our columns / dictionary values (a no-op when it was seeded with the global schema)
  std::vector<FdbHashCol> cols(std::max<size_t>(gcols_.size(), 1));
  std::vector<std::vector<uint32_t>> id_map(gcols_.size());
  for (size_t sc = 0; sc < gcols_.size(); sc++) {
    const GroupColState& sg = gcols_[sc];
    size_t gi = 0;
    for (; gi < layout.gcols_.size(); gi++) if (layout.gcols_[gi].name == sg.name) break;
    if (gi == layout.gcols_.size()) {
      GroupColState g;
      g.name = sg.name; g.kind = sg.kind; g.is_bool = sg.is_bool; g.is_u64 = sg.is_u64; g.plain = sg.plain; g.value_format = sg.value_format; g.cap = 1; g.stride = 0;
      layout.gcols_.push_back(std::move(g));
    }
    GroupColState& g = layout.gcols_[gi];
    if (g.kind != sg.kind || g.plain != sg.plain || g.is_bool != sg.is_bool || g.is_u64 != sg.is_u64) throw Error(FDB_ERR_INVALID, "group column " + sg.name + " has different types in the two plans");
    if (sg.kind == 0) {
      g.owners.insert(g.owners.end(), sg.owners.begin(), sg.owners.end());
      id_map[sc].assign(sg.values.size() + 1, 0);
      for (size_t v = 0; v < sg.values.size(); v++) id_map[sc][v + 1] = g.intern(sg.values[v]);
    }
  }
  if (layout.gcols_.size() > FDB_MAX_HASH_GCOLS) throw Error(FDB_ERR_UNSUPPORTED, "too many group columns");
  if (&layout != this) layout.hash_layout();
  for (size_t sc = 0; sc < gcols_.size(); sc++) {
    size_t gi = 0;
    for (; gi < layout.gcols_.size(); gi++) if (layout.gcols_[gi].name == gcols_[sc].name) break;
    FdbHashCol& C = cols[sc];
    std::memset(&C, 0, sizeof(C));
    C.kind = gcols_[sc].kind; C.word = gcols_[sc].word; C.src_word = layout.gcols_[gi].word; C.gi = (int)gi; C.lut_len = (uint32_t)sc;
    C.lut_lds = FDB_NO_LDS; C.k1 = fdb_fp_k1((int)gi); C.k2 = fdb_fp_k2((int)gi);
    if (C.kind == 0) C.lut = (const uint32_t*)upload(id_map[sc].data(), id_map[sc].size() * 4);
  }
  const int dkw = layout.h_key_words_;
  const int n_vals = (int)(1 + aggs_.size());
  const int rw = fdb_packed_row_words(dkw, n_vals);
  *row_words32 = rw;
  const uint64_t n = hash_groups();
  for (int p = 0; p < n_parts; p++) counts[p] = 0;
  *dev_rows = nullptr;
  if (n == 0) return;
  unsigned long long* d_counts = (unsigned long long*)ctx_->dev_alloc(FDB_MAX_PARTS * 8);
  uint32_t* d_rows = (uint32_t*)ctx_->dev_alloc((size_t)n * rw * 4);
  scratch_.push_back(d_counts); scratch_.push_back(d_rows);
  FdbHashPartArgs a;
  std::memset(&a, 0, sizeof(a));
  a.table = h_table_; a.keys = h_keys_; a.capacity = h_capacity_;
  a.cols = (const FdbHashCol*)upload(cols.data(), cols.size() * sizeof(FdbHashCol));
  a.out = d_rows; a.counts = d_counts;
  a.n_cols = (int)gcols_.size(); a.entry_words = h_entry_words_; a.key_words = h_key_words_; a.dst_key_words = dkw; a.row_words32 = rw;
  a.n_vals = n_vals; a.n_parts = n_parts;
  a.in_words = std::min((h_key_used_ + 3) & ~3, h_key_words_);
  bool same = dkw >= a.in_words && layout.gcols_.size() == gcols_.size();  // (a layout column we do not have would keep what our tuple holds at its word)
  for (size_t sc = 0; sc < gcols_.size(); sc++) if (cols[sc].src_word != cols[sc].word) same = false;
  a.same_layout = same;
  bool same_ids = same;
  for (size_t sc = 0; sc < gcols_.size() && same_ids; sc++) {
    if (cols[sc].gi != (int)sc) same_ids = false;
    for (size_t v = 1; v < id_map[sc].size() && same_ids; v++) if (id_map[sc][v] != (uint32_t)v) same_ids = false;
  }
  a.same_ids = same_ids;  // (the usual case for the parts of one table: every rank interned the same dictionaries in the same order)
  // counts per (wave, partition) → region bases → scatter: three launches back to back, one wait
  void* d_part_scratch = ctx_->dev_alloc(fdb_hash_partition_scratch_bytes(device_, a));
  scratch_.push_back(d_part_scratch);
  hip_check(fdb_launch_hash_partition(a, device_, d_part_scratch, stream_), "hash partition");
  unsigned long long h_counts[FDB_MAX_PARTS];
  hip_check(hipMemcpyAsync(h_counts, d_counts, (size_t)n_parts * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(partition counts)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");  // the caller reads the rows next
  unsigned long long run = 0;
  for (int p = 0; p < n_parts; p++) { run += h_counts[p]; counts[p] = (int64_t)h_counts[p]; }
  if (run != n) throw Error(FDB_ERR_DEVICE, "internal: partition counts do not add up to the group count");
  *dev_rows = d_rows;
}

void Plan::hash_import(const void* dev_rows, int64_t n_rows, bool unique_rows) {
  runs_to_table();
  if (n_rows <= 0) return;
  hip_check(hipSetDevice(device_), "hipSetDevice");
  PhaseTimer pt;
  if (mode_ == TableMode::DENSE) switch_to_hash();
  hash_layout();
  if (h_count_dev_ != nullptr) hash_groups();
  hash_reserve((uint64_t)n_rows);
  if (pt.on) { hip_check(hipStreamSynchronize(stream_), "sync"); pt.mark("import: reserve"); }
  const int kw = h_key_words_, n_vals = (int)(1 + aggs_.size());
  const int rw = fdb_packed_row_words(kw, n_vals);
  std::vector<FdbHashCol> cols(std::max<size_t>(gcols_.size(), 1));
  for (size_t c = 0; c < gcols_.size(); c++) {
    FdbHashCol& C = cols[c];
    std::memset(&C, 0, sizeof(C));
    C.kind = gcols_[c].kind; C.word = gcols_[c].word; C.src_word = gcols_[c].word; C.gi = (int)c; C.lut_len = (uint32_t)c; C.lut_lds = FDB_NO_LDS;
    C.k1 = fdb_fp_k1((int)c); C.k2 = fdb_fp_k2((int)c);
  }
  FdbHashMergeArgs m;
  hash_merge_args(&m, cols, rw);
  m.in_keys = (const uint32_t*)dev_rows;
  m.entries = (const unsigned long long*)((const uint32_t*)dev_rows + kw);
  m.n = n_rows;
  m.in_key_words = rw; m.in_entry_words = rw / 2;
  m.unique_source = unique_rows ? 1 : 0;  // (rows of several ranks in one call: a group comes once per rank that saw it)
  hip_check(fdb_launch_hash_merge(m, device_, stream_), "hash merge");
  hip_check(hipStreamSynchronize(stream_), "sync(hash import)");  // the caller may free `dev_rows` when this returns
  pt.mark("import: merge kernel");
  state_dirty_ = true;
}

// Table into table, on the device, in ONE pass over the source's slots: the merge kernel reads the occupied tuples where they lie,
// re-keys them into this plan's key ids (per-column LUTs; none for a column whose dictionaries agree value for value) and inserts /
// folds them here. No packed intermediate (rounds 2-5 exported rows and imported them: two more passes over every tuple).
void Plan::merge_hash_tables(Plan& src) {
  if (src.device_ != device_) throw Error(FDB_ERR_INVALID, "source plan lives on another device");
  if (src.aggs_.size() != aggs_.size()) throw Error(FDB_ERR_INVALID, "plans have different aggregations");
  hip_check(hipSetDevice(device_), "hipSetDevice");
  src.hash_layout();
  if (mode_ == TableMode::DENSE) switch_to_hash();  // (before the column set changes: the dense slot decoding needs the old one)
  const uint64_t n_src = src.h_table_ != nullptr ? src.hash_groups() : 0;
  // adopt the source's columns and dictionary values
  std::vector<int> dst_of(src.gcols_.size());
  std::vector<std::vector<uint32_t>> id_map(src.gcols_.size());
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
    const GroupColState& sg = src.gcols_[sc];
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == sg.name) break;
    if (gi == gcols_.size()) {
      GroupColState g;
      g.name = sg.name; g.kind = sg.kind; g.is_bool = sg.is_bool; g.is_u64 = sg.is_u64; g.plain = sg.plain; g.value_format = sg.value_format; g.cap = 1; g.stride = 0;
      gcols_.push_back(std::move(g));
    }
    GroupColState& g = gcols_[gi];
    if (g.kind != sg.kind || g.plain != sg.plain || g.is_bool != sg.is_bool || g.is_u64 != sg.is_u64) throw Error(FDB_ERR_INVALID, "group column " + sg.name + " has different types in the two plans");
    dst_of[sc] = (int)gi;
    if (sg.kind == 0) {
      g.owners.insert(g.owners.end(), sg.owners.begin(), sg.owners.end());
      bool identity = true;
      id_map[sc].assign(sg.values.size() + 1, 0);
      for (size_t v = 0; v < sg.values.size(); v++) { id_map[sc][v + 1] = g.intern(sg.values[v]); identity = identity && id_map[sc][v + 1] == (uint32_t)(v + 1); }
      if (identity) id_map[sc].clear();
    }
  }
  if (gcols_.size() > FDB_MAX_HASH_GCOLS) throw Error(FDB_ERR_UNSUPPORTED, "too many group columns");
  hash_layout();
  if (n_src == 0) { hash_reserve(0); return; }
  if (h_count_dev_ != nullptr) hash_groups();
  hash_reserve(n_src);
  std::vector<FdbHashCol> cols(std::max<size_t>(gcols_.size(), 1));
  for (size_t c = 0; c < gcols_.size(); c++) {
    std::memset(&cols[c], 0, sizeof(FdbHashCol));
    cols[c].kind = gcols_[c].kind; cols[c].word = gcols_[c].word; cols[c].gi = (int)c; cols[c].src_word = -1; cols[c].lut_lds = FDB_NO_LDS;
    cols[c].k1 = fdb_fp_k1((int)c); cols[c].k2 = fdb_fp_k2((int)c);
  }
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
    FdbHashCol& C = cols[(size_t)dst_of[sc]];
    C.src_word = src.gcols_[sc].word;
    if (src.gcols_[sc].kind == 0) { if (!id_map[sc].empty()) C.lut = (const uint32_t*)upload(id_map[sc].data(), id_map[sc].size() * 4); }
    else C.lut_len = (uint32_t)sc;  // the source plan's column index: the bit of the incoming valid mask
  }
  FdbHashMergeArgs m;
  hash_merge_args(&m, cols, src.h_key_words_);
  m.src_table = src.h_table_; m.src_keys = src.h_keys_; m.src_capacity = src.h_capacity_;
  m.src_key_words = src.h_key_words_; m.src_entry_words = src.h_entry_words_;
  m.unique_source = 1;
  // the source's stream may still be writing its table
  src.sync();
  hip_check(fdb_launch_hash_merge(m, device_, stream_), "hash merge (table)");
  state_dirty_ = true;
  h_groups_bound_ += n_src;
  h_bound_stale_ = true;
  sync();  // the caller may close `src` when this returns
}

// ≙ Synchronizer + final stage when either side holds a hash table: the source's occupied groups are re-keyed into
// this plan's key ids on the device (per-column translation LUTs) and merged with atomics.
void Plan::merge_hash(Plan& src) {
  runs_to_table();
  if (src.mode_ == TableMode::HASH) { merge_hash_tables(src); return; }
  CompactState cs;
  src.fetch_compact(&cs);
  if (cs.n == 0) return;
  // adopt the source's columns and dictionary values
  std::vector<int> dst_of(src.gcols_.size());
  std::vector<std::vector<uint32_t>> id_map(src.gcols_.size());
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
    const GroupColState& sg = src.gcols_[sc];
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == sg.name) break;
    if (gi == gcols_.size()) {
      GroupColState g;
      g.name = sg.name; g.kind = sg.kind; g.is_bool = sg.is_bool; g.is_u64 = sg.is_u64; g.plain = sg.plain; g.value_format = sg.value_format; g.cap = 1; g.stride = 0;
      gcols_.push_back(std::move(g));
    }
    GroupColState& g = gcols_[gi];
    if (g.kind != sg.kind || g.plain != sg.plain || g.is_bool != sg.is_bool || g.is_u64 != sg.is_u64) throw Error(FDB_ERR_INVALID, "group column " + sg.name + " has different types in the two plans");
    dst_of[sc] = (int)gi;
    if (sg.kind == 0) {
      g.owners.insert(g.owners.end(), sg.owners.begin(), sg.owners.end());
      id_map[sc].assign(sg.values.size() + 1, 0);
      for (size_t v = 0; v < sg.values.size(); v++) id_map[sc][v + 1] = g.intern(sg.values[v]);
    }
  }
  if (mode_ == TableMode::DENSE) switch_to_hash();
  hash_layout();
  // incoming tuples: [mask lo, mask hi, per source column: 1 word (dictionary id) or 2 words (int64)]
  std::vector<int> src_word(src.gcols_.size());
  int in_kw = 2;
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) { src_word[sc] = in_kw; in_kw += src.gcols_[sc].kind == 0 ? 1 : 2; }
  std::vector<uint32_t> keys((size_t)cs.n * in_kw, 0);
  const size_t in_ew = 1 + aggs_.size();
  std::vector<unsigned long long> entries((size_t)cs.n * in_ew);
  for (int64_t i = 0; i < cs.n; i++) {
    uint64_t vm = 0;
    for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
      if (src.gcols_[sc].kind == 0) {
        const uint32_t id = cs.ids[sc].empty() ? 0u : cs.ids[sc][(size_t)i];
        keys[(size_t)i * in_kw + src_word[sc]] = id;
        if (id) vm |= 1ull << sc;
      } else if (!cs.ivalid[sc].empty() && cs.ivalid[sc][(size_t)i]) {
        const uint64_t v = (uint64_t)cs.ivals[sc][(size_t)i];
        keys[(size_t)i * in_kw + src_word[sc]] = (uint32_t)v;
        keys[(size_t)i * in_kw + src_word[sc] + 1] = (uint32_t)(v >> 32);
        vm |= 1ull << sc;
      }
    }
    keys[(size_t)i * in_kw] = (uint32_t)vm; keys[(size_t)i * in_kw + 1] = (uint32_t)(vm >> 32);
    entries[(size_t)i * in_ew] = cs.cnt[(size_t)i];
    for (size_t j = 0; j < aggs_.size(); j++) entries[(size_t)i * in_ew + 1 + j] = cs.acc[j][(size_t)i];
  }
  std::vector<FdbHashCol> cols(gcols_.size());
  for (size_t c = 0; c < gcols_.size(); c++) {
    std::memset(&cols[c], 0, sizeof(FdbHashCol));
    cols[c].kind = gcols_[c].kind; cols[c].word = gcols_[c].word; cols[c].gi = (int)c; cols[c].src_word = -1; cols[c].lut_lds = FDB_NO_LDS;
    cols[c].k1 = fdb_fp_k1((int)c); cols[c].k2 = fdb_fp_k2((int)c);
  }
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
    FdbHashCol& C = cols[(size_t)dst_of[sc]];
    C.src_word = src_word[sc];
    if (src.gcols_[sc].kind == 0) C.lut = (const uint32_t*)upload(id_map[sc].data(), id_map[sc].size() * 4);
    else C.lut_len = (uint32_t)sc;  // the source plan's column index: selects the bit of the incoming valid mask
  }
  hash_insert_entries(entries, keys, cs.n, in_kw, cols);
  sync();
}

// ---- table-free OrderedAggregate: the run store (fdb_plan.h: RunSegment; fdb_kernels.h "run store") ---------------------------------

// May the records of this push go through the run kernel? One aggregation that is not a composite, the specialised kernels
// available, room for the segments — and a run record that the waves' LDS stages can hold a useful number of. Which RECORD a launch
// writes is decided per pushed record (runs_format): the narrow one — a byte per key id — while every group column is a dictionary
// column of ≤ 254 distinct values, there are at most FDB_RUN_TUPLE_BYTES of them and the record carries all of them in the plan's
// order; otherwise the wide one (the table's own key tuple: any cardinality, int64 and computed keys, absent columns).
bool Plan::runs_wanted(const DeviceBatch* const* bs, const std::vector<Resolved>& Rs, const std::vector<int>& live) const {
  static const bool off = std::getenv("FDB_NO_RUNS") != nullptr;  // (A/B and test aid: ordered plans take the hash table + sort)
  if (off || !ordered_ || !jit_possible() || aggs_.size() != 1 || aggs_[0].role != 0) return false;
  if (gcols_.empty() || gcols_.size() > FDB_MAX_HASH_GCOLS) return false;
  if (runs_.size() + live.size() > FDB_MAX_RUN_SEGMENTS) return false;
  // Small key spaces stay with the dense table: a table that fits LDS is the fastest scan there is (one launch for every record, no run
  // store to size and merge — cfg 2's query over a table sorted by labels.path: 0.3 ms per 100 M rows against 0.9 ms of run kernels and a
  // Finish over 390 k runs) and its few thousand groups are sorted on the host in microseconds. The run store is for key spaces that would
  // need the global table. ($FDB_RUNS_ALWAYS: test aid — the run machinery on small shapes)
  if (runs_.empty() && !knobs_.runs_always && gcols_.size() <= FDB_MAX_DENSE_GCOLS) {
    bool dense = true;
    uint64_t space = 1;
    for (const GroupColState& g : gcols_) {
      if (g.kind != 0) { dense = false; break; }
      space *= (uint64_t)g.values.size() + 1;
      if (space > 8192) { dense = false; break; }
    }
    if (dense) return false;
  }
  size_t kw = 4;  // (hash_layout() has not seen the columns this push added yet)
  for (const GroupColState& g : gcols_) kw += g.kind == 0 ? 1 : 2;
  kw = (kw + 3) & ~(size_t)3;
  const uint64_t wide_bytes = (uint64_t)(kw + 4) * 4;
  if (wide_bytes * 16 > FDB_RUN_WAVE_LDS) return false;  // (a wave's stage should hold a tile's worth of runs of ordered input)
  for (int i : live) {
    if ((uint64_t)bs[i]->rows >= (1ull << 31)) return false;
    const int fmt = runs_format(Rs[(size_t)i]);
    const uint64_t rec = fmt == 0 ? (uint64_t)FDB_RUN_BYTES : fmt == 1 ? (uint64_t)FDB_RUN_MEDIUM_BYTES : wide_bytes;
    if ((uint64_t)bs[i]->rows * (rec + 4) > ((uint64_t)48 << 30)) return false;  // (the run store is sized for the worst case — every row a run)
  }
  return true;
}

// Which run record does this record's launch write: 0 narrow (a byte per key id), 1 medium (two bytes), 2 wide.
int Plan::runs_format(const Resolved& R) const {
  const char force = knobs_.runs_wide;  // (A/B and test aid: '1' every launch writes wide records, 'm' medium ones where narrow would do)
  if (force == '1' || gcols_.size() > FDB_RUN_TUPLE_BYTES || R.groups.size() != gcols_.size()) return 2;
  size_t most = 0;
  for (const GroupColState& g : gcols_) { if (g.kind != 0) return 2; most = std::max(most, g.values.size()); }
  for (size_t g = 0; g < R.groups.size(); g++) if (R.groups[g].kind != 0 || R.groups[g].gi != (int)g) return 2;
  if (most > 65534) return 2;
  return most > 254 || force != 0 ? 1 : 0;
}

void Plan::runs_free() {
  for (RunSegment& r : runs_) ctx_->dev_free(r.block);
  runs_.clear();
}

int32_t Plan::runs_func() const {
  const AggState& A = aggs_[0];
  return A.func == FDB_AGG_COUNT ? (final_stage_ ? 1 : 0) : A.func == FDB_AGG_SUM ? (A.type == FDB_T_F64 ? 2 : 1) : A.func == FDB_AGG_MIN ? 3 : 4;
}

// The runs of every segment in row order: concatenated directory → prefix sums → (segment, index) of every logical run →
// "starts a new key" flags (+ the order check) → prefix sums of those = the group of every run.
bool Plan::runs_prepare(RunsView* v, bool check_order, std::vector<void*>* owned) {
  hip_check(hipSetDevice(device_), "hipSetDevice");
  auto alloc = [&](size_t bytes) { void* p = ctx_->dev_alloc(std::max<size_t>(bytes, 256)); owned->push_back(p); return p; };
  std::memset(&v->segs, 0, sizeof(v->segs));
  int64_t n_entries = 0;
  v->segs.n_segs = (int32_t)runs_.size();
  bool any_wide = false;
  for (size_t k = 0; k < runs_.size(); k++) {
    v->segs.tuples[k] = runs_[k].tuples;
    v->segs.run_words[k] = runs_[k].run_words;
    any_wide = any_wide || runs_[k].run_words != 0;
    v->segs.first_entry[k] = (uint32_t)n_entries;
    n_entries += runs_[k].n_entries;
  }
  v->segs.first_entry[runs_.size()] = (uint32_t)n_entries;
  if (n_entries >= (int64_t)1 << 31) throw Error(FDB_ERR_UNSUPPORTED, "ordered aggregate: too many rows for one run store");
  uint32_t* d_dir = (uint32_t*)alloc((size_t)n_entries * 8);
  for (size_t k = 0; k < runs_.size(); k++)
    hip_check(hipMemcpyAsync(d_dir + (size_t)v->segs.first_entry[k] * 2, runs_[k].dir, (size_t)runs_[k].n_entries * 8, hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync(run directory)");
  uint32_t* d_starts = (uint32_t*)alloc((size_t)n_entries * 4);
  unsigned long long* d_scratch = (unsigned long long*)alloc(((size_t)std::max<int64_t>(n_entries, 1) / 1024 + 4) * 8 + 256);
  unsigned long long* d_totals = (unsigned long long*)alloc(256);  // [0] runs, [1] groups, [2] order violation (low word)
  hip_check(hipMemsetAsync(d_totals, 0, 256, stream_), "hipMemsetAsync(totals)");
  hip_check(fdb_launch_scan_u32(d_dir + 1, 2, d_starts, n_entries, d_scratch, d_totals, stream_), "scan run counts");
  unsigned long long h_tot[3] = {0, 0, 0};
  hip_check(hipMemcpyAsync(h_tot, d_totals, 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(run count)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  v->n_runs = (int64_t)h_tot[0];
  v->n_groups = 0;
  if (v->n_runs == 0) return true;
  if (v->n_runs >= (int64_t)1 << 32) throw Error(FDB_ERR_UNSUPPORTED, "ordered aggregate: too many runs");
  v->phys = (unsigned long long*)alloc((size_t)v->n_runs * 8);
  hip_check(fdb_launch_runs_map(d_dir, d_starts, n_entries, v->segs, v->phys, stream_), "runs map");
  if (!check_order) { v->flags = nullptr; v->out_idx = nullptr; v->n_groups = v->n_runs; return true; }
  v->flags = (uint32_t*)alloc((size_t)v->n_runs * 4);
  v->out_idx = (uint32_t*)alloc((size_t)v->n_runs * 4);
  unsigned long long* d_scratch2 = (unsigned long long*)alloc(((size_t)v->n_runs / 1024 + 4) * 8 + 256);
  if (any_wide) {
    runs_rank_tables(v, owned);
    hip_check(fdb_launch_runs_flags_wide(v->phys, v->n_runs, v->segs, v->d_cols, v->d_rank32, (int)gcols_.size(), v->flags, (unsigned int*)(d_totals + 2), stream_), "runs flags (wide)");
    hip_check(fdb_launch_scan_u32(v->flags, 1, v->out_idx, v->n_runs, d_scratch2, d_totals + 1, stream_), "scan run flags");
    hip_check(hipMemcpyAsync(h_tot, d_totals, 24, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(group count)");
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    v->n_groups = (int64_t)h_tot[1];
    return (h_tot[2] & 0xFFFFFFFFull) == 0;
  }
  // rank of every key id among its column's values (bytes ascending; NULL — id 0 — last: cursorHeap.Less, arrowutils/merge.go:84-112)
  std::vector<unsigned char> rank(gcols_.size() * 256, 255);
  for (size_t c = 0; c < gcols_.size(); c++) {
    const GroupColState& g = gcols_[c];
    std::vector<uint32_t> order(g.values.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return g.values[x] < g.values[y]; });
    for (size_t r = 0; r < order.size(); r++) rank[c * 256 + order[r] + 1] = (unsigned char)r;
  }
  const unsigned char* d_rank = (const unsigned char*)upload(rank.data(), rank.size());
  ctx_->flush_staging();
  hip_check(fdb_launch_runs_flags(v->phys, v->n_runs, v->segs, d_rank, (int)gcols_.size(), v->flags, (unsigned int*)(d_totals + 2), stream_), "runs flags");
  hip_check(fdb_launch_scan_u32(v->flags, 1, v->out_idx, v->n_runs, d_scratch2, d_totals + 1, stream_), "scan run flags");
  hip_check(hipMemcpyAsync(h_tot, d_totals, 24, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(group count)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  v->n_groups = (int64_t)h_tot[1];
  return (h_tot[2] & 0xFFFFFFFFull) == 0;
}

// 32-bit rank tables (rank of every key id among its column's values, NULL — id 0 — last) and the columns' places, on the device.
void Plan::runs_rank_tables(RunsView* v, std::vector<void*>* owned) {
  if (v->d_cols != nullptr) return;
  std::vector<FdbRunCol> rc(gcols_.size());
  std::vector<uint32_t> rank32;
  for (size_t c = 0; c < gcols_.size(); c++) {
    const GroupColState& g = gcols_[c];
    rc[c].kind = g.kind == 0 ? 0 : g.is_u64 ? 3 : 1; rc[c].word = g.word; rc[c].gi = (int32_t)c; rc[c].rank_off = (uint32_t)rank32.size();
    if (g.kind != 0) continue;
    const size_t off = rank32.size();
    rank32.resize(off + g.values.size() + 1);
    rank32[off] = 0xFFFFFFFFu;
    if (const HostDict* whole = g.whole_dictionary(); whole != nullptr && whole->unique) {  // the ranks of an interned dictionary are computed once per process
      const std::vector<uint32_t>& r = whole->sorted_ranks();
      std::copy(r.begin(), r.end(), rank32.begin() + (ptrdiff_t)off + 1);
      continue;
    }
    std::vector<uint32_t> order(g.values.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    // (a dictionary whose values are already in order — what a writer that sorts its dictionary pages produces — needs no sort: 65 532 values × 8
    // columns were 20 ms of a Finish)
    if (!std::is_sorted(g.values.begin(), g.values.end())) std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return g.values[x] < g.values[y]; });
    for (size_t r = 0; r < order.size(); r++) rank32[off + order[r] + 1] = (uint32_t)r;
  }
  uint32_t* d_rank32 = (uint32_t*)ctx_->dev_alloc(rank32.size() * 4 + 256);
  owned->push_back(d_rank32);
  FdbRunCol* d_rc = (FdbRunCol*)ctx_->dev_alloc(rc.size() * sizeof(FdbRunCol) + 256);
  owned->push_back(d_rc);
  if (!rank32.empty()) hip_check(hipMemcpyAsync(d_rank32, rank32.data(), rank32.size() * 4, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(rank tables)");
  hip_check(hipMemcpyAsync(d_rc, rc.data(), rc.size() * sizeof(FdbRunCol), hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(run columns)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");  // (the host vectors above must outlive their copies)
  v->d_cols = d_rc;
  v->d_rank32 = d_rank32;
}

// The order of `n` things — runs of a run store (`segs`) or dense key rows (`rows`, `row_kw` words each) — by the plan's group columns:
// `order` (device, n × u64: the initial order, clobbered) and the returned pointer (one of the two buffers of the sort) hold phys / row
// numbers. Stable: things with equal keys keep their initial order. A radix sort of (key, number) pairs per ≤ 64 bits of key, the LAST group
// column first: dictionary columns by the rank of their ids, packed side by side; an int64 column by value, then by "is NULL" (NULLs last).
unsigned long long* Plan::sort_by_group_columns(unsigned long long* order, int64_t n_things, const FdbRunSegs* segs, const uint32_t* rows, int row_kw, RunsView* tables,
                                                std::vector<void*>* owned) {
  auto alloc = [&](size_t bytes) { void* p = ctx_->dev_alloc(std::max<size_t>(bytes, 256)); owned->push_back(p); return p; };
  runs_rank_tables(tables, owned);
  // the passes, least significant first
  std::vector<FdbRunKeyPass> passes;
  std::vector<int> pass_bits;
  FdbRunKeyPass cur;
  std::memset(&cur, 0, sizeof(cur));
  int cur_bits = 0;
  auto flush = [&] {
    if (cur.n > 0 && cur_bits > 0) { passes.push_back(cur); pass_bits.push_back(cur_bits); }
    std::memset(&cur, 0, sizeof(cur));
    cur_bits = 0;
  };
  for (size_t k = gcols_.size(); k-- > 0;) {
    const GroupColState& g = gcols_[k];
    if (g.kind != 0) {
      flush();
      FdbRunKeyPass p;
      std::memset(&p, 0, sizeof(p));
      p.n = 1; p.col[0] = (int32_t)k;
      p.mode = 1; passes.push_back(p); pass_bits.push_back(64);
      p.mode = 2; passes.push_back(p); pass_bits.push_back(1);
      continue;
    }
    const uint64_t n_values = g.values.size();  // ranks 0 … n_values − 1, NULL = n_values
    int bits = 0;
    while (bits < 33 && (n_values >> bits) != 0) bits++;
    if (bits == 0) continue;  // (a column without values: everything holds NULL there)
    if (cur_bits + bits > 64) flush();
    cur.mode = 0;
    cur.col[cur.n] = (int32_t)k; cur.shift[cur.n] = cur_bits; cur.null_rank[cur.n] = (uint32_t)n_values;
    cur.n++;
    cur_bits += bits;
  }
  flush();
  const size_t n = (size_t)n_things;
  unsigned long long* keys_a = (unsigned long long*)alloc(n * 8);
  unsigned long long* keys_b = (unsigned long long*)alloc(n * 8);
  unsigned long long* phys_a = order;
  unsigned long long* phys_b = (unsigned long long*)alloc(n * 8);
  size_t temp_bytes = 0;
  hip_check(fdb_sort_pairs_u64(nullptr, &temp_bytes, keys_a, keys_b, phys_a, phys_b, n_things, 64, stream_), "sort scratch size");
  void* temp = alloc(temp_bytes);
  for (size_t p = 0; p < passes.size(); p++) {
    hip_check(fdb_launch_runs_sort_keys(phys_a, n_things, segs, rows, row_kw, tables->d_cols, tables->d_rank32, passes[p], keys_a, stream_), "sort keys");
    size_t tb = temp_bytes;
    hip_check(fdb_sort_pairs_u64(temp, &tb, keys_a, keys_b, phys_a, phys_b, n_things, pass_bits[p], stream_), "sort pairs");
    std::swap(phys_a, phys_b);
  }
  return phys_a;
}

// Several ordered sets in one run store (≙ the k-way merge of OrderedAggregate's sets at Finish, ordered_aggregate.go:449-470; cursorHeap.Less,
// arrowutils/merge.go:84-112): the runs are sorted by key on the device — a stable LSD radix sort of (key, run) pairs over groups of
// columns, the last group column first, dictionary columns by the RANK of their ids packed as many to a 64-bit key as fit, int64 columns
// by value and then by "is NULL" — and the flags / prefix sums are taken again over the sorted order; Finish then proceeds as if one
// ordered set had arrived. Runs of one key end up next to each other in arrival order (the sort is stable), so the expand kernel folds
// them as it folds the runs that wave and record boundaries cut. Cost ∝ runs × (key bits / 64), not ∝ groups × log groups on the host.
bool Plan::runs_sort(RunsView* v, std::vector<void*>* owned) {
  const bool off = knobs_.runs_no_sort;  // (A/B and test aid: the table + host sort fallback)
  if (off || v->n_runs < 2 || v->n_runs > ((int64_t)1 << 28)) return false;  // (48 bytes of sort buffers per run: 12 GiB at the limit)
  auto alloc = [&](size_t bytes) { void* p = ctx_->dev_alloc(std::max<size_t>(bytes, 256)); owned->push_back(p); return p; };
  const size_t n = (size_t)v->n_runs;
  unsigned long long* phys_a = sort_by_group_columns(v->phys, v->n_runs, &v->segs, nullptr, 0, v, owned);
  v->phys = phys_a;
  if (v->flags == nullptr) v->flags = (uint32_t*)alloc(n * 4);
  if (v->out_idx == nullptr) v->out_idx = (uint32_t*)alloc(n * 4);
  unsigned long long* d_scratch = (unsigned long long*)alloc((n / 1024 + 4) * 8 + 256);
  unsigned long long* d_totals = (unsigned long long*)alloc(256);  // [1] groups, [2] order violation (low word)
  hip_check(hipMemsetAsync(d_totals, 0, 256, stream_), "hipMemsetAsync(totals)");
  hip_check(fdb_launch_runs_flags_wide(v->phys, v->n_runs, v->segs, v->d_cols, v->d_rank32, (int)gcols_.size(), v->flags, (unsigned int*)(d_totals + 2), stream_), "runs flags (sorted)");
  hip_check(fdb_launch_scan_u32(v->flags, 1, v->out_idx, v->n_runs, d_scratch, d_totals + 1, stream_), "scan run flags");
  unsigned long long h_tot[3] = {0, 0, 0};
  hip_check(hipMemcpyAsync(h_tot, d_totals, 24, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(group count)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  if ((h_tot[2] & 0xFFFFFFFFull) != 0) throw Error(FDB_ERR_DEVICE, "internal: the run store is not in key order after its sort");
  v->n_groups = (int64_t)h_tot[1];
  last_kernel_ = "runs_sort_keys_kernel + runs_expand_kernel";  // (what Finish ran: tests and the bench line name the path by it)
  return true;
}

// Two ordered plans merged AS RUNS (≙ the OrderedSynchronizer's k-way merge of the chains' sorted records, ordered_synchronizer.go:59-116,
// and the ordered aggregate's merge of ordered sets, ordered_aggregate.go:449-470): the source's runs are re-keyed into this plan's key ids
// (dictionaries are per plan: first-seen order) and column order and become one more segment of this plan's run store — one more ordered
// set. Finish brings the sets into one key order with the run store's own sort and folds equal keys; no hash table is built.
// false: not applicable (either side holds a table, the segments are used up, the sort is switched off) — the caller merges through the table.
bool Plan::merge_runs(Plan& src) {
  if (!ordered_ || !src.ordered_ || src.runs_.empty() || knobs_.runs_no_sort) return false;
  // (a plan that holds runs keeps nothing in its dense table — runs_to_table relies on the same — and has a hash table only after a fall-back)
  if (src.mode_ != TableMode::DENSE && src.h_table_ != nullptr) return false;
  if (runs_.empty() ? (state_dirty_ || h_table_ != nullptr) : (mode_ != TableMode::DENSE && h_table_ != nullptr)) return false;
  if (runs_.size() + 1 > FDB_MAX_RUN_SEGMENTS || aggs_.size() != 1 || src.aggs_.size() != 1) return false;
  hip_check(hipSetDevice(device_), "hipSetDevice");
  // adopt the source's columns and dictionary values
  std::vector<int> dst_of(src.gcols_.size());
  std::vector<std::vector<uint32_t>> id_map(src.gcols_.size());
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
    const GroupColState& sg = src.gcols_[sc];
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == sg.name) break;
    if (gi == gcols_.size()) {
      if (gcols_.size() >= FDB_MAX_HASH_GCOLS) throw Error(FDB_ERR_UNSUPPORTED, "too many group columns");
      GroupColState g;
      g.name = sg.name; g.kind = sg.kind; g.is_bool = sg.is_bool; g.is_u64 = sg.is_u64; g.plain = sg.plain; g.value_format = sg.value_format; g.cap = 1; g.stride = 0;
      gcols_.push_back(std::move(g));
    }
    GroupColState& g = gcols_[gi];
    if (g.kind != sg.kind || g.plain != sg.plain || g.is_bool != sg.is_bool || g.is_u64 != sg.is_u64) throw Error(FDB_ERR_INVALID, "group column " + sg.name + " has different types in the two plans");
    dst_of[sc] = (int)gi;
    if (sg.kind == 0) {
      g.owners.insert(g.owners.end(), sg.owners.begin(), sg.owners.end());
      bool identity = true;
      id_map[sc].assign(sg.values.size() + 1, 0);
      for (size_t v = 0; v < sg.values.size(); v++) { id_map[sc][v + 1] = g.intern(sg.values[v]); identity = identity && id_map[sc][v + 1] == (uint32_t)(v + 1); }
      if (identity) id_map[sc].clear();
    }
  }
  hash_layout();      // (no table on either side: assigns the wide tuple's words)
  src.hash_layout();
  // the record the new segment holds: the narrowest one this plan's id space allows
  int fmt = 0;
  if (gcols_.size() > FDB_RUN_TUPLE_BYTES) fmt = 2;
  for (const GroupColState& g : gcols_) {
    if (g.kind != 0 || g.values.size() > 65534) { fmt = 2; break; }
    if (g.values.size() > 254) fmt = std::max(fmt, 1);
  }
  if (knobs_.runs_wide == '1') fmt = 2; else if (knobs_.runs_wide == 'm') fmt = std::max(fmt, 1);
  const int out_rw = fmt == 0 ? 0 : fmt == 1 ? FDB_RUN_MEDIUM_WORDS : h_key_words_ + 4;
  const size_t rec_bytes = fmt == 0 ? (size_t)FDB_RUN_BYTES : fmt == 1 ? (size_t)FDB_RUN_MEDIUM_BYTES : (size_t)out_rw * 4;
  // the source's runs in their logical order
  std::vector<void*> owned;
  struct FreeOwned { Context* c; std::vector<void*>* v; hipStream_t a, b; ~FreeOwned() { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); for (void* p : *v) c->dev_free(p); } } free_owned{src.ctx_, &owned, src.stream_, stream_};
  RunsView v;
  src.runs_prepare(&v, /*check_order=*/false, &owned);
  src.sync();  // (the map of its runs is the last thing the source's stream wrote: our stream reads it next)
  if (v.n_runs == 0) return true;
  std::vector<FdbHashCol> cols(gcols_.size());
  for (size_t c = 0; c < gcols_.size(); c++) {
    std::memset(&cols[c], 0, sizeof(FdbHashCol));
    cols[c].kind = gcols_[c].kind; cols[c].word = gcols_[c].word; cols[c].gi = (int)c; cols[c].src_word = -1; cols[c].lut_lds = FDB_NO_LDS;
  }
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
    FdbHashCol& C = cols[(size_t)dst_of[sc]];
    C.src_word = src.gcols_[sc].word;
    C.lut_len = (uint32_t)sc;
    if (src.gcols_[sc].kind == 0 && !id_map[sc].empty()) C.lut = (const uint32_t*)upload(id_map[sc].data(), id_map[sc].size() * 4);
  }
  RunSegment seg;
  seg.n_entries = (v.n_runs + 255) / 256;
  seg.capacity = v.n_runs;
  seg.run_words = out_rw;
  const size_t tuples_bytes = align_up_sz((size_t)v.n_runs * rec_bytes, 256), dir_bytes = align_up_sz((size_t)seg.n_entries * 8, 256);
  seg.block = ctx_->dev_alloc(tuples_bytes + dir_bytes + 256);
  seg.tuples = (unsigned char*)seg.block;
  seg.dir = (uint32_t*)(seg.tuples + tuples_bytes); seg.cursor = (uint32_t*)(seg.tuples + tuples_bytes + dir_bytes);
  runs_.push_back(seg);
  FdbRunsTranslateArgs t;
  std::memset(&t, 0, sizeof(t));
  t.phys = v.phys; t.n_runs = v.n_runs;
  t.cols = (const FdbHashCol*)upload(cols.data(), cols.size() * sizeof(FdbHashCol));
  t.out_tuples = seg.tuples; t.out_dir = seg.dir;
  t.n_cols = (int)gcols_.size(); t.out_run_words = out_rw;
  hip_check(fdb_launch_runs_translate(t, v.segs, stream_), "runs translate");
  ctx_->flush_staging();
  last_kernel_ = "runs_translate_kernel";
  state_dirty_ = true;
  sync();  // the caller may close `src` when this returns
  return true;
}

// Every run becomes a pre-aggregated entry of the hash table (equal keys merge there): what any consumer other than Finish sees,
// and where input that was not ordered ends up.
void Plan::runs_to_table() {
  if (runs_.empty()) return;
  hip_check(hipSetDevice(device_), "hipSetDevice");
  std::vector<void*> owned;
  struct FreeOwned { Context* c; std::vector<void*>* v; hipStream_t s; ~FreeOwned() { (void)hipStreamSynchronize(s); for (void* p : *v) c->dev_free(p); } } free_owned{ctx_, &owned, stream_};
  RunsView v;
  runs_prepare(&v, /*check_order=*/false, &owned);
  std::vector<RunSegment> segs;
  segs.swap(runs_);  // (switch_to_hash / fetch paths below must not come back here)
  struct FreeSegs { Context* c; std::vector<RunSegment>* v; hipStream_t s; ~FreeSegs() { (void)hipStreamSynchronize(s); for (RunSegment& r : *v) c->dev_free(r.block); } } free_segs{ctx_, &segs, stream_};
  if (mode_ == TableMode::DENSE) { state_dirty_ = false; switch_to_hash(); }
  hash_layout();
  if (v.n_runs == 0) return;
  if (h_count_dev_ != nullptr) hash_groups();
  hash_reserve((uint64_t)v.n_runs);
  const int kw = h_key_words_;
  uint32_t* d_keys = (uint32_t*)ctx_->dev_alloc((size_t)v.n_runs * kw * 4 + 256);
  owned.push_back(d_keys);
  unsigned long long* d_entries = (unsigned long long*)ctx_->dev_alloc((size_t)v.n_runs * 16 + 256);
  owned.push_back(d_entries);
  // entries are {count, acc} pairs: expand with every run its own group writes them through two strided views
  FdbRunsExpandArgs x;
  std::memset(&x, 0, sizeof(x));
  x.phys = v.phys; x.flags = nullptr; x.out_idx = nullptr; x.n_runs = v.n_runs;
  x.dense_keys = d_keys; x.vals_cnt = d_entries; x.vals_acc = d_entries + 1; x.val_stride = 2;
  x.n_cols = (int)gcols_.size(); x.key_words = kw; x.func = runs_func();
  for (size_t c = 0; c < gcols_.size() && c < FDB_RUN_TUPLE_BYTES; c++) x.col_word[c] = gcols_[c].word;
  hip_check(fdb_launch_runs_expand(x, v.segs, stream_), "runs expand");
  std::vector<FdbHashCol> cols(gcols_.size());
  for (size_t c = 0; c < gcols_.size(); c++) {
    std::memset(&cols[c], 0, sizeof(FdbHashCol));
    cols[c].kind = gcols_[c].kind; cols[c].word = gcols_[c].word; cols[c].gi = (int)c; cols[c].lut_lds = FDB_NO_LDS;
    cols[c].k1 = fdb_fp_k1((int)c); cols[c].k2 = fdb_fp_k2((int)c);
    cols[c].src_word = gcols_[c].word;  // the expanded rows already have the table's own tuple layout
    if (gcols_[c].kind != 0) cols[c].lut_len = (uint32_t)c;  // (int64 / computed keys: the bit of the incoming valid mask — hash_merge_kernel reads it from lut_len)
  }
  hash_merge_device(d_entries, d_keys, v.n_runs, kw, cols, /*unique_source=*/false);  // (a key may come back in a later run)
  ctx_->flush_staging();
  h_groups_bound_ += (uint64_t)v.n_runs;
  h_bound_stale_ = true;
  state_dirty_ = true;
}

// Finish straight from the run store. *ok = false: the keys were not in order — the runs are in the hash table now and the caller
// takes the ordinary ordered Finish.
int64_t Plan::finish_columns_runs(std::vector<OutColumn>* cols, DeviceBatch* resident, bool* ok) {
  *ok = false;
  std::vector<void*> owned;
  struct FreeOwned { Context* c; std::vector<void*>* v; hipStream_t s; ~FreeOwned() { (void)hipStreamSynchronize(s); for (void* p : *v) c->dev_free(p); } } free_owned{ctx_, &owned, stream_};
  hash_layout();
  RunsView v;
  // keys out of order — several ordered sets, or input that was not ordered at all: sorted on the device; beyond what that takes, the table
  if (!runs_prepare(&v, /*check_order=*/true, &owned) && !runs_sort(&v, &owned)) { runs_to_table(); return 0; }
  const int64_t n = finish_columns_hash(cols, resident, &v);
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  runs_free();
  finished_ = true;
  *ok = true;
  return n;
}

}  // namespace fdb
