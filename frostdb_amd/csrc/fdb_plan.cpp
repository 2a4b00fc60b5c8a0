// fdb_plan.cpp — host side of the fused PredicateFilter → HashAggregate chain (see fdb_plan.h).
//
// Per batch the host does only per-DICTIONARY-ENTRY work (predicate LUTs, group-key id LUTs) and argument
// marshalling; every per-ROW operation runs in fdb_kernels.hip. Reference behaviour mirrored here:
//   * column lookup by exact name, every batch                 aggregate.go:286-361, binaryscalarexpr.go:22-29
//   * missing-column predicate rules                           binaryscalarexpr.go:47-73, regexpfilter.go:23-33
//   * dictionary ==/!=, NULL literal ⇒ IS [NOT] NULL           binaryscalarexpr.go:154-232
//   * contains / regex evaluated on dictionary values          binaryscalarexpr.go:271-311, regexpfilter.go:142-166
//   * group identity: NULL ≡ column absent, first-seen order   aggregate.go:398-409, :492-525, :568-575
//   * result naming and schema                                 aggregate.go:47, :543-633
#include "fdb_plan.h"
#include "fdb_jit.h"

#include "fdb_context.h"
#include "fdb_hostpool.h"
#include "fdb_plan_internal.h"

#include <algorithm>
#include <cstring>
#include <functional>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace fdb {


void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error(e == hipErrorOutOfMemory ? FDB_ERR_OOM : FDB_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t kTailPad = 256;  // bytes readable past every column so tail lanes may over-read
}
void copy_stream(void* dst, const void* src, size_t n);                          // fdb_widen.cc: streaming-store copies into the pinned slab
uint32_t copy_stream_max_u32(uint32_t* dst, const uint32_t* src, size_t n);
namespace {

const char* op_str(int32_t op) {  // logicalplan/expr.go:37-72
  switch (op) {
    case FDB_OP_EQ: return "=="; case FDB_OP_NOT_EQ: return "!="; case FDB_OP_LT: return "<"; case FDB_OP_LT_EQ: return "<=";
    case FDB_OP_GT: return ">"; case FDB_OP_GT_EQ: return ">="; case FDB_OP_REGEX_MATCH: return "=~";
    case FDB_OP_REGEX_NOT_MATCH: return "!~"; case FDB_OP_AND: return "&&"; case FDB_OP_OR: return "||";
    case FDB_OP_CONTAINS: return "contains"; case FDB_OP_NOT_CONTAINS: return "not contains";
  }
  return "?";
}

const char* agg_name(int32_t f) {  // logicalplan/expr.go:731-750
  switch (f) {
    case FDB_AGG_SUM: return "sum"; case FDB_AGG_MIN: return "min"; case FDB_AGG_MAX: return "max"; case FDB_AGG_COUNT: return "count";
    case FDB_AGG_AVG: return "avg"; case FDB_AGG_UNIQUE: return "unique"; case FDB_AGG_AND: return "and";
  }
  return "unknown";
}

bool is_leaf_op(int32_t op) {
  return (op >= FDB_OP_EQ && op <= FDB_OP_REGEX_NOT_MATCH) || op == FDB_OP_CONTAINS || op == FDB_OP_NOT_CONTAINS;
}

bool match_group(const GroupMatcher& m, const std::string& field) {  // logicalplan/expr.go:353-355, :564-566
  if (m.dynamic) return field.size() > m.name.size() && field.compare(0, m.name.size(), m.name) == 0 && field[m.name.size()] == '.';
  return field == m.name;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// DeviceBatch
// ---------------------------------------------------------------------------------------------------------
void DeviceBatch::note_reader(hipStream_t s) const {
  if (arena_ctx != nullptr) return;  // transient record: its arena returns to the cache of the very stream that reads it
  std::lock_guard<std::mutex> lk(readers_mu_);
  for (hipStream_t r : readers_) if (r == s) return;
  readers_.push_back(s);
}

DeviceBatch::~DeviceBatch() {
  if ((arena == nullptr && extra_arenas.empty()) || arena_borrowed) return;
  if (arena_ctx != nullptr) { arena_ctx->dev_free(arena); return; }
  if (!readers_.empty()) {  // (an idle stream answers in about a microsecond)
    (void)hipSetDevice(device);
    for (hipStream_t s : readers_) (void)hipStreamSynchronize(s);
  }
  if (arena != nullptr) device_pool_free(device, arena);
  for (void* p : extra_arenas) device_pool_free(device, p);
}

int DeviceBatch::find(const std::string& name) const {
  int found = -1;
  for (size_t i = 0; i < cols.size(); i++)
    if (cols[i].name == name) { if (found >= 0) return -1; found = (int)i; }
  return found;
}

namespace {
// Dictionary indices of valid rows must be below the dictionary's length (Arrow's own rule). The scan kernels index LUTs with
// them, so a malformed record is refused here — FDB_ERR_INVALID, ≙ the reference's recovered panic (recovery/recovery.go:13-30) —
// instead of reading out of bounds on the device. Host version (small records, which are copied through the pinned ring by the
// CPU anyway): one vectorisable max over all rows; only if that fails, a second pass that skips NULL rows (whose slots may
// hold anything).
void check_indices_host(const uint32_t* idx, const uint8_t* validity_bit0, int64_t n, size_t limit, const std::string& name) {
  uint32_t mx = 0;
  for (int64_t i = 0; i < n; i++) mx = idx[i] > mx ? idx[i] : mx;
  if (n == 0 || (size_t)mx < limit) return;
  for (int64_t i = 0; i < n; i++)
    if ((size_t)idx[i] >= limit && (validity_bit0 == nullptr || ((validity_bit0[i >> 3] >> (i & 7)) & 1)))
      throw Error(FDB_ERR_INVALID, "dictionary index out of range in column " + name + ": row " + std::to_string(i) + " holds " + std::to_string(idx[i]) +
                                       ", the dictionary has " + std::to_string(limit) + " entries");
}
}  // namespace

std::unique_ptr<DeviceBatch> import_batch(const HostRecordView& view, int device, const std::function<bool(const std::string&)>* want,
                                          hipStream_t stream, Context* ctx, bool via_ring, const RecordSink* sink) {
  std::unique_ptr<DeviceBatch> b(new DeviceBatch());
  b->device = device;
  b->rows = view.rows;
  hip_check(hipSetDevice(device), "hipSetDevice");
  static const bool prof = std::getenv("FDB_PROFILE_PUSH") != nullptr;  // (tuning aid: phases of a record's import, summed per process)
  static std::atomic<int64_t> p_ns[4], p_n;
  auto tick = [] { return std::chrono::steady_clock::now(); };
  auto t_prev = tick();
  auto lap = [&](int k) { if (prof) { const auto n = tick(); p_ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t_prev).count(); t_prev = n; } };
  struct Report { ~Report() { if (prof && p_n.load() > 0) std::fprintf(stderr, "[fdb] import of %lld records: plan+dictionaries %.2f, reserve %.2f, copies %.2f, index check %.2f us per record\n", (long long)p_n.load(),
                                                                      p_ns[0] / 1e3 / p_n, p_ns[1] / 1e3 / p_n, p_ns[2] / 1e3 / p_n, p_ns[3] / 1e3 / p_n); } };
  static Report report;
  // plan the arena
  struct Piece { size_t col; bool validity; size_t off; size_t bytes; };
  std::vector<Piece> pieces;
  std::vector<std::pair<size_t, std::vector<uint32_t>>> plain_idx;  // (column, encoded indices) of plain string columns
  size_t total = 0;
  for (size_t i = 0; i < view.cols.size(); i++) {
    const HostColView& c = view.cols[i];
    DevColumn d;
    d.name = c.name; d.format = c.format; d.kind = c.kind; d.length = c.length; d.null_count = c.null_count;
    const bool staged = (want == nullptr || (*want)(c.name));
    if (staged) {
      int64_t vb = 0;
      if (c.kind == ColKind::I64 || c.kind == ColKind::U64 || c.kind == ColKind::F64 || c.kind == ColKind::BOOL) vb = c.length * 8;  // (bool: widened to int64 1 / 2)
      else if (c.kind == ColKind::DICT || c.kind == ColKind::STR) vb = c.length * 4;  // (plain strings: encoded below, one uint32 per row)
      if (vb > 0 || ((c.kind == ColKind::I64 || c.kind == ColKind::U64 || c.kind == ColKind::F64 || c.kind == ColKind::DICT || c.kind == ColKind::BOOL || c.kind == ColKind::STR))) {
        d.value_bytes = c.kind == ColKind::BOOL ? (c.length + 7) / 8 : vb;  // (algorithmic bytes: Arrow's bit-packed buffer)
        pieces.push_back(Piece{i, false, total, (size_t)vb});
        total += align_up((size_t)vb + kTailPad, 256);
        if (c.null_count > 0) {
          d.validity_bytes = (c.length + 7) / 8;
          pieces.push_back(Piece{i, true, total, (size_t)d.validity_bytes});
          total += align_up((size_t)d.validity_bytes + kTailPad, 256);
        }
      }
      if (c.kind == ColKind::DICT) d.dict = read_dictionary(c);
      if (c.kind == ColKind::STR) {  // from here on a dictionary column whose dictionary says `plain`
        plain_idx.emplace_back(i, std::vector<uint32_t>());
        d.dict = encode_plain(c, &plain_idx.back().second);
        d.kind = ColKind::DICT;
      }
    } else {
      d.kind = c.kind;  // present but not staged: d_values stays nullptr
    }
    b->cols.push_back(std::move(d));
  }
  lap(0);
  unsigned char* sink_host = nullptr;
  if (total > 0) {
    if (sink != nullptr) { (*sink)(total, &b->arena, &sink_host); b->arena_borrowed = true; }
    else if (ctx != nullptr) { b->arena = ctx->dev_alloc(total); b->arena_ctx = ctx; }
    else b->arena = device_pool_alloc(device, total);
    b->arena_bytes = total;
  }
  lap(1);
  // transient batches: every copy is queued on `stream`; re-packed buffers stay alive until the one wait at the end
  std::vector<uint32_t> index_max(view.cols.size(), 0);
  std::vector<char> index_max_known(view.cols.size(), 0);
  std::vector<std::vector<uint8_t>> keep_bits;
  std::vector<std::vector<uint32_t>> keep_idx;
  std::vector<std::vector<int64_t>> keep_i64;
  // via_ring: the whole arena is assembled in ONE piece of the pinned ring and shipped with one DMA
  unsigned char* ring = sink_host != nullptr ? sink_host : (ctx != nullptr && via_ring && total > 0) ? ctx->copy_reserve(total) : nullptr;
  auto h2d = [&](void* dst, const void* src, size_t bytes, const char* what) {
    if (ring != nullptr) copy_stream(ring + ((unsigned char*)dst - (unsigned char*)b->arena), src, bytes);  // (splitting a 1 MB copy over pooled threads was measured: their wake-up costs more than it saves)
    else if (ctx != nullptr) hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream), what);
    else hip_check(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), what);
  };
  for (const Piece& p : pieces) {
    const HostColView& c = view.cols[p.col];
    DevColumn& d = b->cols[p.col];
    unsigned char* dst = (unsigned char*)b->arena + p.off;
    if (p.validity) {
      d.d_validity = dst;
      if (p.bytes == 0) continue;
      if ((c.offset & 7) == 0) {  // byte-aligned bitmap: copied as is (bits past `length` are never looked at)
        h2d(dst, c.validity + (c.offset >> 3), p.bytes, "hipMemcpy(validity)");
      } else {
        keep_bits.emplace_back(p.bytes, 0);
        copy_bits(c.validity, c.offset, c.length, keep_bits.back().data());
        h2d(dst, keep_bits.back().data(), p.bytes, "hipMemcpy(validity)");
      }
    } else {
      d.d_values = dst;
      if (p.bytes == 0) continue;
      if (c.kind == ColKind::BOOL) {
        // Arrow booleans are bit-packed; on the device a bool is an int64 holding 1 (false) or 2 (true) — the reference's own
        // hash of a bool key (dynparquet/hashed.go:228-242; 0 is NULL), so a bool group key is an ordinary int64 key, a filter
        // leaf an int64 compare and AND a MIN
        keep_i64.emplace_back((size_t)c.length);
        std::vector<int64_t>& wide = keep_i64.back();
        const uint8_t* bits = (const uint8_t*)c.values;
        for (int64_t i = 0; i < c.length; i++) wide[(size_t)i] = 1 + ((bits[(c.offset + i) >> 3] >> ((c.offset + i) & 7)) & 1);
        h2d(dst, wide.data(), p.bytes, "hipMemcpy(bool values)");
      } else if (c.kind == ColKind::STR) {
        const std::vector<uint32_t>* enc = nullptr;
        for (const auto& pi : plain_idx) if (pi.first == p.col) enc = &pi.second;
        h2d(dst, enc->data(), p.bytes, "hipMemcpy(encoded strings)");
      } else if (c.kind == ColKind::DICT && c.index_width != 4) {
        keep_idx.emplace_back((size_t)c.length);
        std::vector<uint32_t>& wide = keep_idx.back();
        for (int64_t i = 0; i < c.length; i++) {
          switch (c.index_width) {
            case 1: wide[(size_t)i] = ((const uint8_t*)c.values)[c.offset + i]; break;
            case 2: wide[(size_t)i] = ((const uint16_t*)c.values)[c.offset + i]; break;
            default: wide[(size_t)i] = (uint32_t)((const uint64_t*)c.values)[c.offset + i]; break;
          }
        }
        h2d(dst, wide.data(), p.bytes, "hipMemcpy(indices)");
      } else if (ring != nullptr && c.kind == ColKind::DICT && d.dict && !d.dict->plain) {
        // indices into the pinned piece and their maximum in ONE pass (the separate validation pass re-read what had just been written: 27 µs
        // of a 65 536-row record's 85)
        const uint32_t* src = (const uint32_t*)c.values + c.offset;
        uint32_t* out = (uint32_t*)(ring + (dst - (unsigned char*)b->arena));
        index_max[p.col] = copy_stream_max_u32(out, src, (size_t)c.length);
        index_max_known[p.col] = 1;
      } else {
        const size_t w = c.kind == ColKind::DICT ? 4 : 8;
        h2d(dst, (const unsigned char*)c.values + (size_t)c.offset * w, p.bytes, "hipMemcpy(values)");
      }
    }
  }
  for (const DevColumn& d : b->cols) b->payload_bytes += d.value_bytes + d.validity_bytes;
  lap(2);
  // dictionary indices are validated before anything can scan the record (see check_indices_host)
  std::vector<size_t> dict_cols;
  for (size_t i = 0; i < b->cols.size(); i++)
    if (b->cols[i].kind == ColKind::DICT && b->cols[i].d_values != nullptr && b->cols[i].dict && !b->cols[i].dict->plain && b->rows > 0) dict_cols.push_back(i);
  if (ring != nullptr) {
    for (size_t i : dict_cols) {  // the bytes are in the pinned ring: checked by the CPU that just copied them
      const DevColumn& d = b->cols[i];
      if (index_max_known[i] && (size_t)index_max[i] < d.dict->values.size()) continue;  // (the copy already saw every index)
      const unsigned char* base = ring;
      check_indices_host((const uint32_t*)(base + ((unsigned char*)d.d_values - (unsigned char*)b->arena)),
                         d.d_validity ? base + (d.d_validity - (unsigned char*)b->arena) : nullptr, b->rows, d.dict->values.size(), d.name);
    }
    if (sink_host == nullptr) ctx->copy_commit(b->arena, ring, total);  // (a sink's owner ships its slab itself)
    lap(3);
    if (prof) p_n++;
    return b;
  }
  if (!dict_cols.empty()) {  // big records: one streaming pass on the device behind the copies (4 B/row at HBM speed), one flag word per column
    uint32_t* d_flags = nullptr;
    if (ctx != nullptr) d_flags = (uint32_t*)ctx->dev_alloc(dict_cols.size() * 4);
    else hip_check(hipMalloc((void**)&d_flags, dict_cols.size() * 4), "hipMalloc(index check)");
    std::vector<uint32_t> flags(dict_cols.size(), 0);
    hipError_t e = hipMemsetAsync(d_flags, 0, dict_cols.size() * 4, stream);
    for (size_t k = 0; k < dict_cols.size() && e == hipSuccess; k++) {
      const DevColumn& d = b->cols[dict_cols[k]];
      e = fdb_launch_validate_indices((const uint32_t*)d.d_values, d.d_validity, b->rows, (uint32_t)std::min<size_t>(d.dict->values.size(), 0xFFFFFFFFu), d_flags + k, stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), d_flags, dict_cols.size() * 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (ctx != nullptr) ctx->dev_free(d_flags); else (void)hipFree(d_flags);
    hip_check(e, "dictionary index check");
    for (size_t k = 0; k < dict_cols.size(); k++)
      if (flags[k] != 0) {
        const DevColumn& d = b->cols[dict_cols[k]];
        throw Error(FDB_ERR_INVALID, "dictionary index out of range in column " + d.name + ": a valid row holds an index ≥ the dictionary's " +
                                         std::to_string(d.dict->values.size()) + " entries");
      }
  } else if (ctx != nullptr && (!keep_bits.empty() || !keep_idx.empty() || !keep_i64.empty() || !plain_idx.empty())) {
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(import)");
  }
  return b;
}

// ---------------------------------------------------------------------------------------------------------
// Plan
// ---------------------------------------------------------------------------------------------------------
std::string Literal::str() const {
  switch (type) {
    case FDB_LIT_NULL: return "null";
    case FDB_LIT_INT64: return std::to_string(i64);
    case FDB_LIT_UINT64: return std::to_string(u64);
    case FDB_LIT_FLOAT64: return std::to_string(f64);
    case FDB_LIT_BOOL: return i64 ? "true" : "false";
    default: return bytes;
  }
}

bool ExprNode::regex_matches(const std::string& v) const {
  if (re_fn == nullptr) return re->match(v);
  const int32_t r = re_fn(re_user, lit.bytes.data(), (int64_t)lit.bytes.size(), (const uint8_t*)v.data(), (int64_t)v.size());
  if (r < 0) throw Error(FDB_ERR_INVALID, "regexp: the host matcher rejected pattern " + lit.bytes);
  return r != 0;
}

void check_desc_shape(const fdb_plan_desc* d) {
  if (d == nullptr) throw Error(FDB_ERR_INVALID, "null plan descriptor");
  auto arr = [](int32_t n, const void* p, const char* what) {
    if (n < 0) throw Error(FDB_ERR_INVALID, std::string("negative ") + what + " count");
    if (n > 0 && p == nullptr) throw Error(FDB_ERR_INVALID, std::string(what) + " count without an array");
  };
  arr(d->n_filter, d->filter, "filter node");
  arr(d->n_aggs, d->aggs, "aggregation");
  arr(d->n_groups, d->groups, "group expression");
  arr(d->n_projections, d->projections, "projection");
  for (int32_t i = 0; i < d->n_projections; i++) {
    const fdb_projection& fp = d->projections[i];
    if (fp.name == nullptr || fp.nodes == nullptr || fp.n_nodes <= 0 || fp.root < 0 || fp.root >= fp.n_nodes) throw Error(FDB_ERR_INVALID, "malformed projection");
  }
  for (int32_t i = 0; i < d->n_filter; i++)
    if (d->filter[i].literal.len < 0) throw Error(FDB_ERR_INVALID, "literal with a negative length");
}

Plan::Plan(const fdb_plan_desc* d, int device, bool explain_only) : device_(device) {
  check_desc_shape(d);
  for (int32_t i = 0; i < d->n_filter; i++) {
    const fdb_expr& fe = d->filter[i];
    ExprNode e;
    e.op = fe.op; e.left = fe.left; e.right = fe.right;
    if (fe.column) e.column = fe.column;
    e.lit.type = fe.literal.type; e.lit.i64 = fe.literal.i64; e.lit.u64 = fe.literal.u64; e.lit.f64 = fe.literal.f64;
    if (fe.literal.data && fe.literal.len > 0) e.lit.bytes.assign(fe.literal.data, (size_t)fe.literal.len);
    if (e.op == FDB_OP_AND || e.op == FDB_OP_OR) {
      if (e.left < 0 || e.left >= d->n_filter || e.right < 0 || e.right >= d->n_filter) throw Error(FDB_ERR_INVALID, "filter child index out of range");
    } else if (is_leaf_op(e.op)) {
      if (e.column.empty()) throw Error(FDB_ERR_INVALID, "left side of binary expression must be a column");  // filter.go:91-93
      if (e.op == FDB_OP_REGEX_MATCH || e.op == FDB_OP_REGEX_NOT_MATCH) {
        if (e.lit.type != FDB_LIT_STRING && e.lit.type != FDB_LIT_BINARY) throw Error(FDB_ERR_INVALID, "regex literal must be a string");
        if (d->regex_match != nullptr) {
          e.re_fn = d->regex_match; e.re_user = d->regex_user;
          (void)e.regex_matches(std::string());  // surfaces a pattern that does not compile now, like regexp.Compile at plan build (filter.go:105-124)
        } else {
          std::string why;
          e.re = Regex::compile(e.lit.bytes, &why);  // compiled once per query, like regexp.Compile at plan build (filter.go:105-124)
          if (!e.re) throw Error(FDB_ERR_INVALID, why);
        }
      }
    } else {
      throw Error(FDB_ERR_UNSUPPORTED, std::string("binary expr ") + op_str(e.op) + ": unsupported boolean expression");  // filter.go:162-164
    }
    filter_.push_back(std::move(e));
  }
  filter_root_ = d->n_filter > 0 ? d->filter_root : -1;
  if (d->n_filter > 0 && (filter_root_ < 0 || filter_root_ >= d->n_filter)) throw Error(FDB_ERR_INVALID, "filter_root out of range");
  if (filter_root_ >= 0) {
    // The nodes may come in any order, but they must form a TREE below the root: a node that names itself or an ancestor as a
    // child would send every recursive walk over the expression (resolution, truth tables, Draw) into the ground.
    std::vector<char> state(filter_.size(), 0);  // 0 unvisited, 1 on the current path, 2 done
    std::function<void(int, int)> walk = [&](int i, int depth) {
      if (depth > 64) throw Error(FDB_ERR_UNSUPPORTED, "filter expression too deep");
      if (state[(size_t)i] == 1) throw Error(FDB_ERR_INVALID, "filter expression is not a tree: node " + std::to_string(i) + " is its own ancestor");
      if (state[(size_t)i] == 2) throw Error(FDB_ERR_INVALID, "filter expression is not a tree: node " + std::to_string(i) + " has two parents");
      state[(size_t)i] = 1;
      const ExprNode& e = filter_[(size_t)i];
      if (e.op == FDB_OP_AND || e.op == FDB_OP_OR) { walk(e.left, depth + 1); walk(e.right, depth + 1); }
      state[(size_t)i] = 2;
    };
    walk(filter_root_, 0);
  }
  final_stage_ = d->final_stage != 0;
  ordered_ = d->ordered != 0;
  if (ordered_ && d->n_aggs != 1)  // the operator takes ONE Aggregation (NewOrderedAggregate, ordered_aggregate.go:116-122; shouldPlanOrderedAggregate, physicalplan.go:525-528)
    throw Error(FDB_ERR_UNSUPPORTED, "OrderedAggregate: exactly one aggregation is supported");
  for (int32_t i = 0; i < d->n_aggs; i++) {
    AggState a;
    a.func = d->aggs[i].func;
    if (d->aggs[i].column == nullptr) throw Error(FDB_ERR_INVALID, "aggregation without a column");
    if (d->aggs[i].dynamic != 0)  // (fdb_plan_create expands these into a family of plans, fdb_dynamic.h; one Plan never sees them)
      throw Error(FDB_ERR_INVALID, std::string("internal: aggregation over the dynamic column set ") + d->aggs[i].column + ".* reached a single plan");
    a.column = d->aggs[i].column;
    a.result_name = std::string(agg_name(a.func)) + "(" + a.column + ")";
    a.emit_name = (ordered_ && !final_stage_) ? a.column : a.result_name;  // OrderedAggregate.getResultColumnName (ordered_aggregate.go:551-557)
    if (a.func == FDB_AGG_UNIQUE) {  // two physical accumulators, see AggState::role
      AggState lo = a, hi = a;
      lo.func = FDB_AGG_MIN; lo.role = 1; lo.null_value = (unsigned long long)FDB_I64_MIN;
      hi.func = FDB_AGG_MAX; hi.role = 2; hi.null_value = (unsigned long long)FDB_I64_MAX;
      aggs_.push_back(std::move(lo));
      aggs_.push_back(std::move(hi));
    } else if (a.func == FDB_AGG_AND) {
      a.func = FDB_AGG_MIN; a.role = 3; a.null_value = 2ull;  // (a NULL row counts as true = 2)
      aggs_.push_back(std::move(a));
    } else if (a.func == FDB_AGG_SUM || a.func == FDB_AGG_MIN || a.func == FDB_AGG_MAX || a.func == FDB_AGG_COUNT) {
      aggs_.push_back(std::move(a));
    } else {
      throw Error(FDB_ERR_UNSUPPORTED, std::string("unsupported aggregation function: ") + agg_name(a.func));  // aggregate.go:98-100
    }
  }
  if (aggs_.size() > FDB_MAX_AGGS) throw Error(FDB_ERR_UNSUPPORTED, "too many aggregations");
  for (int32_t i = 0; i < d->n_groups; i++) {
    if (d->groups[i].name == nullptr) throw Error(FDB_ERR_INVALID, "group expression without a name");
    matchers_.push_back(GroupMatcher{d->groups[i].name, d->groups[i].dynamic != 0});
  }
  for (int32_t i = 0; i < d->n_projections; i++) {
    const fdb_projection& fp = d->projections[i];
    if (fp.name == nullptr || fp.nodes == nullptr || fp.n_nodes <= 0 || fp.root < 0 || fp.root >= fp.n_nodes) throw Error(FDB_ERR_INVALID, "malformed projection");
    if (final_stage_) throw Error(FDB_ERR_INVALID, "a final-stage plan has no pre-aggregate projection");
    Projection P;
    P.name = fp.name;
    P.root = fp.root;
    for (int32_t k = 0; k < fp.n_nodes; k++) {
      const fdb_proj_node& fn = fp.nodes[k];
      ProjNode n;
      n.kind = fn.kind; n.op = fn.op; n.left = fn.left; n.right = fn.right;
      if (fn.kind == 0) {
        if (fn.column == nullptr) throw Error(FDB_ERR_INVALID, "projection column node without a name");
        n.column = fn.column;
      } else if (fn.kind == 1) {
        // (int64 / float64 literals compute; a string / binary / NULL literal can only be the right side of a comparison with a column —
        // boolExprProjection evaluates its expression like a filter does, project.go:409-470: checked when the record is resolved)
        if (fn.literal.type != FDB_LIT_INT64 && fn.literal.type != FDB_LIT_FLOAT64 && fn.literal.type != FDB_LIT_UINT64 && fn.literal.type != FDB_LIT_STRING && fn.literal.type != FDB_LIT_BINARY &&
            fn.literal.type != FDB_LIT_NULL)
          throw Error(FDB_ERR_UNSUPPORTED, "projection literals must be int64, uint64, float64, string, binary or NULL");
        n.lit_type = fn.literal.type; n.i64 = fn.literal.type == FDB_LIT_UINT64 ? (int64_t)fn.literal.u64 : fn.literal.i64; n.f64 = fn.literal.f64;
        n.lit.type = fn.literal.type; n.lit.i64 = fn.literal.i64; n.lit.u64 = fn.literal.u64; n.lit.f64 = fn.literal.f64;
        if (fn.literal.data && fn.literal.len > 0) n.lit.bytes.assign(fn.literal.data, (size_t)fn.literal.len);
      } else if (fn.kind == 2 || fn.kind == 3) {
        if (fn.kind == 2 && (fn.op < FDB_OP_ADD || fn.op > FDB_OP_DIV)) throw Error(FDB_ERR_UNSUPPORTED, "unsupported binary expression in projection");  // project.go:122-123
        if (fn.kind == 3 && (fn.op < FDB_OP_EQ || fn.op > FDB_OP_GT_EQ) && fn.op != FDB_OP_AND && fn.op != FDB_OP_OR) throw Error(FDB_ERR_UNSUPPORTED, "unsupported comparison in projection");
        if (fn.left < 0 || fn.left >= k || fn.right < 0 || fn.right >= k) throw Error(FDB_ERR_INVALID, "projection nodes must be in post-order");
      } else if (fn.kind == 4 || fn.kind == 5) {  // convert(left, float64) / isnull(left)
        if (fn.left < 0 || fn.left >= k) throw Error(FDB_ERR_INVALID, "projection nodes must be in post-order");
        n.right = -1;
      } else if (fn.kind == 6) {  // if(cond = node `op`) { left } else { right }
        if (fn.left < 0 || fn.left >= k || fn.right < 0 || fn.right >= k || fn.op < 0 || fn.op >= k) throw Error(FDB_ERR_INVALID, "projection nodes must be in post-order");
      } else {
        throw Error(FDB_ERR_INVALID, "unknown projection node kind");
      }
      P.nodes.push_back(std::move(n));
    }
    projs_.push_back(std::move(P));
  }
  if (explain_only) return;  // fdb_plan_explain: the descriptor is validated and drawn, nothing can be pushed
  hip_check(hipSetDevice(device_), "hipSetDevice");
  ctx_ = Context::acquire(device_);
  stream_ = ctx_->stream;
}

Plan::Plan(const Plan& proto, CloneTag)
    : projs_(proto.projs_), device_(proto.device_), filter_(proto.filter_), filter_root_(proto.filter_root_), aggs_(proto.aggs_),
      matchers_(proto.matchers_), final_stage_(proto.final_stage_), ordered_(proto.ordered_) {
  for (AggState& a : aggs_) a.d_acc = nullptr;  // (value types are kept: a clone merges with its prototype)
  hip_check(hipSetDevice(device_), "hipSetDevice");
  ctx_ = Context::acquire(device_);
  stream_ = ctx_->stream;
}

Plan::~Plan() {
  if (prof_push_n_ > 0 && std::getenv("FDB_PROFILE_PUSH") != nullptr)
    std::fprintf(stderr, "[fdb] push of %lld small records: view %.2f import %.2f resolve %.2f queue/settle %.2f us per record\n", (long long)prof_push_n_, prof_push_[0] / prof_push_n_,
                 prof_push_[1] / prof_push_n_, prof_push_[2] / prof_push_n_, prof_push_[3] / prof_push_n_);
  if (ctx_ == nullptr) return;
  (void)hipSetDevice(device_);
  (void)hipStreamSynchronize(stream_);
  for (auto& p : pending_events_) { ctx_->put_event(p.first); ctx_->put_event(p.second); }
  for (auto& p : merge_events_) { ctx_->put_event(p.first); ctx_->put_event(p.second); }
  for (auto& t : trace_) (void)hipEventDestroy(t.second);
  if (h_mirror_ != nullptr) ctx_->host_free(h_mirror_);
  ctx_->dev_free(d_state_);
  ctx_->dev_free(h_table_);
  ctx_->dev_free(h_keys_);
  ctx_->dev_free(h_count_dev_);
  for (RunSegment& r : runs_) ctx_->dev_free(r.block);
  for (RecordSlab& sl : inflight_slabs_) { ctx_->dev_free(sl.d); ctx_->host_free(sl.h); }
  if (slab_.d != nullptr) { ctx_->dev_free(slab_.d); ctx_->host_free(slab_.h); }
  for (void* p : scratch_) ctx_->dev_free(p);
  pending_.clear();   // (queued / in-flight records hand their arenas back to this context: before it is released)
  inflight_.clear();
  ctx_->reset_staging();
  Context::release(ctx_);
}

const Projection* Plan::find_projection(const std::string& name) const {
  for (const Projection& p : projs_) if (p.name == name) return &p;
  return nullptr;
}

// One expression becomes a run of FdbExprNode in the record's argument block (children before parents). Types follow
// binaryExprProjection.Project: the result has the left operand's type and the right operand must have the same one
// (project.go:104-160 type-switches on the left array and type-asserts the right one).
static void resolve_leaf(const ExprNode& e, const DeviceBatch& b, Plan::Resolved* R, FdbLeaf* L);

int Plan::resolve_projection(const Projection& p, const DeviceBatch& b, Resolved* R) {
  FdbScanArgs& a = R->args;
  const int base = a.n_expr;
  if (base + (int)p.nodes.size() > FDB_MAX_EXPR_NODES) throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + " is too large for the device path");
  for (size_t k = 0; k < p.nodes.size(); k++) {
    const ProjNode& n = p.nodes[k];
    FdbExprNode& e = a.expr[base + (int)k];
    std::memset(&e, 0, sizeof(e));
    e.kind = n.kind; e.op = n.op; e.left = e.right = -1; e.slot = -1;
    // A comparison whose sides are a column and a string / binary / NULL literal — or a dictionary / string column and any literal — is
    // what a FILTER leaf is: boolExprProjection.Project runs the expression through BooleanExpression.Eval (project.go:409-447), i.e.
    // BinaryScalarExpr with its per-type rules (binaryscalarexpr.go:41-152). It is resolved as one more leaf of the record's argument
    // block (truth table per dictionary entry; not part of the filter program) and the node reads that leaf's match bits (kind 7).
    auto leaf_compare = [&](const ProjNode& cmp) {
      if (cmp.kind != 3 || cmp.op == FDB_OP_AND || cmp.op == FDB_OP_OR) return false;
      const ProjNode& l = p.nodes[(size_t)cmp.left];
      const ProjNode& r = p.nodes[(size_t)cmp.right];
      if (l.kind != 0 || r.kind != 1) return false;
      if (r.lit_type == FDB_LIT_STRING || r.lit_type == FDB_LIT_BINARY || r.lit_type == FDB_LIT_NULL) return true;
      const int ci = b.find(l.column);
      return ci < 0 || b.cols[(size_t)ci].kind == ColKind::DICT || b.cols[(size_t)ci].kind == ColKind::U64;  // (uint64: compared unsigned, as a leaf is)
    };
    auto operand_of_leaf = [&](size_t idx) {  // node `idx` is only ever read by leaf comparisons
      bool any = false;
      for (const ProjNode& q : p.nodes)
        if (q.kind >= 2 && (q.left == (int)idx || q.right == (int)idx || (q.kind == 6 && q.op == (int)idx))) { if (!leaf_compare(q)) return false; any = true; }
      return any;
    };
    if (n.kind == 0 && operand_of_leaf(k)) {
      e.kind = 8; e.type = FDB_T_NONE;  // (an operand the leaf comparison consumes: the leaf reads the column itself; nothing is generated for the node)
    } else if (n.kind == 1 && operand_of_leaf(k)) {
      e.kind = 8; e.type = FDB_T_NONE;
    } else if (leaf_compare(n)) {
      if (a.n_leaves >= FDB_MAX_LEAVES) throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": too many predicate leaves");
      ExprNode fe;
      fe.op = n.op; fe.column = p.nodes[(size_t)n.left].column; fe.lit = p.nodes[(size_t)n.right].lit;
      const int li = a.n_leaves++;
      FdbLeaf& L = a.leaves[li];
      std::memset(&L, 0, sizeof(L));
      L.lut_lds = FDB_NO_LDS; L.slot = -1;
      R->luts.reserve(64);
      R->cur_node = -1 - (int)(base + k);  // (truth tables are cached per (node, dictionary): projection nodes get ids of their own)
      resolve_leaf(fe, b, R, &a.leaves[li]);
      e.kind = 7; e.slot = li; e.left = e.right = -1; e.type = FDB_T_BOOL;
    } else if (n.kind == 0) {
      const int ci = b.find(n.column);
      if (ci < 0) throw Error(FDB_ERR_NOT_FOUND, "projection " + p.name + ": column " + n.column + " not found");
      const DevColumn& c = b.cols[(size_t)ci];
      e.type = c.kind == ColKind::I64 ? FDB_T_I64 : c.kind == ColKind::F64 ? FDB_T_F64 : c.kind == ColKind::U64 ? FDB_T_U64 : FDB_T_NONE;  // (Int32 columns do not exist on this path)
      if (e.type == FDB_T_NONE) throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": unsupported type of column " + n.column);  // project.go:157-159
      if (c.d_values == nullptr) throw Error(FDB_ERR_INVALID, "column not staged: " + c.name);
      R->count(b, ci);
      R->expr_col[base + (int)k] = ci;
    } else if (n.kind == 1) {
      if (n.lit_type != FDB_LIT_INT64 && n.lit_type != FDB_LIT_FLOAT64 && n.lit_type != FDB_LIT_UINT64) throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": a string / NULL literal can only be compared with a column");
      e.type = n.lit_type == FDB_LIT_INT64 ? FDB_T_I64 : n.lit_type == FDB_LIT_UINT64 ? FDB_T_U64 : FDB_T_F64;
      if (e.type != FDB_T_F64) e.lit = n.i64; else std::memcpy(&e.lit, &n.f64, 8);
    } else if (n.kind == 4) {  // convertProjection.convert (project.go:507-521): only int64 → float64 exists
      e.left = base + n.left;
      if (a.expr[e.left].type != FDB_T_I64) throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": unsupported conversion (only int64 to float64)");
      e.type = FDB_T_F64;
    } else if (n.kind == 5) {  // isNullProjection (project.go:571-601): the validity of a COLUMN
      e.left = base + n.left;
      if (a.expr[e.left].kind != 0) throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": isnull takes a column");
      e.type = FDB_T_BOOL;
    } else if (n.kind == 6) {  // ifExprProjection (project.go:619-683): boolean condition, int64 branches
      e.left = base + n.left; e.right = base + n.right; e.op = base + n.op;
      if (a.expr[e.op].type != FDB_T_BOOL) throw Error(FDB_ERR_INVALID, "projection " + p.name + ": invalid projection for if: condition column must be of type boolean");
      if (a.expr[e.left].type != a.expr[e.right].type) throw Error(FDB_ERR_INVALID, "projection " + p.name + ": invalid projection for if: then and else columns must be of the same type");
      if (a.expr[e.left].type != FDB_T_I64) throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": unsupported if expression type (int64 only)");
      e.type = FDB_T_I64;
    } else if (n.kind == 3) {  // comparison → bool; int64 / float64 operands may mix like in a filter leaf (compared as doubles)
      e.left = base + n.left; e.right = base + n.right;
      const bool logical = n.op == FDB_OP_AND || n.op == FDB_OP_OR;
      if (logical && (a.expr[e.left].type != FDB_T_BOOL || a.expr[e.right].type != FDB_T_BOOL))
        throw Error(FDB_ERR_INVALID, "projection " + p.name + ": AND / OR need boolean operands");
      if (!logical && (a.expr[e.left].type == FDB_T_BOOL || a.expr[e.right].type == FDB_T_BOOL))
        throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": comparison of boolean values");
      if (!logical && (a.expr[e.left].type == FDB_T_U64 || a.expr[e.right].type == FDB_T_U64))
        throw Error(FDB_ERR_UNSUPPORTED, "projection " + p.name + ": comparison of computed uint64 values");
      e.type = FDB_T_BOOL;
    } else {
      e.left = base + n.left; e.right = base + n.right;
      if (a.expr[e.left].type != a.expr[e.right].type || a.expr[e.left].type == FDB_T_BOOL)
        throw Error(FDB_ERR_INVALID, "projection " + p.name + ": operand types differ (the reference type-asserts the right array to the left one's type, project.go:112-160)");
      e.type = a.expr[e.left].type;
    }
  }
  a.n_expr = base + (int)p.nodes.size();
  return base + p.root;
}

bool Plan::references(const std::string& column) const {
  for (const Projection& p : projs_)
    for (const ProjNode& n : p.nodes) if (n.kind == 0 && n.column == column) return true;
  for (const ExprNode& e : filter_) if (is_leaf_op(e.op) && e.column == column) return true;
  for (const AggState& a : aggs_) if ((final_stage_ ? a.result_name : a.column) == column) return true;
  for (const GroupMatcher& m : matchers_) if (match_group(m, column)) return true;
  return false;
}

void Plan::sync() {
  hip_check(hipSetDevice(device_), "hipSetDevice");
  // (polling the stream with hipStreamQuery before the blocking wait was tried in round 5 to cut the ≈ 10 µs wake-up: 53 µs outside the
  // kernel of a 125 M-row step against 37 µs with the blocking wait alone — profiles/round5_step_probe_polling.txt)
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  collect_timing();
  ctx_->reset_staging();
  for (void* p : scratch_) ctx_->dev_free(p);
  scratch_.clear();
  inflight_.clear();  // (their arenas go back to the block cache)
  for (RecordSlab& sl : inflight_slabs_) { ctx_->dev_free(sl.d); ctx_->host_free(sl.h); }
  inflight_slabs_.clear();
  if (slab_.d != nullptr && pending_.empty() && slab_.shipped == slab_.used) slab_.used = slab_.shipped = 0;  // nothing refers to the open slab any more
}

void Plan::state_idents(unsigned long long* idents) const {
  idents[0] = 0ull;
  for (size_t j = 0; j < aggs_.size(); j++) {
    const int32_t f = aggs_[j].func;
    idents[1 + j] = f == FDB_AGG_MIN ? (unsigned long long)FDB_I64_MAX : f == FDB_AGG_MAX ? (unsigned long long)FDB_I64_MIN : 0ull;
  }
}

unsigned long long* Plan::mirror_target() {
  constexpr size_t kMaxMirror = (size_t)256 << 10;
  static const bool off = std::getenv("FDB_NO_HOST_MIRROR") != nullptr;  // (A/B aid)
  const size_t bytes = (size_t)slots_alloc_ * 8 * (1 + aggs_.size());
  if (off || d_state_ == nullptr || bytes == 0 || bytes > kMaxMirror) return nullptr;
  if (h_mirror_ != nullptr && mirror_bytes_ != bytes) {
    // a fold kernel of an earlier push may still be writing the old mirror: it goes back to the process-wide pinned pool only
    // once the stream is idle (rare: the table was re-laid-out between two pushes)
    (void)hipStreamSynchronize(stream_);
    ctx_->host_free(h_mirror_); h_mirror_ = nullptr; mirror_valid_ = false;
  }
  if (h_mirror_ == nullptr) { h_mirror_ = (unsigned long long*)ctx_->host_alloc(bytes); mirror_bytes_ = bytes; }
  return h_mirror_;
}

void Plan::materialize_state() {
  if (!state_virgin_ || d_state_ == nullptr) { state_virgin_ = false; return; }
  unsigned long long idents[1 + FDB_MAX_AGGS];
  state_idents(idents);
  hip_check(hipSetDevice(device_), "hipSetDevice");
  hip_check(fdb_launch_fill_state(d_state_, (int64_t)slots_alloc_, (int)(1 + aggs_.size()), idents, stream_), "fill state");
  state_virgin_ = false;
}

void Plan::trace(const char* what) {
  if (std::getenv("FDB_PROFILE") == nullptr) return;
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return;
  if (hipEventRecord(e, stream_) != hipSuccess) { (void)hipEventDestroy(e); return; }
  trace_.emplace_back(what, e);
}

void Plan::collect_timing() {
  if (!trace_.empty()) {
    for (size_t i = 1; i < trace_.size(); i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, trace_[i - 1].second, trace_[i].second) == hipSuccess)
        std::fprintf(stderr, "[fdb] stream: %-24s → %-24s %8.1f us\n", trace_[i - 1].first, trace_[i].first, ms * 1e3);
    }
    for (auto& t : trace_) (void)hipEventDestroy(t.second);
    trace_.clear();
  }
  for (auto& p : pending_events_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) stat_ms += ms;
    ctx_->put_event(p.first);
    ctx_->put_event(p.second);
  }
  pending_events_.clear();
  for (auto& p : merge_events_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) stat_merge_ms += ms;
    ctx_->put_event(p.first);
    ctx_->put_event(p.second);
  }
  merge_events_.clear();
}

void* Plan::upload(const void* host, size_t bytes) { return ctx_->stage(host, bytes); }

const char* Plan::draw() {
  if (draw_.empty()) {
    std::function<std::string(int)> show = [&](int i) -> std::string {
      const ExprNode& e = filter_[(size_t)i];
      if (e.op == FDB_OP_AND) return "(" + show(e.left) + " AND " + show(e.right) + ")";  // filter.go:192-194
      if (e.op == FDB_OP_OR) return "(" + show(e.left) + " OR " + show(e.right) + ")";
      if (e.op == FDB_OP_REGEX_MATCH) return e.column + " =~ \"" + e.lit.bytes + "\"";     // regexpfilter.go:41-46
      if (e.op == FDB_OP_REGEX_NOT_MATCH) return e.column + " !~ \"" + e.lit.bytes + "\"";
      return e.column + " " + op_str(e.op) + " " + e.lit.str();                            // binaryscalarexpr.go:78-80
    };
    std::string s;
    if (filter_root_ >= 0) s += "PredicateFilter (" + show(filter_root_) + ")";
    if (!aggs_.empty() && ordered_) {  // "OrderedAggregate (%s by %s)" with the aggregated COLUMN's name (ordered_aggregate.go:154-158)
      if (!s.empty()) s += " - ";
      s += "OrderedAggregate (" + aggs_[0].column + " by ";
      for (size_t i = 0; i < matchers_.size(); i++) s += (i ? "," : "") + matchers_[i].name;
      s += ")";
    } else if (!aggs_.empty()) {
      if (!s.empty()) s += " - ";
      s += "HashAggregate (";
      for (size_t i = 0, shown = 0; i < aggs_.size(); i++) if (aggs_[i].role != 2) s += (shown++ ? "," : "") + aggs_[i].result_name;
      s += " by ";
      for (size_t i = 0; i < matchers_.size(); i++) s += (i ? "," : "") + matchers_[i].name;
      s += ")";
    } else if (!matchers_.empty()) {  // distinct.go:32-45
      if (!s.empty()) s += " - ";
      s += "Distinction (";
      for (size_t i = 0; i < matchers_.size(); i++) s += (i ? "," : "") + matchers_[i].name;
      s += ")";
    }
    draw_ = s + " [gfx950]";
  }
  return draw_.c_str();
}

// ---- filter resolution ---------------------------------------------------------------------------------
namespace {

// Truth table of a predicate over ONE dictionary column: an answer per dictionary entry plus the answer for a NULL
// row (last element). This is what replaces the reference's per-row string compare (binaryscalarexpr.go:154-311).
typedef std::vector<uint8_t> Truth;

Truth leaf_truth(const ExprNode& e, const HostDict& dict) {
  const size_t n = dict.values.size();
  Truth t(n + 1, 0);
  if (!dict.plain && !e.lit.valid() && (e.op == FDB_OP_EQ || e.op == FDB_OP_NOT_EQ || e.op == FDB_OP_CONTAINS || e.op == FDB_OP_NOT_CONTAINS)) {
    // == NULL ⇒ IS NULL, != NULL ⇒ IS NOT NULL (:165-172, :205-212); (not) contains NULL ⇒ every non-null row (:287-295)
    const bool on_value = e.op != FDB_OP_EQ;
    for (size_t i = 0; i < n; i++) t[i] = on_value;
    t[n] = e.op == FDB_OP_EQ;
    return t;
  }
  for (size_t i = 0; i < n; i++) {
    const std::string& v = dict.values[i];
    bool m = false;
    switch (e.op) {
      case FDB_OP_EQ: m = (v == e.lit.bytes); break;          // non string/binary literals compare against "" (binaryscalarexpr.go:196-202)
      case FDB_OP_NOT_EQ: m = (v != e.lit.bytes); break;
      case FDB_OP_CONTAINS: m = v.find(e.lit.bytes) != std::string::npos; break;
      case FDB_OP_NOT_CONTAINS: m = v.find(e.lit.bytes) == std::string::npos; break;
      // order comparisons exist for plain columns only (resolve_leaf rejects them on dictionaries): bytewise, like Arrow's
      case FDB_OP_LT: m = v.compare(e.lit.bytes) < 0; break;
      case FDB_OP_LT_EQ: m = v.compare(e.lit.bytes) <= 0; break;
      case FDB_OP_GT: m = v.compare(e.lit.bytes) > 0; break;
      case FDB_OP_GT_EQ: m = v.compare(e.lit.bytes) >= 0; break;
      case FDB_OP_REGEX_MATCH: m = e.regex_matches(v); break;
      case FDB_OP_REGEX_NOT_MATCH: m = !e.regex_matches(v); break;
    }
    t[i] = m ? 1 : 0;
  }
  t[n] = 0;  // NULL rows never satisfy a value predicate (:175-177, :215-217)
  return t;
}

bool dict_leaf_op(int32_t op) {
  return op == FDB_OP_EQ || op == FDB_OP_NOT_EQ || op == FDB_OP_CONTAINS || op == FDB_OP_NOT_CONTAINS || op == FDB_OP_REGEX_MATCH ||
         op == FDB_OP_REGEX_NOT_MATCH;
}

// True if every leaf under `idx` tests the same dictionary column `*col` of `b` with an operator the dictionary
// path supports; such a subtree collapses into ONE truth table (e.g. code=='200' OR code=='500').
bool single_dict_subtree(const std::vector<ExprNode>& nodes, int idx, const DeviceBatch& b, int* col, int* n_leaves) {
  const ExprNode& e = nodes[(size_t)idx];
  if (e.op == FDB_OP_AND || e.op == FDB_OP_OR)
    return single_dict_subtree(nodes, e.left, b, col, n_leaves) && single_dict_subtree(nodes, e.right, b, col, n_leaves);
  if (!dict_leaf_op(e.op)) return false;
  const int ci = b.find(e.column);
  if (ci < 0 || b.cols[(size_t)ci].kind != ColKind::DICT || b.cols[(size_t)ci].d_values == nullptr) return false;
  const bool is_regex = e.op == FDB_OP_REGEX_MATCH || e.op == FDB_OP_REGEX_NOT_MATCH;
  const HostDict& d = *b.cols[(size_t)ci].dict;
  if (is_regex && !d.plain && d.utf8()) return false;  // must raise, see resolve_leaf
  // plain columns: = / != with a NULL or non-string literal have their own rules (resolve_leaf), not a truth table
  if (d.plain && (e.op == FDB_OP_EQ || e.op == FDB_OP_NOT_EQ) && e.lit.type != FDB_LIT_STRING && e.lit.type != FDB_LIT_BINARY) return false;
  if (*col >= 0 && *col != ci) return false;
  *col = ci;
  (*n_leaves)++;
  return true;
}

Truth subtree_truth(const std::vector<ExprNode>& nodes, int idx, const HostDict& dict) {
  const ExprNode& e = nodes[(size_t)idx];
  if (e.op == FDB_OP_AND || e.op == FDB_OP_OR) {
    Truth l = subtree_truth(nodes, e.left, dict);
    const Truth r = subtree_truth(nodes, e.right, dict);
    for (size_t i = 0; i < l.size(); i++) l[i] = e.op == FDB_OP_AND ? (l[i] & r[i]) : (l[i] | r[i]);
    return l;
  }
  return leaf_truth(e, dict);
}

}  // namespace

static void resolve_leaf(const ExprNode& e, const DeviceBatch& b, Plan::Resolved* R, FdbLeaf* L);

// The truth table of filter node `node` over `dict`: computed by `make` once per (node, dictionary object).
template <typename F>
static std::shared_ptr<const Truth> cached_truth(Plan::Resolved* R, int node, const std::shared_ptr<HostDict>& dict, F make) {
  if (R->truths != nullptr && node >= 0) {
    std::shared_ptr<const Truth> t = R->truths->find(node, dict.get());
    if (t) return t;
  }
  std::shared_ptr<const Truth> t = std::make_shared<const Truth>(make());
  if (R->truths != nullptr && node >= 0) R->truths->put(node, dict, t);
  return t;
}

// Turns a truth table over dictionary column `ci` into a leaf: ≤ 64 answers ride in a 64-bit immediate (no LDS,
// no memory access at all), larger tables become a byte LUT staged in LDS.
static void emit_truth_leaf(const Truth& t, const DeviceBatch& b, int ci, Plan::Resolved* R, FdbLeaf* L) {
  const int li = (int)(L - R->args.leaves);
  const DevColumn& c = b.cols[(size_t)ci];
  L->values = c.d_values;
  L->validity = c.d_validity;
  L->wide = 0;
  L->lut_len = (uint32_t)t.size();
  R->leaf_col[li] = ci;
  R->count(b, ci);
  if (t.size() <= 64) {
    unsigned long long bits = 0;
    for (size_t i = 0; i < t.size(); i++) if (t[i]) bits |= 1ull << i;
    L->kind = FDB_LEAF_DICT_BITS;
    L->lit = (int64_t)bits;
    return;
  }
  L->kind = FDB_LEAF_DICT_LUT;
  const size_t off = R->blob.add(t.data(), t.size());
  R->luts.push_back(PendingLut{0, li, off, t.size()});
}

static void emit_filter(const std::vector<ExprNode>& nodes, int idx, const DeviceBatch& b, Plan::Resolved* R, int depth, int* max_depth) {
  const ExprNode& e = nodes[(size_t)idx];
  FdbScanArgs& a = R->args;
  if (e.op == FDB_OP_AND || e.op == FDB_OP_OR) {
    int col = -1, n_sub = 0;
    if (single_dict_subtree(nodes, idx, b, &col, &n_sub) && n_sub >= 2) {
      if (a.n_leaves >= FDB_MAX_LEAVES || a.n_code >= FDB_MAX_CODE) throw Error(FDB_ERR_UNSUPPORTED, "filter expression too large");
      if (depth + 1 > *max_depth) *max_depth = depth + 1;
      if (*max_depth > 8) throw Error(FDB_ERR_UNSUPPORTED, "filter expression too deep");
      const int li = a.n_leaves++;
      FdbLeaf& L = a.leaves[li];
      std::memset(&L, 0, sizeof(L));
      L.lut_lds = FDB_NO_LDS;
      L.slot = -1;
      const std::shared_ptr<HostDict>& dict = b.cols[(size_t)col].dict;
      emit_truth_leaf(*cached_truth(R, idx, dict, [&] { return subtree_truth(nodes, idx, *dict); }), b, col, R, &L);
      a.code[a.n_code++] = (uint8_t)li;
      return;
    }
    const int leaves0 = a.n_leaves, code0 = a.n_code;
    const size_t luts0 = R->luts.size();
    emit_filter(nodes, e.left, b, R, depth, max_depth);
    try {
      emit_filter(nodes, e.right, b, R, depth + 1, max_depth);
    } catch (const Error& err) {
      // The reference would only notice a right side it cannot evaluate if the left side selected a row of this record
      // (AndExpr.Eval returns the empty left bitmap without touching the right side, filter.go:178-180; filter_test.go:66-82).
      if (e.op != FDB_OP_AND || err.code != FDB_ERR_UNSUPPORTED || !R->count_selected || b.rows == 0 || R->count_selected(e.left) != 0) throw;
      // nothing selected on the left: for this record the whole AND is the empty selection
      a.n_leaves = leaves0; a.n_code = code0;
      R->luts.resize(luts0);
      if (a.n_leaves >= FDB_MAX_LEAVES || a.n_code >= FDB_MAX_CODE) throw Error(FDB_ERR_UNSUPPORTED, "filter expression too large");
      FdbLeaf& L = a.leaves[a.n_leaves];
      std::memset(&L, 0, sizeof(L));
      L.lut_lds = FDB_NO_LDS; L.slot = -1; L.kind = FDB_LEAF_CONST; L.op = 0;
      a.code[a.n_code++] = (uint8_t)a.n_leaves++;
      return;
    }
    if (a.n_code >= FDB_MAX_CODE) throw Error(FDB_ERR_UNSUPPORTED, "filter expression too large");
    a.code[a.n_code++] = e.op == FDB_OP_AND ? FDB_CODE_AND : FDB_CODE_OR;
    return;
  }
  if (a.n_leaves >= FDB_MAX_LEAVES || a.n_code >= FDB_MAX_CODE) throw Error(FDB_ERR_UNSUPPORTED, "filter expression too large");
  if (depth + 1 > *max_depth) *max_depth = depth + 1;
  if (*max_depth > 8) throw Error(FDB_ERR_UNSUPPORTED, "filter expression too deep");
  FdbLeaf L;
  std::memset(&L, 0, sizeof(L));
  L.lut_lds = FDB_NO_LDS;
  L.slot = -1;
  R->luts.reserve(64);
  const int li = a.n_leaves;
  a.leaves[li] = L;
  // resolve (may append a LUT that refers to leaf index li)
  a.n_leaves++;
  R->cur_node = idx;
  resolve_leaf(e, b, R, &a.leaves[li]);
  a.code[a.n_code++] = (uint8_t)li;
}

static void resolve_leaf(const ExprNode& e, const DeviceBatch& b, Plan::Resolved* R, FdbLeaf* L) {
  const int li = (int)(L - R->args.leaves);
  const bool is_regex = e.op == FDB_OP_REGEX_MATCH || e.op == FDB_OP_REGEX_NOT_MATCH;
  const int ci = b.find(e.column);
  auto set_const = [&](bool v) { L->kind = FDB_LEAF_CONST; L->op = v ? 1 : 0; };
  if (ci < 0) {
    if (is_regex) {  // regexpfilter.go:23-33
      const bool empty_match = e.regex_matches(std::string());
      const bool neg = e.op == FDB_OP_REGEX_NOT_MATCH;
      set_const((neg && !empty_match) || (!neg && empty_match));
      return;
    }
    switch (e.op) {  // binaryscalarexpr.go:47-73
      case FDB_OP_EQ:
        if (e.lit.valid() && (e.lit.type == FDB_LIT_STRING || e.lit.type == FDB_LIT_BINARY) && !e.lit.bytes.empty()) { set_const(false); return; }
        break;
      case FDB_OP_NOT_EQ:
        if (!e.lit.valid()) { set_const(false); return; }
        break;
      case FDB_OP_LT: case FDB_OP_LT_EQ: case FDB_OP_GT: case FDB_OP_GT_EQ:
        set_const(false);
        return;
    }
    set_const(true);
    return;
  }
  const DevColumn& c = b.cols[(size_t)ci];
  const bool is_contains = e.op == FDB_OP_CONTAINS || e.op == FDB_OP_NOT_CONTAINS;
  if (c.kind == ColKind::DICT && c.dict->plain) {
    // A plain string / binary column (encoded on import). Not the dictionary rules: regex and contains take both String and
    // Binary arrays (regexpfilter.go:48-54, binaryscalarexpr.go:87-89, :234-270 — a NULL literal searches for ""), and
    // = != < <= > >= go to Arrow's compare kernels (binaryscalarexpr.go:116, :119-152): bytewise order, a NULL scalar
    // yields NULL for every row ⇒ no row, a non-string scalar has no kernel.
    if (c.d_values == nullptr) throw Error(FDB_ERR_INVALID, "column not staged: " + c.name);
    if (!is_regex && !is_contains) {
      if (!(e.op >= FDB_OP_EQ && e.op <= FDB_OP_GT_EQ)) throw Error(FDB_ERR_UNSUPPORTED, "unsupported binary operation");
      if (!e.lit.valid()) { set_const(false); return; }
      if (e.lit.type != FDB_LIT_STRING && e.lit.type != FDB_LIT_BINARY)
        throw Error(FDB_ERR_UNSUPPORTED, "unsupported binary operation (column/literal type combination) on " + c.name);
    }
    emit_truth_leaf(*cached_truth(R, R->cur_node, c.dict, [&] { return leaf_truth(e, *c.dict); }), b, ci, R, L);
    return;
  }
  if (c.kind == ColKind::DICT) {
    if (c.d_values == nullptr) throw Error(FDB_ERR_INVALID, "column not staged: " + c.name);
    if (is_regex && c.dict->utf8())  // regexpfilter.go:55-61: only *array.Binary dictionaries
      throw Error(FDB_ERR_UNSUPPORTED, "ArrayScalarRegexMatch: unsupported dictionary type: *array.String");
    if (!is_regex && !is_contains && e.op != FDB_OP_EQ && e.op != FDB_OP_NOT_EQ)
      throw Error(FDB_ERR_UNSUPPORTED, std::string("unsupported operator: ") + op_str(e.op));  // binaryscalarexpr.go:106-108
    L->values = c.d_values;
    L->validity = c.d_validity;
    L->wide = 0;
    R->leaf_col[li] = ci;
    if (!is_regex && !e.lit.valid()) {
      // == NULL ⇒ IS NULL, != NULL ⇒ IS NOT NULL (:165-172, :205-212); contains/not-contains NULL ⇒ every non-null row (:287-295)
      if (c.d_validity == nullptr) { R->leaf_col[li] = -1; set_const(e.op != FDB_OP_EQ); return; }  // no NULLs in this record
      L->kind = FDB_LEAF_VALIDITY;
      L->op = (e.op == FDB_OP_EQ) ? 0 : 1;
      R->count(b, ci, /*values=*/false);  // the index buffer is not read by this leaf: only the bitmap counts
      return;
    }
    emit_truth_leaf(*cached_truth(R, R->cur_node, c.dict, [&] { return leaf_truth(e, *c.dict); }), b, ci, R, L);
    return;
  }
  if (is_regex) throw Error(FDB_ERR_UNSUPPORTED, "ArrayScalarRegexMatch: unsupported type on the device path: " + c.format);
  if (is_contains) throw Error(FDB_ERR_UNSUPPORTED, "contains on a non-dictionary column is not supported on the device path: " + c.name);
  if (!(e.op >= FDB_OP_EQ && e.op <= FDB_OP_GT_EQ)) throw Error(FDB_ERR_UNSUPPORTED, "unsupported binary operation");
  if (c.kind != ColKind::I64 && c.kind != ColKind::U64 && c.kind != ColKind::F64 && c.kind != ColKind::BOOL)
    throw Error(FDB_ERR_UNSUPPORTED, "unsupported binary operation: compare on column type " + c.format + " (" + c.name + ")");
  if (c.d_values == nullptr) throw Error(FDB_ERR_INVALID, "column not staged: " + c.name);
  if (!e.lit.valid()) { set_const(false); return; }  // compare with a NULL scalar yields NULL for every row (binaryscalarexpr.go:143-146)
  R->count(b, ci);
  L->values = c.d_values;
  L->validity = c.d_validity;
  L->op = e.op;
  L->wide = 1;
  R->leaf_col[li] = ci;
  auto dbits = [](double d) { int64_t v; std::memcpy(&v, &d, 8); return v; };
  if (c.kind == ColKind::BOOL) {  // staged as int64 1 (false) / 2 (true); Arrow orders false < true
    if (e.lit.type == FDB_LIT_BOOL) { L->kind = FDB_LEAF_CMP_I64; L->lit = e.lit.i64 ? 2 : 1; return; }
  } else if (c.kind == ColKind::I64) {
    if (e.lit.type == FDB_LIT_INT64) { L->kind = FDB_LEAF_CMP_I64; L->lit = e.lit.i64; return; }
    if (e.lit.type == FDB_LIT_FLOAT64) { L->kind = FDB_LEAF_CMP_I64_F64; L->lit = dbits(e.lit.f64); return; }
  } else if (c.kind == ColKind::U64) {
    if (e.lit.type == FDB_LIT_UINT64) { L->kind = FDB_LEAF_CMP_U64; L->lit = (int64_t)e.lit.u64; return; }
    if (e.lit.type == FDB_LIT_INT64 && e.lit.i64 >= 0) { L->kind = FDB_LEAF_CMP_U64; L->lit = e.lit.i64; return; }
  } else {
    if (e.lit.type == FDB_LIT_FLOAT64) { L->kind = FDB_LEAF_CMP_F64; L->lit = dbits(e.lit.f64); return; }
    if (e.lit.type == FDB_LIT_INT64) { L->kind = FDB_LEAF_CMP_F64; L->lit = dbits((double)e.lit.i64); return; }
  }
  throw Error(FDB_ERR_UNSUPPORTED, "unsupported binary operation (column/literal type combination) on " + c.name);
}

// ---- group-key dictionaries ---------------------------------------------------------------------------------
void GroupColState::build_ids() {
  if (ids_built_ == values.size()) return;
  ids_.reserve(values.size() * 2);
  for (; ids_built_ < values.size(); ids_built_++) ids_.emplace(values[ids_built_], (uint32_t)ids_built_ + 1);
}

uint32_t GroupColState::intern(std::string_view v) {
  build_ids();
  auto it = ids_.find(v);
  if (it != ids_.end()) return it->second;
  values.push_back(v);
  ids_.emplace(v, (uint32_t)values.size());
  ids_built_ = values.size();
  return (uint32_t)values.size();
}

std::shared_ptr<const std::vector<uint32_t>> GroupColState::lut_for(const std::shared_ptr<HostDict>& d) {
  auto range = lut_cache_.equal_range(d->hash);
  for (auto it = range.first; it != range.second; ++it)
    if (it->second.dict.get() == d.get() || it->second.dict->same_content(*d)) return it->second.lut;
  auto lut = std::make_shared<std::vector<uint32_t>>(std::max<size_t>(d->values.size(), 1), 0u);
  owners.push_back(d);
  if (values.empty() && d->unique) {
    // first dictionary of this column: ids are entry + 1, no hashing at all (the hash map is built only if a
    // different dictionary ever shows up)
    values.reserve(d->values.size());
    for (size_t e = 0; e < d->values.size(); e++) { values.emplace_back(d->values[e]); (*lut)[e] = (uint32_t)e + 1; }
    adopted = d;
  } else {
    for (size_t e = 0; e < d->values.size(); e++) (*lut)[e] = intern(std::string_view(d->values[e]));
  }
  lut_cache_.emplace(d->hash, Cached{d, lut});
  return lut;
}

// ---- table layout ----------------------------------------------------------------------------------------
void Plan::ensure_layout(const std::vector<uint32_t>& new_caps) {
  // new_caps has one entry per gcols_ element (columns may have been appended since the last layout)
  uint64_t n_new = 1;
  std::vector<uint32_t> new_strides(new_caps.size());
  for (size_t c = 0; c < new_caps.size(); c++) {
    new_strides[c] = (uint32_t)n_new;
    n_new *= new_caps[c];
    if (n_new > (1ull << 26)) throw Error(FDB_ERR_UNSUPPORTED, "group-key space exceeds the dense table limit (2^26 slots); high-cardinality hash path not built yet");
  }
  // Slots keep their index unless a column that already holds non-NULL ids changes its multiplier.
  bool remap = false;
  for (size_t c = 0; c < gcols_.size(); c++)
    if (gcols_[c].cap > 1 && gcols_[c].stride != new_strides[c]) remap = true;
  const bool need_alloc = n_new > slots_alloc_ || d_state_ == nullptr || (remap && state_dirty_);
  if (need_alloc) {
    // One block holds the whole table: [cnt | acc 0 | acc 1 | …], each `alloc` slots long, identity-filled.
    const uint64_t cap = std::max<uint64_t>(n_new, 64);
    const uint64_t alloc = (remap && state_dirty_) ? cap : std::max<uint64_t>(cap, slots_alloc_ * 2);
    const size_t n_arrays = 1 + aggs_.size();
    unsigned long long* n_state = (unsigned long long*)ctx_->dev_alloc(alloc * 8 * n_arrays);
    mirror_valid_ = false;
    const bool carry = state_dirty_ && d_state_ != nullptr;  // the old table holds groups: they move into the new one, which must be filled first
    if (carry) {
      unsigned long long idents[1 + FDB_MAX_AGGS];
      state_idents(idents);
      hip_check(fdb_launch_fill_state(n_state, (int64_t)alloc, (int)n_arrays, idents, stream_), "fill state");
    }
    state_virgin_ = !carry;  // (otherwise filled lazily: by the first scan launch itself, or by materialize_state())
    if (carry) {
      const uint32_t* d_map = nullptr;
      std::vector<uint32_t> map;
      if (remap) {
        map.resize(n_slots_);
        for (uint32_t s = 0; s < n_slots_; s++) {
          uint64_t t = 0;
          for (size_t c = 0; c < gcols_.size(); c++) {
            if (gcols_[c].cap <= 1) continue;
            const uint32_t digit = (s / gcols_[c].stride) % gcols_[c].cap;
            t += (uint64_t)digit * new_strides[c];
          }
          map[s] = (uint32_t)t;
        }
        d_map = (const uint32_t*)upload(map.data(), map.size() * 4);
      }
      hip_check(fdb_launch_merge_u64(n_state, d_cnt_, d_map, n_slots_, FDB_AGG_SUM, 0, stream_), "remap cnt");
      for (size_t j = 0; j < aggs_.size(); j++) {
        const int32_t f = (aggs_[j].func == FDB_AGG_COUNT) ? FDB_AGG_SUM : aggs_[j].func;
        hip_check(fdb_launch_merge_u64(n_state + (1 + j) * alloc, aggs_[j].d_acc, d_map, n_slots_, f,
                                       aggs_[j].type == FDB_T_F64 && f == FDB_AGG_SUM, stream_), "remap acc");
      }
      hip_check(hipStreamSynchronize(stream_), "sync(remap)");
    }
    ctx_->dev_free(d_state_);  // same stream ⇒ any later reuse is ordered after the kernels above
    d_state_ = n_state;
    d_cnt_ = n_state;
    for (size_t j = 0; j < aggs_.size(); j++) aggs_[j].d_acc = n_state + (1 + j) * alloc;
    slots_alloc_ = alloc;
  }
  for (size_t c = 0; c < gcols_.size(); c++) { gcols_[c].cap = new_caps[c]; gcols_[c].stride = new_strides[c]; }
  n_slots_ = (uint32_t)n_new;
}

// ---- push -------------------------------------------------------------------------------------------------
namespace {
constexpr size_t kCoalesceMaxBytes = (size_t)8 << 20;   // records above this are scanned right away (the copy dominates anyway)
constexpr int64_t kFlushRows = 2 << 20;                 // pending rows that justify a launch
constexpr size_t kFlushBytes = (size_t)96 << 20;
constexpr size_t kFlushRecords = 1024;
}  // namespace

bool Plan::jit_possible() const { return sub_tiles != 4 && ablate == 0 && !knobs_.no_jit; }

Plan::Knobs::Knobs() {
  no_jit = std::getenv("FDB_NO_JIT") != nullptr;
  runs_always = std::getenv("FDB_RUNS_ALWAYS") != nullptr;
  no_identity_lut = std::getenv("FDB_NO_IDENTITY_LUT") != nullptr;
  runs_no_sort = std::getenv("FDB_RUNS_NO_SORT") != nullptr;
  no_uniform_fold = std::getenv("FDB_NO_UNIFORM_FOLD") != nullptr;
  no_present_ids = std::getenv("FDB_NO_PRESENT_IDS") != nullptr;  // (A/B aid: Finish ships dictionary indices at the dictionary's width)
  const char* w = std::getenv("FDB_RUNS_WIDE");
  runs_wide = w != nullptr ? (w[0] == '1' ? '1' : 'm') : 0;
  const char* e = std::getenv("FDB_ORDERED_SORT_MIN");
  ordered_sort_min = e != nullptr ? std::max<long long>(0, std::atoll(e)) : 4096;
  const char* pm = std::getenv("FDB_PRESENT_IDS_MIN_BYTES");
  present_ids_min_bytes = pm != nullptr ? std::max<long long>(0, std::atoll(pm)) : (long long)32 << 20;
  const char* mk = std::getenv("FDB_TEST_MAX_KEY_BYTES");
  max_key_bytes = mk != nullptr ? std::max<long long>(1, std::atoll(mk)) : 0x7FFFFFFFll;
  const char* fs = std::getenv("FDB_FINISH_SLICE_SHIFT");
  finish_slice_shift = fs != nullptr ? std::max(6, std::min(20, std::atoi(fs))) : 20;
}

void Plan::push(const ArrowArray* array, const ArrowSchema* schema) {
  if (finished_) throw Error(FDB_ERR_STATE, "push after finish");
  if (aggs_.empty() && matchers_.empty()) throw Error(FDB_ERR_STATE, "filter-only plan: use fdb_plan_filter / fdb_plan_select");
  // ($FDB_PROFILE_PUSH: tuning aid — where a small record's Callback goes, summed per plan and printed by the destructor)
  static const bool prof = std::getenv("FDB_PROFILE_PUSH") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto lap = [&](std::chrono::steady_clock::time_point& t, double& acc) { if (prof) { const auto n = now(); acc += std::chrono::duration<double, std::micro>(n - t).count(); t = n; } };
  std::chrono::steady_clock::time_point tp = now();
  HostRecordView view;
  view_record(array, schema, &view);
  std::function<bool(const std::string&)> want = [this](const std::string& n) { return references(n); };
  hip_check(hipSetDevice(device_), "hipSetDevice");
  lap(tp, prof_push_[0]);
  size_t payload = 0;
  for (const HostColView& c : view.cols)
    if (want(c.name)) payload += (size_t)c.length * (c.kind == ColKind::DICT ? 4 : 8) + (size_t)(c.length + 7) / 8;
  if (payload > kCoalesceMaxBytes) {
    settle();  // keep arrival order (it fixes the first-seen order of group columns and key ids)
    std::unique_ptr<DeviceBatch> b = import_batch(view, device_, &want, stream_, ctx_);
    push_batch(*b);
    sync();  // the caller's buffers are only borrowed (table.go:808): every copy has landed; the arena returns to the block cache
    return;
  }
  // small record: copy now (through pinned staging — the source is not touched after this call), scan later
  const RecordSink sink = [this](size_t bytes, void** dev, unsigned char** pinned) { slab_reserve(bytes, dev, pinned); };
  std::unique_ptr<DeviceBatch> b = import_batch(view, device_, &want, stream_, ctx_, /*via_ring=*/true, &sink);
  static const size_t ship_bytes = std::getenv("FDB_SLAB_SHIP_MB") ? (size_t)std::atoll(std::getenv("FDB_SLAB_SHIP_MB")) << 20 : (size_t)2 << 20;  // (tuning aid; 2 … 32 MiB measured within noise of each other, 2 MiB keeps the device busy earliest)
  if (slab_.used - slab_.shipped >= ship_bytes) slab_ship();
  lap(tp, prof_push_[1]);
  {
    // errors the record would raise surface here, at its own Callback, not at some later launch
    Resolved R;
    std::vector<int> gc;
    resolve_batch(*b, &R, &gc);
    if (R.args.n_expr > 0 && !jit_possible())
      throw Error(FDB_ERR_UNSUPPORTED, "computed (projected) columns need the run-time specialised kernel (hiprtc unavailable or disabled)");
  }
  lap(tp, prof_push_[2]);
  pending_rows_ += b->rows;
  pending_bytes_ += b->arena_bytes;
  pending_.push_back(std::move(b));
  if (pending_rows_ >= kFlushRows || pending_bytes_ >= kFlushBytes || pending_.size() >= kFlushRecords) settle();
  lap(tp, prof_push_[3]);
  prof_push_n_++;
}

void Plan::slab_reserve(size_t bytes, void** dev, unsigned char** pinned) {
  constexpr size_t kSlabBytes = (size_t)32 << 20;
  bytes = (bytes + 255) / 256 * 256;
  if (slab_.d != nullptr && slab_.used + bytes > slab_.cap) {
    settle();  // the queued records (pieces of this slab) are scanned; the slab stays alive until the stream is idle
    inflight_slabs_.push_back(slab_);
    slab_ = RecordSlab();
    if (inflight_slabs_.size() >= 8) sync();  // (bounds what a long stream of records without a Finish can hold)
  }
  if (slab_.d == nullptr) {
    slab_.cap = std::max(kSlabBytes, bytes);
    slab_.d = ctx_->dev_alloc(slab_.cap);
    slab_.h = (unsigned char*)ctx_->host_alloc(slab_.cap);
    slab_.used = slab_.shipped = 0;
  }
  *dev = (unsigned char*)slab_.d + slab_.used;
  *pinned = slab_.h + slab_.used;
  slab_.used += bytes;
}

void Plan::slab_ship() {
  if (slab_.d == nullptr || slab_.used <= slab_.shipped) return;
  hip_check(hipMemcpyAsync((unsigned char*)slab_.d + slab_.shipped, slab_.h + slab_.shipped, slab_.used - slab_.shipped, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(record slab)");
  slab_.shipped = slab_.used;
}

void Plan::settle() {
  if (pending_.empty()) return;
  slab_ship();
  // The queued records must outlive the launch (until the next sync), and push_batches itself may synchronise (table
  // migration, growth) — which empties inflight_ — so they are held here until it returns.
  std::vector<std::unique_ptr<DeviceBatch>> batch = std::move(pending_);
  pending_.clear();
  pending_rows_ = 0;
  pending_bytes_ = 0;
  std::vector<const DeviceBatch*> ptrs;
  for (const auto& b : batch) ptrs.push_back(b.get());
  try {
    push_batches(ptrs.data(), (int)ptrs.size());
  } catch (...) {
    (void)hipStreamSynchronize(stream_);  // nothing may still be reading the arenas when `batch` is destroyed
    throw;
  }
  for (auto& b : batch) inflight_.push_back(std::move(b));
}

void Plan::push_batch(const DeviceBatch& b) {
  const DeviceBatch* p = &b;
  push_batches(&p, 1);
}

// Resolves one record against the plan (per-dictionary-entry work only): predicate program + LUTs, group columns
// + key-id LUTs (may assign new key ids), aggregated columns. Nothing is launched.
void Plan::resolve_batch(const DeviceBatch& b, Resolved* Rp, std::vector<int>* batch_gcols) {
  Resolved& R = *Rp;
  std::memset(&R.args, 0, sizeof(R.args));
  FdbScanArgs& a = R.args;
  a.n_rows = b.rows;
  int max_depth = 0;
  R.truths = &truth_cache_;
  R.count_selected = [this, &b](int n) { return count_subtree(b, n); };
  if (filter_root_ >= 0) emit_filter(filter_, filter_root_, b, &R, 0, &max_depth);

  // group columns: every field matched by a matcher, in the record's field order (aggregate.go:286-303)
  for (size_t ci = 0; ci < b.cols.size(); ci++) {
    const DevColumn& c = b.cols[ci];
    bool matched = false;
    for (const GroupMatcher& m : matchers_) if (match_group(m, c.name)) { matched = true; break; }
    if (!matched) continue;
    // HashArray (hashed.go:86-105): dictionary / string / binary by value, int64 / uint64 by identity, bool as 1 / 2; anything else panics
    if (c.kind != ColKind::DICT && c.kind != ColKind::I64 && c.kind != ColKind::U64 && c.kind != ColKind::BOOL)
      throw Error(FDB_ERR_UNSUPPORTED, "group by on column type " + c.format + " (" + c.name + ") is not supported");
    if (c.d_values == nullptr) throw Error(FDB_ERR_INVALID, "column not staged: " + c.name);
    const int kind = c.kind == ColKind::DICT ? 0 : 1;
    const bool key_bool = c.kind == ColKind::BOOL, key_u64 = c.kind == ColKind::U64;
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == c.name) break;
    if (gi == gcols_.size()) {
      if (gcols_.size() >= FDB_MAX_HASH_GCOLS) throw Error(FDB_ERR_UNSUPPORTED, "more than 64 group-by columns");
      GroupColState g;
      g.name = c.name;
      g.kind = kind;
      if (kind == 0) { g.value_format = c.dict->value_format; g.plain = c.dict->plain; }
      g.is_bool = key_bool; g.is_u64 = key_u64;
      g.cap = 1;
      g.stride = 0;
      gcols_.push_back(std::move(g));
    }
    GroupColState& g = gcols_[gi];
    if (g.kind != kind || g.is_bool != key_bool || g.is_u64 != key_u64 || (kind == 0 && g.plain != c.dict->plain))  // (the reference's key builder is typed by the first batch)
      throw Error(FDB_ERR_UNSUPPORTED, "group column " + c.name + " changed type between batches");
    GroupRes gr;
    gr.gi = (int)gi; gr.ci = (int)ci; gr.kind = kind;
    if (kind == 0) gr.lut = g.lut_for(c.dict);
    R.groups.push_back(std::move(gr));
    R.count(b, (int)ci);
    (void)batch_gcols;
  }

  // computed group keys: a plain matcher that names a projection (`… timestamp / 1000 * 1000 as timestamp_bucket` grouped by
  // timestamp_bucket, logictest/testdata/plan/aggregate/window); the Projection appends its columns after the stored ones
  for (const GroupMatcher& m : matchers_) {
    if (m.dynamic) continue;
    const Projection* P = find_projection(m.name);
    if (P == nullptr) continue;
    const int root = resolve_projection(*P, b, &R);
    if (a.expr[root].type != FDB_T_I64 && a.expr[root].type != FDB_T_BOOL && a.expr[root].type != FDB_T_U64)  // HashArray panics on float64 (dynparquet/hashed.go:102-103)
      throw Error(FDB_ERR_UNSUPPORTED, "group by on a float64 expression (" + m.name + ") is not supported");
    const bool is_bool = a.expr[root].type == FDB_T_BOOL, is_u64 = a.expr[root].type == FDB_T_U64;
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == m.name) break;
    if (gi == gcols_.size()) {
      if (gcols_.size() >= FDB_MAX_HASH_GCOLS) throw Error(FDB_ERR_UNSUPPORTED, "more than 64 group-by columns");
      GroupColState g;
      g.name = m.name; g.kind = 1; g.is_bool = is_bool; g.is_u64 = is_u64; g.cap = 1; g.stride = 0;
      gcols_.push_back(std::move(g));
    }
    if (gcols_[gi].kind != 1 || gcols_[gi].is_bool != is_bool || gcols_[gi].is_u64 != is_u64) throw Error(FDB_ERR_UNSUPPORTED, "group column " + m.name + " changed type between batches");
    GroupRes gr;
    gr.gi = (int)gi; gr.ci = -1; gr.kind = 2; gr.expr_root = root;
    R.groups.push_back(std::move(gr));
  }

  // aggregated columns, by exact name (aggregate.go:340-361); all must be present (:367-380)
  a.n_aggs = (int32_t)aggs_.size();
  int found = 0;
  for (size_t j = 0; j < aggs_.size(); j++)
    if (find_projection(aggs_[j].column) != nullptr || b.find(final_stage_ ? aggs_[j].result_name : aggs_[j].column) >= 0) found++;
  if (found == 0 && !aggs_.empty())
    throw Error(FDB_ERR_NOT_FOUND, std::string("aggregate field(s) not found, ") + (final_stage_ ? "final " : "") + "aggregations are not possible without it");
  for (size_t j = 0; j < aggs_.size(); j++) {
    AggState& A = aggs_[j];
    FdbAgg& K = a.aggs[j];
    K.func = A.func;
    K.type = FDB_T_NONE;
    K.slot = -1;
    K.expr = 0;
    K.null_value = A.null_value;
    if (const Projection* P = find_projection(A.column)) {  // sum(value * timestamp): the aggregate reads a computed column
      if (A.func == FDB_AGG_COUNT) continue;  // arr.Len(): nothing is read
      const int root = resolve_projection(*P, b, &R);
      const int32_t t = a.expr[root].type;
      if (t == FDB_T_BOOL || t == FDB_T_U64) throw Error(FDB_ERR_UNSUPPORTED, std::string("unsupported type for ") + agg_name(A.func) + " aggregation, expected int64 or float64");  // aggregate.go:736, :782, :862
      if (A.type == FDB_T_NONE) A.type = t;
      else if (A.type != t) throw Error(FDB_ERR_UNSUPPORTED, "aggregated column " + A.column + " changed type between batches");
      K.type = t;
      K.expr = 1 + root;
      continue;
    }
    const int ci = b.find(final_stage_ ? A.result_name : A.column);
    if (ci < 0) throw Error(FDB_ERR_NOT_FOUND, "aggregate field not found: " + A.column);
    const DevColumn& c = b.cols[(size_t)ci];
    if (A.func == FDB_AGG_COUNT && !final_stage_) continue;  // CountAggregation = arr.Len(): the column is not read (aggregate.go:937-950)
    int32_t t = c.kind == ColKind::I64 ? FDB_T_I64 : c.kind == ColKind::F64 ? FDB_T_F64 : FDB_T_NONE;
    if (A.role == 1 || A.role == 2) {  // ErrUnsupportedIsUniqueType (aggregate.go:679)
      if (c.kind != ColKind::I64) throw Error(FDB_ERR_UNSUPPORTED, "unsupported type for is unique aggregation, expected int64");
    } else if (A.role == 3) {          // ErrUnsupportedAndType (aggregate.go:637); bool columns are staged as int64 1 / 2
      if (c.kind != ColKind::BOOL) throw Error(FDB_ERR_UNSUPPORTED, "unsupported type for is and aggregation, expected bool");
      t = FDB_T_I64;
    }
    if (t == FDB_T_NONE)  // ErrUnsupportedSumType / MinType / MaxType (aggregate.go:736, :782, :862)
      throw Error(FDB_ERR_UNSUPPORTED, std::string("unsupported type for ") + agg_name(A.func) + " aggregation, expected int64 or float64");
    if (A.type == FDB_T_NONE) A.type = t;
    else if (A.type != t) throw Error(FDB_ERR_UNSUPPORTED, "aggregated column " + A.column + " changed type between batches");
    if (c.d_values == nullptr) throw Error(FDB_ERR_INVALID, "column not staged: " + c.name);
    R.count(b, ci);
    K.type = t;
    K.values = c.d_values;
    K.validity = c.d_validity;
    R.agg_col[j] = ci;
    if (A.func == FDB_AGG_COUNT) K.func = FDB_AGG_SUM;  // final stage: COUNT merges by SUM (aggregate.go:965-969)
  }
}

// Assigns the column slots of the load-hoisting kernel. Returns 0 if the record references more columns than the
// kernel has slots (→ sequential kernel), 1 for the single-phase layout (≤ 2 four-byte + ≤ 1 eight-byte columns in
// total, all in c4/c8), 2 for the two-phase layout (filter columns in c4/c8, group-by / aggregate columns in l4/l8).
// `relaxed`: only the limits of the argument block apply (the run-time specialised kernel has no register-resident plan);
// *interp_ok then tells whether the interpreting slot kernel could run this record too.
static int assign_slots(const DeviceBatch& b, Plan::Resolved& R, int first_layout = 1, bool relaxed = false, bool* interp_ok = nullptr) {
  FdbScanArgs& a = R.args;
  // static limits of the slot kernel's register-resident plan
  bool strict = !(a.n_leaves > 6 || a.n_gcols > 2 || a.n_aggs > 6 || a.n_expr > 0);
  if (!strict && !relaxed) return 0;
  if (a.n_leaves > FDB_MAX_LEAVES || a.n_gcols > FDB_MAX_DENSE_GCOLS || a.n_aggs > FDB_MAX_AGGS) return 0;
  auto fail_strict = [&]() -> bool { strict = false; return !relaxed; };
  {
    // postfix program → per-leaf trailing op lists (leaves are always pushed in index order)
    int leaf = -1;
    for (int l = 0; l < FDB_MAX_LEAVES; l++) a.ops_after[l] = 0;
    for (int pc = 0; pc < a.n_code; pc++) {
      const uint8_t op = a.code[pc];
      if (op < 0x80) { if ((int)op != leaf + 1) { if (fail_strict()) return 0; break; } leaf = op; continue; }
      if (leaf < 0) { if (fail_strict()) return 0; break; }
      uint32_t& w = a.ops_after[leaf];
      const uint32_t n = w & 0xFu;
      if (n >= 7) { if (fail_strict()) return 0; break; }
      w = (w & ~0xFu) | (n + 1) | ((op == FDB_CODE_AND ? 1u : 2u) << (4 + 2 * n));
    }
  }
  if (interp_ok != nullptr) *interp_ok = strict;
  struct Pool { FdbColSlot* slots; int32_t* n; int cap; int cols[8]; };
  auto slot_in = [&](Pool& P, int ci, bool need_values) -> int {
    const DevColumn& c = b.cols[(size_t)ci];
    for (int s = 0; s < *P.n; s++)
      if (P.cols[s] == ci) { if (need_values) P.slots[s].values = c.d_values; return s; }
    if (*P.n >= P.cap) return -1;
    P.cols[*P.n] = ci;
    P.slots[*P.n].values = need_values ? c.d_values : nullptr;
    P.slots[*P.n].validity = c.d_validity;
    return (*P.n)++;
  };
  for (int layout = first_layout; layout <= 2; layout++) {
    a.n_c4 = a.n_c8 = a.n_l4 = a.n_l8 = 0;
    Pool e4{a.c4, &a.n_c4, layout == 1 ? 2 : (relaxed ? FDB_ARG_C4 : FDB_MAX_C4), {0}}, e8{a.c8, &a.n_c8, layout == 1 ? 1 : (relaxed ? FDB_ARG_C8 : FDB_MAX_C8), {0}};
    Pool l4{a.l4, &a.n_l4, relaxed ? FDB_ARG_L4 : FDB_MAX_L4, {0}}, l8{a.l8, &a.n_l8, relaxed ? FDB_ARG_L8 : FDB_MAX_L8, {0}};
    bool ok = true;
    for (int l = 0; l < a.n_leaves && ok; l++) {
      if (R.leaf_col[l] < 0) continue;
      a.leaves[l].slot = slot_in(a.leaves[l].wide ? e8 : e4, R.leaf_col[l], a.leaves[l].kind != FDB_LEAF_VALIDITY);
      ok = a.leaves[l].slot >= 0;
    }
    // group-by and aggregate columns: same pools in the single-phase layout, the late pools otherwise (a column the
    // filter also reads is simply loaded a second time — it is in L2 by then)
    for (int g = 0; g < a.n_gcols && ok; g++) { a.gcols[g].slot = slot_in(layout == 1 ? e4 : l4, R.gcol_col[g], true); ok = a.gcols[g].slot >= 0; }
    for (int j = 0; j < a.n_aggs && ok; j++)
      if (R.agg_col[j] >= 0 && a.aggs[j].values != nullptr) { a.aggs[j].slot = slot_in(layout == 1 ? e8 : l8, R.agg_col[j], true); ok = a.aggs[j].slot >= 0; }
    for (int k = 0; k < a.n_expr && ok; k++)  // the columns computed aggregate inputs read share the aggregates' pool
      if (a.expr[k].kind == 0) { a.expr[k].slot = slot_in(layout == 1 ? e8 : l8, R.expr_col[k], true); ok = a.expr[k].slot >= 0; }
    if (ok) {
      if (a.n_c4 > FDB_MAX_C4 || a.n_c8 > FDB_MAX_C8 || a.n_l4 > FDB_MAX_L4 || a.n_l8 > FDB_MAX_L8) strict = false;
      if (interp_ok != nullptr) *interp_ok = strict;
      return layout;
    }
  }
  a.n_c4 = a.n_c8 = a.n_l4 = a.n_l8 = 0;
  return 0;
}

// ≙ Callback for `n` resident records at once: ONE fused scan launch covers all of them (per-launch costs — LDS
// table init, ramp-up, tail, table flush — are paid once per scan instead of once per record).
void Plan::push_batches(const DeviceBatch* const* bs, int n) {
  if (finished_) throw Error(FDB_ERR_STATE, "push after finish");
  // no aggregations but group matchers: `Filter → Distinction` (distinct.go:21-170) — the table only records which key tuples
  // exist; Finish emits them (the reference emits them record by record as they are first seen; the set is the same)
  if (aggs_.empty() && matchers_.empty()) throw Error(FDB_ERR_STATE, "filter-only plan: use fdb_plan_filter / fdb_plan_select");
  for (int i = 0; i < n; i++)
    if (bs[i]->device != device_) throw Error(FDB_ERR_INVALID, "batch lives on a different device than the plan");
  hip_check(hipSetDevice(device_), "hipSetDevice");
  for (int i = 0; i < n; i++) bs[i]->note_reader(stream_);
  mirror_valid_ = false;
  PhaseTimer pt;

  std::vector<Resolved> Rs((size_t)n);
  std::vector<std::vector<int>> part_gcols((size_t)n);
  for (int i = 0; i < n; i++) resolve_batch(*bs[i], &Rs[(size_t)i], &part_gcols[(size_t)i]);
  pt.mark("resolve");
  std::vector<int> live;  // records with rows
  for (int i = 0; i < n; i++) if (bs[i]->rows > 0) live.push_back(i);
  if (live.empty()) return;

  // reproducible float sums (fdb_plan_set_deterministic) exist on the dense path of the specialised kernel only
  bool fixed_order = false;
  if (deterministic) for (const AggState& A : aggs_) if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) fixed_order = true;
  // OrderedAggregate without a table: nothing accumulated yet (or already collecting runs) and the records fit the run kernel
  // (not under fixed_order: groups that a wave or record boundary cuts into several runs are folded with atomics at Finish — such a
  // scan goes on to the dense kernel, or to the explicit refusal below)
  if (ordered_ && ((mode_ == TableMode::DENSE && !state_dirty_ && h_table_ == nullptr) || !runs_.empty())) {
    if (!fixed_order && runs_wanted(bs, Rs, live)) {
      push_hash(bs, Rs, live, /*runs=*/true);
      pt.mark("run scan");
      return;
    }
    runs_to_table();  // (no-op without runs) — from here on the ordinary paths
  }
  // Dense (mixed-radix) table while the key space is small and every key column is a dictionary; otherwise the
  // global hash table (cfg 5: tens of label columns, millions of groups; int64 keys such as time buckets).
  if (mode_ == TableMode::DENSE) {
    bool want_hash = gcols_.size() > FDB_MAX_DENSE_GCOLS;
    uint64_t space = 1;
    for (const GroupColState& g : gcols_) {
      if (g.kind != 0) want_hash = true;
      space *= (uint64_t)g.values.size() + 1;
      if (space > (1ull << 22)) { want_hash = true; break; }
    }
    if (want_hash) switch_to_hash();
  }
  if (mode_ == TableMode::HASH) {
    if (fixed_order) throw Error(FDB_ERR_UNSUPPORTED, "deterministic float sums: this scan needs the hash table (too many groups or non-dictionary keys)");
    push_hash(bs, Rs, live);
    pt.mark("hash scan");
    return;
  }
  for (int i : live) {  // dense path: group columns → mixed-radix digits through per-dictionary LUTs
    Resolved& R = Rs[(size_t)i];
    FdbScanArgs& a = R.args;
    for (const GroupRes& gr : R.groups) {
      FdbGroupCol& G = a.gcols[a.n_gcols];
      const DevColumn& c = bs[i]->cols[(size_t)gr.ci];
      G.idx = (const uint32_t*)c.d_values;
      G.validity = c.d_validity;
      G.lut_len = (uint32_t)gr.lut->size();
      G.lut_lds = FDB_NO_LDS;
      G.slot = -1;
      R.gcol_col[a.n_gcols] = gr.ci;
      const size_t off = R.blob.add(gr.lut->data(), gr.lut->size() * 4);
      R.luts.push_back(PendingLut{1, a.n_gcols, off, gr.lut->size() * 4});
      part_gcols[(size_t)i].push_back(gr.gi);
      a.n_gcols++;
    }
  }
  std::vector<uint32_t> caps;
  for (const GroupColState& g : gcols_) caps.push_back((uint32_t)g.values.size() + 1);
  ensure_layout(caps);
  pt.mark("layout");

  // one blob for every LUT of every record
  Blob blob;
  std::vector<size_t> blob_base((size_t)n, 0);
  std::vector<int> lut_class((size_t)n, 0);
  {
    // records whose LUT sets are byte-identical (the usual case: parts of one table share dictionaries) share one
    // device copy and one class id, so the kernel re-stages LUTs in LDS only when the class changes
    std::vector<int> reps;
    for (int i : live) {
      const Resolved& R = Rs[(size_t)i];
      int found = -1;
      for (int r : reps) {
        const Resolved& Q = Rs[(size_t)r];
        if (Q.blob.bytes == R.blob.bytes && Q.luts.size() == R.luts.size()) {
          bool same = true;
          for (size_t k = 0; k < R.luts.size() && same; k++)
            same = Q.luts[k].kind == R.luts[k].kind && Q.luts[k].index == R.luts[k].index && Q.luts[k].blob_off == R.luts[k].blob_off &&
                   Q.luts[k].len_bytes == R.luts[k].len_bytes;
          if (same) { found = r; break; }
        }
      }
      if (found >= 0) { blob_base[(size_t)i] = blob_base[(size_t)found]; lut_class[(size_t)i] = lut_class[(size_t)found]; }
      else { blob_base[(size_t)i] = blob.add(R.blob.bytes.data(), R.blob.bytes.size()); lut_class[(size_t)i] = (int)reps.size(); reps.push_back(i); }
    }
  }
  // the LUT blob and the argument blocks below go to the device with ONE copy command, issued right before the launch
  struct StageScope {
    Context* c;
    explicit StageScope(Context* ctx) : c(ctx) { c->defer_staging(true); }
    ~StageScope() { try { c->defer_staging(false); } catch (...) {} }
  } stage_scope(ctx_);
  unsigned char* d_blob = blob.bytes.empty() ? nullptr : (unsigned char*)upload(blob.bytes.data(), blob.bytes.size());

  const size_t acc_bytes = align_up((size_t)n_slots_ * 4, 16) + (size_t)n_slots_ * 8 * aggs_.size();
  size_t lut_lds_max = 0;
  bool slots_ok = rows_per_thread == 0;
  const bool jit_possible = this->jit_possible();
  bool interp_ok = true;  // every record also fits the interpreting slot kernel's limits
  std::vector<int> layouts;
  for (int i : live) {
    Resolved& R = Rs[(size_t)i];
    FdbScanArgs& a = R.args;
    for (int g = 0; g < a.n_gcols; g++) a.gcols[g].stride = gcols_[(size_t)part_gcols[(size_t)i][(size_t)g]].stride;
    for (size_t j = 0; j < aggs_.size(); j++) a.aggs[j].acc = aggs_[j].d_acc;
    a.cnt = d_cnt_;
    a.n_slots = n_slots_;
    a.need_count = 0;
    a.ablate = ablate;
    for (size_t j = 0; j < aggs_.size(); j++) if (a.aggs[j].func == FDB_AGG_COUNT) a.need_count = 1;
    a.lut_class = lut_class[(size_t)i];
    // LDS plan: [LUT copies][cnt u32 × n_slots][acc u64 × n_slots × n_aggs]
    size_t lds_off = 0;
    for (const PendingLut& p : R.luts) {
      const bool in_lds = p.len_bytes <= 16384 && lds_off + p.len_bytes <= 32768;
      uint32_t lds = FDB_NO_LDS;
      if (in_lds) { lds = (uint32_t)lds_off; lds_off = align_up(lds_off + p.len_bytes, 16); }
      unsigned char* at = d_blob + blob_base[(size_t)i] + p.blob_off;
      if (p.kind == 0) { a.leaves[p.index].lut = at; a.leaves[p.index].lut_lds = lds; }
      else { a.gcols[p.index].lut = (const uint32_t*)at; a.gcols[p.index].lut_lds = lds; }
    }
    lut_lds_max = std::max(lut_lds_max, align_up(lds_off, 16));
    if (slots_ok) {
      bool strict = true;
      const int layout = assign_slots(*bs[i], R, 1, jit_possible, &strict);
      if (layout == 0) slots_ok = false;
      interp_ok = interp_ok && strict;
      layouts.push_back(layout);
    }
  }
  int lds_acc = 0;
  size_t lds_bytes = lut_lds_max;
  int base_grid = grid_override > 0 ? grid_override : fdb_scan_default_grid(device_);
  if (fixed_order) {
    // a table per wave, 4 waves (256-thread workgroups); no combining cache, no interpreting or sequential kernel
    if (!jit_possible || !slots_ok || !use_partials) throw Error(FDB_ERR_UNSUPPORTED, "deterministic float sums need the run-time specialised dense kernel");
    if (lut_lds_max + 4 * acc_bytes > 150 * 1024) throw Error(FDB_ERR_UNSUPPORTED, "deterministic float sums: four per-wave tables of " + std::to_string(acc_bytes) + " bytes do not fit LDS");
    lds_acc = 1; lds_bytes += 4 * acc_bytes;
  }
  else if (lut_lds_max + acc_bytes <= FDB_LDS_BUDGET) { lds_acc = 1; lds_bytes += acc_bytes; }
  else if (lut_lds_max + acc_bytes <= 150 * 1024) { lds_acc = 1; lds_bytes += acc_bytes; if (grid_override <= 0) base_grid /= 2; }
  // table too big for LDS: the specialised kernel gets a combining cache instead (JitShape::cache), ≤ 48 KiB per workgroup
  int cache_slots = 0;
  if (!lds_acc && jit_possible) {
    const size_t entry = 8 + 8 * aggs_.size();
    cache_slots = 1;
    while ((size_t)cache_slots * 2 * entry <= 48 * 1024) cache_slots *= 2;
    lds_bytes = lut_lds_max + (size_t)cache_slots * entry;
  }
  for (int i : live) {
    Rs[(size_t)i].args.lds_lut_bytes = (uint32_t)lut_lds_max; Rs[(size_t)i].args.lds_acc = lds_acc; Rs[(size_t)i].args.cache_slots = cache_slots;
  }
  pt.mark("lut upload");

  int32_t funcs[1 + FDB_MAX_AGGS] = {0};
  funcs[0] = 1;
  {
    const FdbScanArgs& a0 = Rs[(size_t)live[0]].args;
    for (size_t j = 0; j < aggs_.size(); j++) {
      const int32_t f = a0.aggs[j].func;
      const int32_t ty = aggs_[j].type;
      funcs[1 + j] = f == FDB_AGG_COUNT ? 0 : f == FDB_AGG_SUM ? (ty == FDB_T_F64 ? 2 : 1) : f == FDB_AGG_MIN ? 3 : 4;
    }
  }
  auto timed_launch = [&](const std::function<void()>& launch) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
      e0 = ctx_->get_event(); e1 = ctx_->get_event();
      hip_check(hipEventRecord(e0, stream_), "hipEventRecord");
    }
    launch();
    if (timing) {
      hip_check(hipEventRecord(e1, stream_), "hipEventRecord");
      pending_events_.emplace_back(e0, e1);
    }
  };
  auto alloc_partials = [&](int grid) -> unsigned long long* {
    if (!lds_acc || !use_partials || grid <= 0) return nullptr;
    void* p = ctx_->dev_alloc((size_t)grid * (1 + aggs_.size()) * n_slots_ * 8);
    scratch_.push_back(p);
    return (unsigned long long*)p;
  };

  // ---- choose the scan kernel: run-time specialised (fdb_jit.cpp) → interpreting slot kernel → sequential kernel ----------
  int two_phase = 0, tile_rows_i = 0, per_cu = 1;
  const int sub = sub_tiles == 4 ? 0 : sub_tiles;  // kernel variant mode (0 = default, 4 = interpreting kernel only)
  hipFunction_t jit_fn = nullptr;
  int jit_block = sub == 1 ? 512 : sub == 2 ? 256 : sub == 3 ? 1024 : 0;  // 0: picked by occupancy
  if (fixed_order) jit_block = 256;
  if (slots_ok) {
    // records that fit the single-phase layout also fit the two-phase one: re-assign them if the launch is mixed
    for (int l : layouts) if (l == 2) two_phase = 1;
    if (two_phase) {
      size_t k = 0;
      for (int i : live)
        if (layouts[k++] == 1 && assign_slots(*bs[i], Rs[(size_t)i], 2, jit_possible) != 2) throw Error(FDB_ERR_INVALID, "internal: slot re-assignment failed");
    }
    // A kernel specialised for this plan shape (fdb_jit.cpp), when every record of the launch has the same shape;
    // otherwise (or when hiprtc is unavailable) the interpreting slot kernel.
    if (sub_tiles != 4 && ablate == 0 && lds_bytes <= 150 * 1024) {  // (gfx950: up to 160 KiB of LDS per workgroup)
      JitShape shape;
      bool same = true, first = true;
      const FdbScanArgs& args0 = Rs[(size_t)live[0]].args;
      for (int i : live) {
        if (first) { shape = jit_shape(args0, two_phase != 0, jit_block ? jit_block : 256); first = false; }
        else if (!jit_shape_merge_args(&shape, args0, Rs[(size_t)i].args, two_phase != 0)) { same = false; break; }
      }
      if (!same && live.size() > 1 && jit_possible) {
        // Records of different shapes (schema drift: a filter column missing here, a predicate value absent from that part's
        // dictionary there) cannot share one specialised kernel: the launch is split into one launch per shape, so that a
        // few odd parts do not push the whole scan onto the interpreting kernel.
        std::vector<std::string> keys;
        std::vector<std::vector<const DeviceBatch*>> groups;
        for (int i : live) {
          const std::string k = jit_shape(Rs[(size_t)i].args, two_phase != 0, 256).key(false);
          size_t g = 0;
          for (; g < keys.size(); g++) if (keys[g] == k) break;
          if (g == keys.size()) { keys.push_back(k); groups.emplace_back(); }
          groups[g].push_back(bs[i]);
        }
        if (groups.size() > 1) {
          ctx_->defer_staging(false);  // (each group stages its own tables)
          for (auto& g : groups) push_batches(g.data(), (int)g.size());
          return;
        }
      }
      {
        // tiny tables live in registers (JitShape::reg_slots): ≤ 8 slots and ≤ 48 accumulator registers per lane
        int regs_per_slot = 2;
        for (const AggState& A : aggs_) if (A.func != FDB_AGG_COUNT || final_stage_) regs_per_slot += 2;
        if (n_slots_ >= 1 && n_slots_ <= 8 && (int)n_slots_ * regs_per_slot <= 48) shape.reg_slots = (int)n_slots_;
      }
      shape.wave_tables = fixed_order;
      shape.uniform_fold = !knobs_.no_uniform_fold;
      if (same) {
        // ($FDB_JIT_ASYNC=1: a shape whose kernel still has to be built is scanned by the interpreting kernel meanwhile — when that one can)
        JitDeferScope defer(interp_ok && !fixed_order);
        if (jit_block != 0) { jit_fn = jit_get(shape); if (jit_fn != nullptr) per_cu = jit_blocks_per_cu(jit_fn, jit_block, lds_bytes); }
        else {
          int row_bytes = 0;
          for (int k = 0; k < shape.n_c4; k++) row_bytes += shape.c4[k].has_values ? 4 : 0;
          for (int k = 0; k < shape.n_l4; k++) row_bytes += shape.l4[k].has_values ? 4 : 0;
          for (int k = 0; k < shape.n_c8; k++) row_bytes += shape.c8[k].has_values ? 8 : 0;
          for (int k = 0; k < shape.n_l8; k++) row_bytes += shape.l8[k].has_values ? 8 : 0;
          jit_fn = jit_select(shape, lds_bytes, row_bytes, &jit_block, &per_cu);
        }
      }
    }
    // a launch only the specialised kernel could serve (more leaves / aggregates than the interpreting kernel's register-resident
    // plan holds, computed columns) falls back to the sequential kernel when specialisation is unavailable
    if (jit_fn == nullptr && !interp_ok) slots_ok = false;
  }
  if (fixed_order && jit_fn == nullptr) throw Error(FDB_ERR_UNSUPPORTED, "deterministic float sums need the run-time specialised dense kernel (records of one shape, hiprtc available)");

  pt.mark("kernel select");
  if (slots_ok) {
    // ---- one launch of the slot kernel (specialised or interpreting) over every record -------------------------------------
    if (jit_fn != nullptr) {
      tile_rows_i = jit_block * 4;
    } else {
      fdb_slot_geometry(two_phase, sub, lds_acc, lds_bytes, device_, &tile_rows_i, &per_cu);
    }
    int grid = grid_override > 0 ? grid_override : (fdb_scan_default_grid(device_) / 2) * per_cu;
    const int64_t tile_rows = tile_rows_i;
    int64_t total_tiles = 0;
    std::vector<FdbScanArgs> parts;
    parts.reserve(live.size());  // (3 KB apiece: no re-growing copies)
    for (int i : live) {
      FdbScanArgs& a = Rs[(size_t)i].args;
      a.tile_begin = total_tiles;
      total_tiles += (a.n_rows + tile_rows - 1) / tile_rows;
      a.tile_end = total_tiles;
      parts.push_back(a);
    }
    if (grid > total_tiles) grid = (int)total_tiles;
    unsigned long long* partials = alloc_partials(grid);
    for (FdbScanArgs& a : parts) a.partials = partials;
    bool fill_rides = false;
    if (state_virgin_ && jit_fn != nullptr && partials != nullptr && (uint64_t)slots_alloc_ * (1 + aggs_.size()) < (1ull << 32)) {
      // nothing accumulates into the table before reduce_partials (the next kernel on the stream): the scan fills it on its way in
      FdbScanArgs& a0 = parts[0];
      a0.fill_state = d_state_;
      a0.fill_words = (uint32_t)(slots_alloc_ * (1 + aggs_.size()));
      a0.fill_alloc = (uint32_t)slots_alloc_;
      state_idents(a0.fill_idents);
      fill_rides = true;
    } else {
      materialize_state();
    }
    trace("before scan");
    pt.mark("parts build");
    const FdbScanArgs* d_parts = (const FdbScanArgs*)upload(parts.data(), parts.size() * sizeof(FdbScanArgs));
    ctx_->flush_staging();
    pt.mark("parts upload");
    trace("after upload");
    timed_launch([&] {
      if (jit_fn != nullptr) hip_check(jit_launch(jit_fn, d_parts, (int)parts.size(), total_tiles, parts[0], grid, jit_block, lds_bytes, stream_), "scan launch");
      else hip_check(fdb_launch_scan_slots(d_parts, (int)parts.size(), parts[0], total_tiles, grid, lds_bytes, two_phase, sub, stream_), "scan launch");
    });
    if (fill_rides) state_virgin_ = false;  // (only once the launch that carries the fill was accepted: a throw above leaves the table marked unfilled)
    last_kernel_ = jit_fn != nullptr ? "fdb_plan_kernel" : "scan_slots_kernel";
    pt.mark("scan launch");
    trace("after scan");
    if (partials != nullptr) {
      unsigned long long* host_out = mirror_target();
      hip_check(fdb_launch_reduce_partials(partials, grid, (int)(1 + aggs_.size()), n_slots_, d_state_, slots_alloc_, funcs, host_out, stream_), "reduce partials");
      mirror_valid_ = host_out != nullptr;
    }
    trace("after reduce");
    stat_launches += 1;
  } else {
    // ---- sequential kernel, one launch per record ----------------------------------------------------------------
    const int rpt = rows_per_thread == 8 ? 8 : 4;
    materialize_state();
    ctx_->flush_staging();
    for (int i : live) {
      FdbScanArgs& a = Rs[(size_t)i].args;
      if (a.n_expr > 0) throw Error(FDB_ERR_UNSUPPORTED, "computed (projected) columns are not supported by the sequential kernel (too many referenced columns)");
      a.n_c4 = a.n_c8 = 0;
      const int grid = fdb_scan_grid(a, base_grid, rpt);
      a.partials = alloc_partials(grid);
      timed_launch([&] { hip_check(fdb_launch_scan_dense(a, grid, lds_bytes, rpt, stream_), "scan launch"); });
      last_kernel_ = "scan_dense_kernel";
      mirror_valid_ = false;  // (a launch that flushes with atomics leaves the host copy behind)
      if (a.partials != nullptr) {
        unsigned long long* host_out = mirror_target();
        hip_check(fdb_launch_reduce_partials(a.partials, grid, (int)(1 + aggs_.size()), n_slots_, d_state_, slots_alloc_, funcs, host_out, stream_), "reduce partials");
        mirror_valid_ = host_out != nullptr;
      }
      stat_launches += 1;
    }
  }
  pt.mark("reduce launch");
  state_dirty_ = true;
  for (int i : live) { stat_bytes += Rs[(size_t)i].bytes; stat_rows += bs[i]->rows; }
}

// ---- finish / export ----------------------------------------------------------------------------------------
void Plan::fetch_state(std::vector<unsigned long long>* cnt, std::vector<std::vector<unsigned long long>>* acc) {
  hip_check(hipSetDevice(device_), "hipSetDevice");
  acc->assign(aggs_.size(), {});
  if (d_state_ == nullptr) { sync(); cnt->clear(); return; }
  materialize_state();
  if (mirror_valid_ && h_mirror_ != nullptr) {  // the fold kernel already wrote the table to pinned memory: one wait, no copy
    sync();
    cnt->assign(h_mirror_, h_mirror_ + n_slots_);
    for (size_t j = 0; j < aggs_.size(); j++) (*acc)[j].assign(h_mirror_ + (1 + j) * slots_alloc_, h_mirror_ + (1 + j) * slots_alloc_ + n_slots_);
    return;
  }
  // one device→pinned copy of the whole table, then one wait
  const size_t n_arrays = 1 + aggs_.size();
  const size_t bytes = (size_t)slots_alloc_ * 8 * n_arrays;
  unsigned long long* h = (unsigned long long*)ctx_->host_alloc(bytes);
  hipError_t e = hipMemcpyAsync(h, d_state_, bytes, hipMemcpyDeviceToHost, stream_);
  if (e != hipSuccess) { ctx_->host_free(h); hip_check(e, "hipMemcpyAsync(state)"); }
  trace("after state copy");
  try { sync(); } catch (...) { ctx_->host_free(h); throw; }
  cnt->assign(h, h + n_slots_);
  for (size_t j = 0; j < aggs_.size(); j++) (*acc)[j].assign(h + (1 + j) * slots_alloc_, h + (1 + j) * slots_alloc_ + n_slots_);
  ctx_->host_free(h);
}

// Occupied groups of the table, in a mode-independent host form (dense: enumerate slots with count > 0 and decode
// the mixed-radix digits; hash: compact on the device, copy, decode key tuples).
void Plan::fetch_compact(CompactState* cs) {
  runs_to_table();
  cs->n = 0;
  cs->cnt.clear();
  cs->acc.assign(aggs_.size(), {});
  cs->ids.assign(gcols_.size(), {});
  cs->ivals.assign(gcols_.size(), {});
  cs->ivalid.assign(gcols_.size(), {});
  if (mode_ == TableMode::HASH) { fetch_compact_hash(cs); return; }
  std::vector<unsigned long long> cnt;
  std::vector<std::vector<unsigned long long>> acc;
  fetch_state(&cnt, &acc);
  std::vector<uint32_t> slots;
  for (uint32_t s = 0; s < cnt.size(); s++) if (cnt[s] != 0) slots.push_back(s);
  const size_t n = slots.size();
  cs->n = (int64_t)n;
  cs->cnt.resize(n);
  for (size_t i = 0; i < n; i++) cs->cnt[i] = cnt[slots[i]];
  for (size_t j = 0; j < aggs_.size(); j++) {
    cs->acc[j].resize(n);
    for (size_t i = 0; i < n; i++) cs->acc[j][i] = acc[j][slots[i]];
  }
  for (size_t c = 0; c < gcols_.size(); c++) {
    const GroupColState& g = gcols_[c];
    cs->ids[c].resize(n);
    for (size_t i = 0; i < n; i++) cs->ids[c][i] = g.cap > 1 ? (slots[i] / g.stride) % g.cap : 0;
  }
}

// OrderedAggregate emits its groups in key order: the merge of its ordered sets sorts by every group column in first-seen
// order, ascending, NULLs last (ordered_aggregate.go:449-470; cursorHeap.Less, arrowutils/merge.go:84-112: SortingColumn's zero
// value is ascending / NullsFirst = false; binary and string keys compare bytewise, int64 numerically). Dictionary keys are
// compared through the RANK of their key id among the column's distinct values, computed once per column.
void Plan::sort_compact(CompactState* cs) const {
  const size_t n = (size_t)cs->n;
  if (n < 2) return;
  std::vector<std::vector<uint32_t>> rank(gcols_.size());
  for (size_t c = 0; c < gcols_.size(); c++) {
    const GroupColState& g = gcols_[c];
    if (g.kind != 0) continue;
    std::vector<uint32_t> order(g.values.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return g.values[a] < g.values[b]; });
    rank[c].assign(g.values.size() + 1, 0xFFFFFFFFu);  // id 0 (NULL) sorts last
    for (size_t r = 0; r < order.size(); r++) rank[c][order[r] + 1] = (uint32_t)r;
  }
  std::vector<size_t> perm(n);
  for (size_t i = 0; i < n; i++) perm[i] = i;
  std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) {
    for (size_t c = 0; c < gcols_.size(); c++) {
      const GroupColState& g = gcols_[c];
      if (g.kind == 0) {
        const uint32_t ra = cs->ids[c].empty() ? 0xFFFFFFFFu : rank[c][cs->ids[c][a]], rb = cs->ids[c].empty() ? 0xFFFFFFFFu : rank[c][cs->ids[c][b]];
        if (ra != rb) return ra < rb;
      } else {
        const bool va = !cs->ivalid[c].empty() && cs->ivalid[c][a], vb = !cs->ivalid[c].empty() && cs->ivalid[c][b];
        if (va != vb) return va;  // NULLs last
        if (!va) continue;
        const int64_t xa = cs->ivals[c][a], xb = cs->ivals[c][b];
        if (xa != xb) return g.is_u64 ? (uint64_t)xa < (uint64_t)xb : xa < xb;
      }
    }
    return false;
  });
  auto apply = [&](auto& v) {
    if (v.empty()) return;
    auto copy = v;
    for (size_t i = 0; i < n; i++) v[i] = copy[perm[i]];
  };
  apply(cs->cnt);
  for (auto& a : cs->acc) apply(a);
  for (auto& a : cs->ids) apply(a);
  for (auto& a : cs->ivals) apply(a);
  for (auto& a : cs->ivalid) apply(a);
}

void Plan::build_key_columns(const CompactState& cs, std::vector<OutColumn>* cols) const {
  const int64_t n = cs.n;
  for (size_t gc = 0; gc < gcols_.size(); gc++) {
    const GroupColState& g = gcols_[gc];
    OutColumn c;
    c.name = g.name;
    c.length = n;
    c.validity.assign((size_t)(n + 7) / 8, 0);
    if (g.kind == 1 && g.is_bool) {  // bool key (stored column or boolean projection): 1 = false, 2 = true on the device → bits
      c.format = "b";
      c.values.assign((size_t)(n + 7) / 8 + 8, 0);
      for (int64_t i = 0; i < n; i++) {
        if (!cs.ivalid[gc][(size_t)i]) { c.null_count++; continue; }
        c.validity[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
        if (cs.ivals[gc][(size_t)i] >= 2) c.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
      }
      cols->push_back(std::move(c));
      continue;
    }
    if (g.kind == 1) {  // int64 / uint64 key column
      c.format = g.is_u64 ? "L" : "l";
      c.values.resize((size_t)n * 8);
      for (int64_t i = 0; i < n; i++) {
        const int64_t v = cs.ivalid[gc][(size_t)i] ? cs.ivals[gc][(size_t)i] : 0;
        std::memcpy(c.values.data() + (size_t)i * 8, &v, 8);
        if (cs.ivalid[gc][(size_t)i]) c.validity[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7)); else c.null_count++;
      }
      cols->push_back(std::move(c));
      continue;
    }
    c.format = "I";
    c.values.resize((size_t)n * 4);
    uint32_t* idx = (uint32_t*)c.values.data();
    for (int64_t i = 0; i < n; i++) {
      const uint32_t id = cs.ids[gc][(size_t)i];
      if (id == 0) { idx[i] = 0; c.null_count++; }
      else { idx[i] = id - 1; c.validity[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7)); }
    }
    if (g.plain) {  // the key column keeps its input type (array.NewBuilder(type), pqarrow/builder/utils.go:12-52)
      const std::vector<uint8_t> idx_bytes = std::move(c.values);
      set_plain_strings(&c, (const uint32_t*)idx_bytes.data(), c.validity.data(), n, g.values, g.value_format);
    } else {
      set_dictionary(&c, g.values, g.value_format);
    }
    cols->push_back(std::move(c));
  }
}

void Plan::build_agg_columns(const CompactState& cs, std::vector<OutColumn>* cols) const {
  const int64_t n = cs.n;
  for (size_t j = 0; j < aggs_.size(); j++) {
    const AggState& A = aggs_[j];
    if (A.role == 2) continue;  // the MAX half of UNIQUE: consumed with its MIN half
    OutColumn c;
    c.name = A.emit_name;
    c.length = n;
    if (A.role == 1) {  // UNIQUE: min == max ⇒ the value, else NULL (uniqueInt64arrays, aggregate.go:694-710)
      c.format = "l";
      c.values.assign((size_t)n * 8, 0);
      c.validity.assign((size_t)(n + 7) / 8, 0);
      for (int64_t i = 0; i < n; i++) {
        const unsigned long long lo = cs.acc[j][(size_t)i], hi = cs.acc[j + 1][(size_t)i];
        if (lo == hi) { std::memcpy(c.values.data() + (size_t)i * 8, &lo, 8); c.validity[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7)); }
        else c.null_count++;
      }
      cols->push_back(std::move(c));
      continue;
    }
    if (A.role == 3) {  // AND: a bool column (AndArrays, aggregate.go:654-675); no valid value ⇒ true
      c.format = "b";
      c.values.assign((size_t)(n + 7) / 8 + 8, 0);
      for (int64_t i = 0; i < n; i++)
        if (cs.acc[j][(size_t)i] >= 2ull) c.values[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7));  // MIN over 1 (false) / 2 (true)
      cols->push_back(std::move(c));
      continue;
    }
    c.values.resize((size_t)n * 8);
    const bool count_from_cnt = A.func == FDB_AGG_COUNT && !final_stage_;
    const bool is_f64 = !count_from_cnt && A.type == FDB_T_F64;
    c.format = is_f64 ? "g" : "l";
    for (int64_t i = 0; i < n; i++) {
      unsigned long long v;
      if (count_from_cnt) v = cs.cnt[(size_t)i];
      else {
        v = cs.acc[j][(size_t)i];
        if (is_f64 && (A.func == FDB_AGG_MIN || A.func == FDB_AGG_MAX)) {
          const double d = fdb_ordered_to_f64_host((int64_t)v);
          std::memcpy(&v, &d, 8);
        }
      }
      std::memcpy(c.values.data() + (size_t)i * 8, &v, 8);
    }
    cols->push_back(std::move(c));
  }
}

// An ordered plan whose groups sit in the hash table: from a few thousand groups on they are sorted on the device (finish_columns_hash), below
// that on the host (sort_compact: a handful of launches and a round trip cost more than sorting a few rows). $FDB_ORDERED_SORT_MIN moves
// the threshold (0: always on the device — tests).
bool Plan::ordered_finish_on_device() {
  if (!ordered_ || mode_ != TableMode::HASH || h_table_ == nullptr) return false;
  const uint64_t least = (uint64_t)knobs_.ordered_sort_min;
  const uint64_t n = hash_groups();
  const bool on_device = n >= least && n >= 2 && n <= ((uint64_t)1 << 28);
  fresh_groups_ = on_device ? (int64_t)n : -1;  // (finish_columns_hash follows at once and does not fetch the count a second time)
  return on_device;
}

int64_t Plan::finish_columns(std::vector<OutColumn>* cols) {
  PhaseTimer pt;
  cols->clear();
  int64_t n = 0;
  if (!runs_.empty()) {
    bool ok = false;
    n = finish_columns_runs(cols, nullptr, &ok);
    if (ok) { finished_ = true; return n; }
    cols->clear();  // the keys were not in order: the runs are in the hash table now, the ordinary ordered Finish below sorts them
  }
  if (mode_ == TableMode::HASH && h_table_ != nullptr && (!ordered_ || ordered_finish_on_device())) {
    // big result sets: columns are materialised on the device (an ordered plan's: sorted there first), the host only copies finished Arrow buffers
    n = finish_columns_hash(cols);
    pt.mark("finish: columns");
  } else {
    CompactState cs;
    fetch_compact(&cs);
    pt.mark("finish: fetch");
    if (ordered_) sort_compact(&cs);
    build_key_columns(cs, cols);
    build_agg_columns(cs, cols);
    n = cs.n;
  }
  finished_ = true;
  return n;
}

void Plan::finish(ArrowArray* out, ArrowSchema* out_schema, int64_t* n_rows) {
  PhaseTimer pt;
  std::vector<OutColumn> cols;
  int64_t n = finish_columns(&cols);
  // ---- several records where one would pass the key builders' size limit (aggregate.go:426-468) -------------------------------
  // The reference appends a new group's keys to the CURRENT aggregate's builders and, when a binary builder answers ErrMaxSizeReached
  // (data + value > math.MaxInt32, optbuilders.go:221-224), rolls the group back and starts a new aggregate — a new output record —
  // with it. Here the groups are cut the same way at Finish, in output order: a record ends in front of the first row whose value
  // would take one of its plain string / binary key columns past the limit. (Which groups share a record differs — the reference
  // cuts in arrival order —, the union of the records and the per-record bound are the same.)
  pending_out_.clear();
  const int64_t limit = knobs_.max_key_bytes;
  // (only a plain BINARY column's builder has the limit: builder.NewBuilder gives *arrow.BinaryType an OptBinaryBuilder and every other
  // type arrow's own builder, pqarrow/builder/utils.go:12-24 — a utf8 or large_binary key column keeps one record, with 64-bit offsets past 2 GiB)
  auto limited = [&](size_t c) { return c < gcols_.size() && cols[c].is_str && gcols_[c].plain && gcols_[c].value_format == "z"; };
  std::vector<size_t> str_cols;
  for (size_t c = 0; c < cols.size(); c++)
    if (limited(c) && n > 0 && plain_string_total(cols[c], n) > limit) str_cols.push_back(c);
  if (!str_cols.empty()) {
    str_cols.clear();
    for (size_t c = 0; c < cols.size(); c++) if (limited(c)) str_cols.push_back(c);
    std::vector<int64_t> cuts{0};
    std::vector<int64_t> held(str_cols.size(), 0);
    for (int64_t i = 0; i < n; i++) {
      bool cut = false;
      for (size_t k = 0; k < str_cols.size(); k++) {
        const int64_t b = plain_string_bytes(cols[str_cols[k]], i);
        if (b > limit) throw Error(FDB_ERR_INVALID, "max size reached: one value of group column " + cols[str_cols[k]].name + " is larger than a record's key builder takes");  // (the retry in the new aggregate fails too, aggregate.go:464-466)
        if (held[k] + b > limit) cut = true;
      }
      if (cut) { cuts.push_back(i); std::fill(held.begin(), held.end(), 0); }
      for (size_t k = 0; k < str_cols.size(); k++) held[k] += plain_string_bytes(cols[str_cols[k]], i);
    }
    cuts.push_back(n);
    for (size_t r = cuts.size() - 1; r >= 2; r--)  // (kept in reverse: finish_next pops from the back)
      pending_out_.emplace_back(slice_columns(cols, cuts[r - 1], cuts[r] - cuts[r - 1]), cuts[r] - cuts[r - 1]);
    std::vector<OutColumn> first = slice_columns(cols, 0, cuts[1]);
    n = cuts[1];
    cols = std::move(first);
  }
  if (n_rows) *n_rows = n;
  export_record(std::move(cols), n, out, out_schema);
  pt.mark("finish: export");
}

bool Plan::finish_next(ArrowArray* out, ArrowSchema* out_schema, int64_t* n_rows) {
  if (pending_out_.empty()) { if (n_rows) *n_rows = 0; return false; }
  std::pair<std::vector<OutColumn>, int64_t> rec = std::move(pending_out_.back());
  pending_out_.pop_back();
  if (n_rows) *n_rows = rec.second;
  export_record(std::move(rec.first), rec.second, out, out_schema);
  return true;
}

std::unique_ptr<DeviceBatch> Plan::finish_batch(int64_t* n_rows) {
  if (aggs_.empty() && matchers_.empty()) throw Error(FDB_ERR_STATE, "filter-only plan: nothing to finish");
  settle();
  bool composite = false;
  for (const AggState& A : aggs_) if (A.role != 0) composite = true;  // UNIQUE / AND are finished on the host (validity from two accumulators)
  if (!runs_.empty()) {
    std::unique_ptr<DeviceBatch> b(new DeviceBatch());
    bool ok = false;
    const int64_t n = finish_columns_runs(nullptr, b.get(), &ok);
    if (ok) { if (n_rows) *n_rows = n; return b; }
  }
  if (mode_ == TableMode::HASH && h_table_ != nullptr && (!ordered_ || ordered_finish_on_device()) && !composite) {
    std::unique_ptr<DeviceBatch> b(new DeviceBatch());
    const int64_t n = finish_columns_hash(nullptr, b.get());
    if (n_rows) *n_rows = n;
    return b;
  }
  // small (dense) tables, small ordered results, composite reducers: the ordinary Finish, then the record is made resident
  ArrowArray arr;
  ArrowSchema sch;
  std::memset(&arr, 0, sizeof(arr));
  std::memset(&sch, 0, sizeof(sch));
  int64_t n = 0;
  finish(&arr, &sch, &n);
  struct Rel { ArrowArray* a; ArrowSchema* s; ~Rel() { if (a->release) a->release(a); if (s->release) s->release(s); } } rel{&arr, &sch};
  HostRecordView view;
  view_record(&arr, &sch, &view);
  std::unique_ptr<DeviceBatch> b = import_batch(view, device_, nullptr, stream_);
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  if (n_rows) *n_rows = n;
  return b;
}

int64_t Plan::num_groups() {
  if (mode_ == TableMode::HASH && runs_.empty()) {  // the table counts its own inserts: one 8-byte read behind the stream (a compaction of 10 M groups took 3 s)
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (h_table_ == nullptr || h_count_dev_ == nullptr) { sync(); return 0; }
    return (int64_t)hash_groups();
  }
  CompactState cs;
  fetch_compact(&cs);
  return cs.n;
}

void Plan::partial_keys(ArrowArray* out, ArrowSchema* out_schema) {
  CompactState cs;
  fetch_compact(&cs);
  std::vector<OutColumn> cols;
  build_key_columns(cs, &cols);
  export_record(std::move(cols), cs.n, out, out_schema);
}

char Plan::agg_format(int32_t agg) const {
  if (agg < 0 || agg >= (int32_t)aggs_.size()) throw Error(FDB_ERR_INVALID, "aggregation index out of range");
  const AggState& A = aggs_[(size_t)agg];
  if (A.func == FDB_AGG_COUNT && !final_stage_) return 'l';
  if (A.type == FDB_T_NONE) return 0;
  return A.type == FDB_T_F64 ? 'g' : 'l';
}

int32_t Plan::state_array_op(int32_t array) const {
  if (array < 0 || array > (int32_t)aggs_.size()) throw Error(FDB_ERR_INVALID, "table array index out of range");
  if (array == 0) return 1;
  const AggState& A = aggs_[(size_t)array - 1];
  if (A.func == FDB_AGG_COUNT) return final_stage_ ? 1 : 0;
  if (A.func == FDB_AGG_SUM) return A.type == FDB_T_F64 ? 2 : 1;
  return A.func == FDB_AGG_MIN ? 3 : 4;
}

void Plan::partial_state(int32_t agg, void* dst, int64_t capacity_bytes) {
  if (has_composite_aggs()) throw Error(FDB_ERR_UNSUPPORTED, "partial_state: not available for plans with UNIQUE / AND aggregations (merge them through the table arrays)");
  if (agg < 0 || agg >= (int32_t)aggs_.size()) throw Error(FDB_ERR_INVALID, "aggregation index out of range");
  CompactState cs;
  fetch_compact(&cs);
  std::vector<OutColumn> cols;
  build_agg_columns(cs, &cols);
  const OutColumn& c = cols[(size_t)agg];
  if ((int64_t)c.values.size() > capacity_bytes) throw Error(FDB_ERR_INVALID, "partial_state: destination too small");
  if (!c.values.empty()) hip_check(hipMemcpy(dst, c.values.data(), c.values.size(), hipMemcpyDefault), "hipMemcpy(partial_state)");
}

// ---- raw table access for the aligned-layout all-reduce (frostdb_amd/distributed.py) ---------------------------------
uint64_t Plan::state_signature(int64_t* n_slots_out) {
  runs_to_table();
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  auto mix64 = [&](uint64_t v) { mix(&v, 8); };
  mix64(n_slots_);
  mix64(d_state_ != nullptr);
  for (const GroupColState& g : gcols_) {
    mix(g.name.data(), g.name.size()); mix64(g.name.size());
    mix64(g.cap); mix64(g.stride);
    for (const std::string_view& v : g.values) { mix64(v.size()); mix(v.data(), v.size()); }
  }
  for (const AggState& a : aggs_) { mix64((uint64_t)a.func); mix64((uint64_t)a.type); mix(a.column.data(), a.column.size()); }
  if (n_slots_out) *n_slots_out = (d_state_ != nullptr && mode_ == TableMode::DENSE) ? (int64_t)n_slots_ : 0;
  return h;
}

void Plan::state_pointers(void** base, int64_t* array_stride, int64_t* n_slots) {
  runs_to_table();
  const bool ok = mode_ == TableMode::DENSE && d_state_ != nullptr;
  if (ok) { materialize_state(); mirror_valid_ = false; }  // (the caller may write through these pointers)
  *base = ok ? (void*)d_state_ : nullptr;
  *array_stride = ok ? (int64_t)slots_alloc_ : 0;
  *n_slots = ok ? (int64_t)n_slots_ : 0;
}

void Plan::state_read(int32_t array, void* dst, int64_t capacity_bytes) {
  runs_to_table();
  if (mode_ != TableMode::DENSE) throw Error(FDB_ERR_STATE, "raw table access needs the dense table");
  if (array < 0 || array > (int32_t)aggs_.size()) throw Error(FDB_ERR_INVALID, "table array index out of range");
  if (d_state_ == nullptr) throw Error(FDB_ERR_STATE, "the plan has no table yet");
  const size_t bytes = (size_t)n_slots_ * 8;
  if ((int64_t)bytes > capacity_bytes) throw Error(FDB_ERR_INVALID, "state_read: destination too small");
  hip_check(hipSetDevice(device_), "hipSetDevice");
  materialize_state();
  hip_check(hipMemcpyAsync(dst, d_state_ + (size_t)array * slots_alloc_, bytes, hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync(state_read)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

void Plan::state_write(int32_t array, const void* src, int64_t bytes) {
  runs_to_table();
  if (mode_ != TableMode::DENSE) throw Error(FDB_ERR_STATE, "raw table access needs the dense table");
  if (array < 0 || array > (int32_t)aggs_.size()) throw Error(FDB_ERR_INVALID, "table array index out of range");
  if (d_state_ == nullptr) throw Error(FDB_ERR_STATE, "the plan has no table yet");
  if (bytes != (int64_t)n_slots_ * 8) throw Error(FDB_ERR_INVALID, "state_write: size mismatch");
  hip_check(hipSetDevice(device_), "hipSetDevice");
  materialize_state();
  mirror_valid_ = false;
  hip_check(hipMemcpyAsync(d_state_ + (size_t)array * slots_alloc_, src, (size_t)bytes, hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync(state_write)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

// ---- merge (≙ Synchronizer + final-stage HashAggregate, same device) -------------------------------------------
void Plan::merge_from(Plan& src) {
  if (&src == this) throw Error(FDB_ERR_INVALID, "cannot merge a plan into itself");
  if (src.device_ != device_) throw Error(FDB_ERR_INVALID, "merge across devices goes through frostdb_amd.distributed (RCCL)");
  if (src.aggs_.size() != aggs_.size()) throw Error(FDB_ERR_INVALID, "plans have different aggregations");
  for (size_t j = 0; j < aggs_.size(); j++) {
    if (src.aggs_[j].func != aggs_[j].func || src.aggs_[j].column != aggs_[j].column) throw Error(FDB_ERR_INVALID, "plans have different aggregations");
    if (aggs_[j].type == FDB_T_NONE) aggs_[j].type = src.aggs_[j].type;
    else if (src.aggs_[j].type != FDB_T_NONE && src.aggs_[j].type != aggs_[j].type) throw Error(FDB_ERR_INVALID, "aggregation types differ between plans");
  }
  if (merge_runs(src)) return;  // two ordered plans whose state is runs: merged as runs (no table)
  runs_to_table();
  src.runs_to_table();
  if (mode_ == TableMode::HASH || src.mode_ == TableMode::HASH) {
    src.sync();
    sync();
    if (!src.state_dirty_) return;
    merge_hash(src);
    return;
  }
  // Dense into dense — the Synchronizer + final stage of N chains on one GPU (physicalplan.go:438-471), N − 1 times per query: no host
  // round trip until the end. The source's table is ordered in front of our merge kernels by an EVENT (its scan may still be running),
  // every source slot is mapped — an empty one folds its identity into its destination, so the source's counts need not be fetched to
  // know which slots are occupied (that blocking copy and the two waits in front of it were most of a merge: 250 µs, now ≈ 70) — and one
  // wait at the end lets the caller close the source right away.
  settle();
  src.settle();
  if (!src.state_dirty_) { src.sync(); return; }  // (callers close or free `src` right after a merge: its stream is idle on every return)
  src.materialize_state();
  {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hipEvent_t ev = ctx_->get_event();
    struct Put { Context* c; hipEvent_t e; ~Put() { c->put_event(e); } } put{ctx_, ev};
    hip_check(hipEventRecord(ev, src.stream_), "hipEventRecord(merge source)");
    hip_check(hipStreamWaitEvent(stream_, ev, 0), "hipStreamWaitEvent(merge source)");
  }
  // unify key ids
  std::vector<size_t> col_map(src.gcols_.size());
  std::vector<std::vector<uint32_t>> id_map(src.gcols_.size());
  std::vector<uint32_t> caps;
  for (const GroupColState& g : gcols_) caps.push_back((uint32_t)g.values.size() + 1);
  for (size_t sc = 0; sc < src.gcols_.size(); sc++) {
    const GroupColState& sg = src.gcols_[sc];
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == sg.name) break;
    if (gi == gcols_.size()) {
      GroupColState g;
      g.name = sg.name; g.value_format = sg.value_format; g.plain = sg.plain; g.cap = 1; g.stride = 0;
      gcols_.push_back(std::move(g));
      caps.push_back(1);
    }
    GroupColState& g = gcols_[gi];
    if (g.plain != sg.plain) throw Error(FDB_ERR_INVALID, "group column " + sg.name + " has different types in the two plans");
    col_map[sc] = gi;
    id_map[sc].assign(sg.values.size() + 1, 0);
    g.owners.insert(g.owners.end(), sg.owners.begin(), sg.owners.end());
    for (size_t v = 0; v < sg.values.size(); v++) id_map[sc][v + 1] = g.intern(sg.values[v]);
    caps[gi] = (uint32_t)g.values.size() + 1;
  }
  ensure_layout(caps);
  materialize_state();
  mirror_valid_ = false;
  std::vector<uint32_t> map((size_t)src.n_slots_, 0xFFFFFFFFu);
  for (uint32_t s = 0; s < src.n_slots_; s++) {
    uint64_t t = 0;
    bool real = true;  // (a digit beyond the column's values — head-room of the source's layout — is a slot no row can have reached)
    for (size_t sc = 0; sc < src.gcols_.size() && real; sc++) {
      const GroupColState& sg = src.gcols_[sc];
      const uint32_t id = sg.cap > 1 ? (s / sg.stride) % sg.cap : 0;
      if (id >= id_map[sc].size()) { real = false; break; }
      t += (uint64_t)id_map[sc][id] * gcols_[col_map[sc]].stride;
    }
    if (real) map[s] = (uint32_t)t;
  }
  const uint32_t* d_map = (const uint32_t*)upload(map.data(), map.size() * 4);
  hip_check(fdb_launch_merge_u64(d_cnt_, src.d_cnt_, d_map, src.n_slots_, FDB_AGG_SUM, 0, stream_), "merge cnt");
  for (size_t j = 0; j < aggs_.size(); j++) {
    const int32_t f = aggs_[j].func == FDB_AGG_COUNT ? FDB_AGG_SUM : aggs_[j].func;
    hip_check(fdb_launch_merge_u64(aggs_[j].d_acc, src.aggs_[j].d_acc, d_map, src.n_slots_, f, aggs_[j].type == FDB_T_F64 && f == FDB_AGG_SUM, stream_, src.d_cnt_), "merge acc");
  }
  state_dirty_ = true;
  sync();
}

namespace {
struct DevBuf {  // scratch from the context's caching allocator; the plan's stream orders every reuse
  Context* ctx;
  void* p = nullptr;
  DevBuf(Context* c, size_t bytes) : ctx(c), p(c->dev_alloc(bytes)) {}
  ~DevBuf() { ctx->dev_free(p); }
  DevBuf(const DevBuf&) = delete;
};
}  // namespace

// ---- selection / filter-only -------------------------------------------------------------------------------------
// ≙ filter() (filter.go:276-323): the reference turns the predicate's bitmap into index ranges, slices every column per range
// and concatenates the slices. Here ONE kernel pass evaluates the predicate, gives every selected row its output position
// (decoupled look-back over per-tile totals) and writes the compacted columns — see fdb_launch_compact.
// Rows of `b` the filter sub-tree rooted at `node` selects (one launch of the interpreting flags kernel + a wait): only used where
// the reference's lazy AND decides whether an error exists at all (emit_filter).
int64_t Plan::count_subtree(const DeviceBatch& b, int node) {
  if (b.rows == 0) return 0;
  hip_check(hipSetDevice(device_), "hipSetDevice");
  slab_ship();  // (a small pushed record may still sit in the pinned slab: its bytes go first)
  Resolved R;
  std::memset(&R.args, 0, sizeof(R.args));
  R.args.n_rows = b.rows;
  int max_depth = 0;
  R.truths = &truth_cache_;
  R.count_selected = [this, &b](int n) { return count_subtree(b, n); };
  emit_filter(filter_, node, b, &R, 0, &max_depth);
  FdbScanArgs& a = R.args;
  size_t lds_off = 0;
  unsigned char* d_blob = R.blob.bytes.empty() ? nullptr : (unsigned char*)upload(R.blob.bytes.data(), R.blob.bytes.size());
  for (const PendingLut& p : R.luts) {
    const bool in_lds = p.len_bytes <= 16384 && lds_off + p.len_bytes <= 32768;
    uint32_t lds = FDB_NO_LDS;
    if (in_lds) { lds = (uint32_t)lds_off; lds_off = align_up(lds_off + p.len_bytes, 16); }
    a.leaves[p.index].lut = d_blob + p.blob_off;
    a.leaves[p.index].lut_lds = lds;
  }
  a.lds_lut_bytes = (uint32_t)align_up(lds_off, 16);
  ctx_->flush_staging();
  b.note_reader(stream_);
  uint8_t* masks = nullptr;
  uint32_t* offs = nullptr;
  return run_flags(a, &masks, &offs);
}

void Plan::resolve_filter_only(const DeviceBatch& b, Resolved* Rp) {
  Resolved& R = *Rp;
  std::memset(&R.args, 0, sizeof(R.args));
  R.args.n_rows = b.rows;
  int max_depth = 0;
  R.truths = &truth_cache_;
  R.count_selected = [this, &b](int n) { return count_subtree(b, n); };
  emit_filter(filter_, filter_root_, b, &R, 0, &max_depth);
  FdbScanArgs& a = R.args;
  size_t lds_off = 0;
  unsigned char* d_blob = R.blob.bytes.empty() ? nullptr : (unsigned char*)upload(R.blob.bytes.data(), R.blob.bytes.size());
  for (const PendingLut& p : R.luts) {
    const bool in_lds = p.len_bytes <= 16384 && lds_off + p.len_bytes <= 32768;
    uint32_t lds = FDB_NO_LDS;
    if (in_lds) { lds = (uint32_t)lds_off; lds_off = align_up(lds_off + p.len_bytes, 16); }
    a.leaves[p.index].lut = d_blob + p.blob_off;
    a.leaves[p.index].lut_lds = lds;
  }
  a.lds_lut_bytes = (uint32_t)align_up(lds_off, 16);
}

// Steps 1 + 2 of filter(): selection bitmap and per-tile output offsets on the device (scratch of this plan until its next
// sync()), number of selected rows on the host.
int64_t Plan::run_flags(const FdbScanArgs& a, uint8_t** d_masks, uint32_t** d_offsets) {
  const int64_t n_tiles = (a.n_rows + FDB_COMPACT_TILE - 1) / FDB_COMPACT_TILE;
  uint8_t* masks = (uint8_t*)ctx_->dev_alloc((size_t)n_tiles * (FDB_COMPACT_TILE / 8) + 64);
  uint32_t* offs = (uint32_t*)ctx_->dev_alloc((size_t)(n_tiles + 4 + n_tiles / 1024 + 8) * 4 + 64);  // (+ the scan's per-1024 sums)
  unsigned long long* d_total = (unsigned long long*)ctx_->dev_alloc(64);
  scratch_.push_back(masks); scratch_.push_back(offs); scratch_.push_back(d_total);
  hip_check(hipMemsetAsync(offs, 0, (size_t)(n_tiles + 4) * 4, stream_), "hipMemsetAsync(tile counts)");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
  hip_check(fdb_launch_filter_flags(a, masks, offs, device_, stream_), "filter flags launch");
  hip_check(fdb_launch_exclusive_scan(offs, n_tiles, offs + n_tiles + 4, d_total, stream_), "tile offsets launch");
  if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); pending_events_.emplace_back(e0, e1); }
  unsigned long long total = 0;
  hip_check(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(total)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  *d_masks = masks; *d_offsets = offs;
  stat_launches += 1;
  return (int64_t)total;
}

int64_t Plan::select_batch(const DeviceBatch& in, uint32_t* d_indices, int64_t capacity) {
  if (filter_root_ < 0) throw Error(FDB_ERR_STATE, "plan has no filter");
  if (in.device != device_) throw Error(FDB_ERR_INVALID, "batch lives on a different device than the plan");
  if (capacity < in.rows) throw Error(FDB_ERR_INVALID, "indices capacity smaller than the record");
  hip_check(hipSetDevice(device_), "hipSetDevice");
  if (in.rows == 0) return 0;
  Resolved R;
  resolve_filter_only(in, &R);
  uint8_t* masks = nullptr;
  uint32_t* offs = nullptr;
  const int64_t n = run_flags(R.args, &masks, &offs);
  if (n > 0) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
    hip_check(fdb_launch_compact_col(0, nullptr, nullptr, d_indices, nullptr, masks, offs, in.rows, nullptr, device_, stream_), "compact launch");
    if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); pending_events_.emplace_back(e0, e1); }
  }
  last_kernel_ = "compact_col_kernel";
  sync();
  stat_bytes += R.bytes + n * 4;
  stat_rows += in.rows;
  return n;
}

std::unique_ptr<DeviceBatch> Plan::filter_batch(const DeviceBatch& in, int64_t* n_selected) {
  const DeviceBatch* p = &in;
  std::vector<std::unique_ptr<DeviceBatch>> out = filter_batches(&p, 1, n_selected);  // (falls back to filter_batch_interp by itself)
  return std::move(out[0]);
}

// The per-record path: interpreting flags kernel, one compaction launch per column (what runs when the predicate cannot be
// specialised at run time, and the parity reference of the generated kernels).
std::unique_ptr<DeviceBatch> Plan::filter_batch_interp(const DeviceBatch& in, int64_t* n_selected) {
  if (filter_root_ < 0) throw Error(FDB_ERR_STATE, "plan has no filter");
  if (in.device != device_) throw Error(FDB_ERR_INVALID, "batch lives on a different device than the plan");
  if (in.cols.size() > 128) throw Error(FDB_ERR_UNSUPPORTED, "filter output: more than 128 columns");
  hip_check(hipSetDevice(device_), "hipSetDevice");
  std::unique_ptr<DeviceBatch> out(new DeviceBatch());
  out->device = device_;
  // an error below must not hand `out`'s arena (or `in`, which the caller may release) back to the pool with kernels still queued
  struct DrainOnUnwind {
    hipStream_t s; int n = std::uncaught_exceptions();
    ~DrainOnUnwind() { if (std::uncaught_exceptions() > n) (void)hipStreamSynchronize(s); }
  } drain{stream_};
  for (const DevColumn& c : in.cols) {
    if (c.d_values == nullptr && in.rows > 0)
      throw Error(FDB_ERR_UNSUPPORTED, "filter output: column type " + c.format + " (" + c.name + ") is not supported on the device path");
    DevColumn d;
    d.name = c.name; d.format = c.format; d.kind = c.kind; d.dict = c.dict;
    out->cols.push_back(std::move(d));
  }
  *n_selected = 0;
  if (in.rows == 0) return out;
  Resolved R;
  resolve_filter_only(in, &R);
  uint8_t* masks = nullptr;
  uint32_t* offs = nullptr;
  const int64_t total = run_flags(R.args, &masks, &offs);  // (one host round trip: the output is allocated at its exact size)
  *n_selected = total;
  out->rows = total;
  stat_bytes += R.bytes;
  stat_rows += in.rows;
  if (total == 0) { sync(); return out; }
  const uint64_t cap = (uint64_t)total;
  struct Piece { size_t val_off, bit_off; };
  std::vector<Piece> pieces(in.cols.size());
  size_t total_bytes = 0;
  for (size_t k = 0; k < in.cols.size(); k++) {
    const DevColumn& c = in.cols[k];
    const size_t w = c.kind == ColKind::DICT ? 4 : 8;
    pieces[k].val_off = total_bytes;
    total_bytes += align_up(cap * w + kTailPad, 256);
    pieces[k].bit_off = total_bytes;
    if (c.d_validity != nullptr) total_bytes += align_up((cap + 7) / 8 + kTailPad, 256);
  }
  out->arena = device_pool_alloc(device_, std::max<size_t>(total_bytes, 256));
  out->arena_bytes = std::max<size_t>(total_bytes, 256);
  unsigned long long* d_nulls = (unsigned long long*)ctx_->dev_alloc(128 * 64 * 8);  // 64 partial counts per column
  scratch_.push_back(d_nulls);
  hip_check(hipMemsetAsync(d_nulls, 0, in.cols.size() * 64 * 8, stream_), "hipMemsetAsync(null counts)");
  struct OutCol { void* dst; uint8_t* dst_valid; int width; };
  std::vector<OutCol> cols(in.cols.size());
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
  for (size_t k = 0; k < in.cols.size(); k++) {  // one streaming pass per column (fdb_launch_compact_col)
    const DevColumn& c = in.cols[k];
    OutCol& C = cols[k];
    C.dst = (unsigned char*)out->arena + pieces[k].val_off;
    C.width = c.kind == ColKind::DICT ? 4 : 8;
    C.dst_valid = nullptr;
    if (c.d_validity != nullptr) {  // the output bitmap is OR-ed into: zero it first
      C.dst_valid = (uint8_t*)out->arena + pieces[k].bit_off;
      hip_check(hipMemsetAsync(C.dst_valid, 0, align_up((cap + 7) / 8 + kTailPad, 256), stream_), "hipMemsetAsync(validity)");
    }
    hip_check(fdb_launch_compact_col(C.width, c.d_values, c.d_validity, C.dst, C.dst_valid, masks, offs, in.rows, d_nulls + k * 64, device_, stream_), "compact launch");
  }
  if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); pending_events_.emplace_back(e0, e1); }
  last_kernel_ = "compact_col_kernel";
  std::vector<unsigned long long> h_parts(cols.size() * 64), h_nulls(cols.size(), 0);
  hip_check(hipMemcpyAsync(h_parts.data(), d_nulls, cols.size() * 64 * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(null counts)");
  hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  for (size_t k = 0; k < cols.size(); k++) for (int i = 0; i < 64; i++) h_nulls[k] += h_parts[k * 64 + (size_t)i];
  for (size_t k = 0; k < in.cols.size(); k++) {
    const DevColumn& c = in.cols[k];
    DevColumn& d = out->cols[k];
    d.length = total;
    d.null_count = (int64_t)h_nulls[k];
    d.d_values = cols[k].dst;
    d.value_bytes = c.kind == ColKind::BOOL ? (total + 7) / 8 : total * cols[k].width;
    if (c.d_validity != nullptr && d.null_count > 0) {
      d.d_validity = cols[k].dst_valid;
      d.validity_bytes = (total + 7) / 8;
    }
    out->payload_bytes += d.value_bytes + d.validity_bytes;
    // algorithmic bytes of the compaction: every selected value read once and written once, validity likewise
    stat_bytes += 2 * (total * cols[k].width) + (c.d_validity != nullptr ? 2 * ((total + 7) / 8) : 0);
  }
  sync();
  return out;
}

// ≙ PredicateFilter.Callback for every record of a scan at once (filter.go:255-323). Four launches whatever the number of records
// (five above ≈130 M rows): selection bitmap + per-tile counts by a kernel generated for the predicate (fdb_flags_kernel) →
// prefix sums (sel_scan_kernel) → [host: the records' row counts, outputs allocated at their exact sizes] → zero the output
// bitmaps → every column of every record compacted (compact_multi_kernel). Two host round trips in total. Falls back to the
// per-record path (interpreting flags kernel, one compaction launch per column) when the predicate cannot be specialised.
// The one-pass select kernel's workers and its scanner wait for each other with BOUNDED polls (2^22 of them, ≈ 1–2 s): on a GPU shared
// with long kernels of other processes, or with a queue that is preempted or debugged, a valid filter() can run into the bound. That is
// not the caller's error: the batch is filtered again through the three-launch path (flags → prefix sums → compaction), which waits for
// nothing on the device. ($FDB_TEST_SELECT_STALL: pretends the bound was hit — the test of this road)
std::vector<std::unique_ptr<DeviceBatch>> Plan::filter_batches(const DeviceBatch* const* in, int n, int64_t* n_selected) {
  try {
    return filter_batches_impl(in, n, n_selected, /*force_two_pass=*/false);
  } catch (const SelectStall&) {
    return filter_batches_impl(in, n, n_selected, /*force_two_pass=*/true);
  }
}

std::vector<std::unique_ptr<DeviceBatch>> Plan::filter_batches_impl(const DeviceBatch* const* in, int n, int64_t* n_selected, bool force_two_pass) {
  if (filter_root_ < 0) throw Error(FDB_ERR_STATE, "plan has no filter");
  std::vector<std::unique_ptr<DeviceBatch>> out;
  auto per_record = [&]() {
    out.clear();
    for (int i = 0; i < n; i++) out.push_back(filter_batch_interp(*in[i], &n_selected[i]));
    return std::move(out);
  };
  std::vector<int> live;
  for (int i = 0; i < n; i++) {
    if (in[i]->device != device_) throw Error(FDB_ERR_INVALID, "batch lives on a different device than the plan");
    if (in[i]->cols.size() > 128) throw Error(FDB_ERR_UNSUPPORTED, "filter output: more than 128 columns");
    if (in[i]->rows > 0) live.push_back(i);
  }
  // one column layout for the whole launch (parts of one table); anything else takes the per-record path
  bool multi = jit_possible() && !live.empty() && !in[live[0]]->cols.empty();
  for (size_t k = 1; k < live.size() && multi; k++) {
    const DeviceBatch& a = *in[live[0]];
    const DeviceBatch& b = *in[live[k]];
    multi = a.cols.size() == b.cols.size();
    for (size_t c = 0; c < a.cols.size() && multi; c++) multi = a.cols[c].kind == b.cols[c].kind && a.cols[c].name == b.cols[c].name;
  }
  if (!multi) return per_record();
  hip_check(hipSetDevice(device_), "hipSetDevice");
  for (int i : live) in[i]->note_reader(stream_);
  struct DrainOnUnwind {
    hipStream_t s; int n = std::uncaught_exceptions();
    ~DrainOnUnwind() { if (std::uncaught_exceptions() > n) (void)hipStreamSynchronize(s); }
  };
  for (int i = 0; i < n; i++) {
    std::unique_ptr<DeviceBatch> o(new DeviceBatch());
    o->device = device_;
    for (const DevColumn& c : in[i]->cols) {
      if (c.d_values == nullptr && in[i]->rows > 0)
        throw Error(FDB_ERR_UNSUPPORTED, "filter output: column type " + c.format + " (" + c.name + ") is not supported on the device path");
      DevColumn d;
      d.name = c.name; d.format = c.format; d.kind = c.kind; d.dict = c.dict;
      o->cols.push_back(std::move(d));
    }
    out.push_back(std::move(o));
    n_selected[i] = 0;
  }
  DrainOnUnwind drain{stream_};  // (declared after `out`: an error waits for the queued kernels before the arenas go back to the pool)

  // ---- predicate of every record: program + LUTs; records with byte-identical LUT sets share one device copy and one class ----
  const size_t nl = live.size();
  std::vector<Resolved> Rs(nl);
  std::vector<int> lut_class(nl, 0);
  std::vector<size_t> blob_base(nl, 0);
  Blob blob;
  {
    std::vector<size_t> reps;
    for (size_t k = 0; k < nl; k++) {
      Resolved& R = Rs[k];
      std::memset(&R.args, 0, sizeof(R.args));
      R.args.n_rows = in[live[k]]->rows;
      int max_depth = 0;
      R.truths = &truth_cache_;
      R.count_selected = [this, rec = in[live[k]]](int n) { return count_subtree(*rec, n); };
      emit_filter(filter_, filter_root_, *in[live[k]], &R, 0, &max_depth);
      size_t found = reps.size();
      for (size_t r = 0; r < reps.size(); r++) {
        const Resolved& Q = Rs[reps[r]];
        if (Q.blob.bytes == R.blob.bytes && Q.luts.size() == R.luts.size()) { found = r; break; }
      }
      if (found < reps.size()) { lut_class[k] = (int)found; blob_base[k] = blob_base[reps[found]]; }
      else { lut_class[k] = (int)reps.size(); blob_base[k] = blob.add(R.blob.bytes.data(), R.blob.bytes.size()); reps.push_back(k); }
    }
  }
  struct StageScope {
    Context* c;
    explicit StageScope(Context* ctx) : c(ctx) { c->defer_staging(true); }
    ~StageScope() { try { c->defer_staging(false); } catch (...) {} }
  };
  size_t lut_lds_max = 0;
  JitShape shape;
  int row_bytes = 0;
  const FdbScanArgs* d_parts = nullptr;
  const FdbCompactRec* d_recs = nullptr;
  std::vector<FdbScanArgs> parts;
  std::vector<FdbCompactRec> recs;
  int64_t total_tiles = 0, total_super = 0;
  bool fallback = false;
  {
    StageScope stage_scope(ctx_);
    unsigned char* d_blob = blob.bytes.empty() ? nullptr : (unsigned char*)upload(blob.bytes.data(), blob.bytes.size());
    for (size_t k = 0; k < nl; k++) {
      Resolved& R = Rs[k];
      FdbScanArgs& a = R.args;
      size_t lds_off = 0;
      for (const PendingLut& p : R.luts) {
        const bool in_lds = p.len_bytes <= 16384 && lds_off + p.len_bytes <= 32768;
        uint32_t lds = FDB_NO_LDS;
        if (in_lds) { lds = (uint32_t)lds_off; lds_off = align_up(lds_off + p.len_bytes, 16); }
        a.leaves[p.index].lut = d_blob + blob_base[k] + p.blob_off;
        a.leaves[p.index].lut_lds = lds;
      }
      lut_lds_max = std::max(lut_lds_max, align_up(lds_off, 16));
      a.lut_class = lut_class[k];
      // every filter column in the early pools (the flags kernel has no late phase)
      // (no `return per_record()` in here: the scope's destructor — which ends the deferral and ships what was staged — runs only
      // AFTER a return expression has been evaluated, and the per-record path stages LUTs of its own)
      if (assign_slots(*in[live[k]], R, 2, /*relaxed=*/true) == 0) { fallback = true; break; }
      if (k == 0) shape = jit_shape(a, true, 512);
      else if (!jit_shape_merge_args(&shape, Rs[0].args, a, true)) { fallback = true; break; }  // records of different predicate shapes (schema drift)
      // the flags kernel counts in workgroup shares of four tiles (all of one record), the other kernels in tiles
      const int64_t rec_tiles = (a.n_rows + FDB_COMPACT_TILE - 1) / FDB_COMPACT_TILE;
      a.out_tile_base = total_tiles;
      a.tile_begin = total_super;
      total_super += (rec_tiles + 3) / 4;
      a.tile_end = total_super;
      recs.push_back(FdbCompactRec{total_tiles, a.n_rows});
      total_tiles += rec_tiles;
    }
    for (size_t k = 0; k < nl && !fallback; k++) { Rs[k].args.lds_lut_bytes = (uint32_t)lut_lds_max; parts.push_back(Rs[k].args); }
    for (int k = 0; k < shape.n_c4; k++) row_bytes += shape.c4[k].has_values ? 4 : 0;
    for (int k = 0; k < shape.n_c8; k++) row_bytes += shape.c8[k].has_values ? 8 : 0;
    if (!fallback) {
      d_parts = (const FdbScanArgs*)upload(parts.data(), parts.size() * sizeof(FdbScanArgs));
      d_recs = (const FdbCompactRec*)upload(recs.data(), recs.size() * sizeof(FdbCompactRec));
    }
  }
  if (fallback) return per_record();
  const size_t n_cols = in[live[0]]->cols.size();

  // ---- which kernel: one pass over the filter columns (fdb_select_kernel), or bitmap → prefix sums → compaction ----------------------
  // One pass: the wave that evaluates a tile also places it (look-back) and writes the compacted values of the filter columns it has
  // in registers — columns WITHOUT a validity bitmap in any record, ≤ 8 bytes per row together (their tile is staged in LDS), whose
  // slot reads the record's own column (a remapped or widened copy is not the column). Their outputs must exist before the row
  // count does: worst-case sized pool blocks (extra_arenas), repacked into the exact arena when less than 40 % of them is used.
  const bool two_pass_env = force_two_pass || std::getenv("FDB_SELECT_TWO_PASS") != nullptr;  // (A/B, tests, fall-back: the three-launch prefix sum; read per call)
  struct FusedCol { bool wide; int slot; int col; };
  std::vector<FusedCol> fused;
  std::vector<int> fused_of(n_cols, -1);  // column → index in `fused`
  if (!two_pass_env) {
    int budget = 8;
    auto try_slot = [&](bool wide, int slot) {
      const JitSlot& js = wide ? shape.c8[slot] : shape.c4[slot];
      const int w = wide ? 8 : 4;
      if (!js.has_values || js.has_validity != 0 || w > budget || (int)fused.size() >= FDB_SELECT_MAX_FUSED) return;
      int col = -1;
      for (size_t k = 0; k < nl; k++) {
        const DeviceBatch& b = *in[live[k]];
        const void* v = wide ? Rs[k].args.c8[slot].values : Rs[k].args.c4[slot].values;
        int found = -1;
        for (size_t c = 0; c < n_cols; c++) if (b.cols[c].d_values == v && v != nullptr) { found = (int)c; break; }
        if (found < 0 || (k > 0 && found != col)) return;
        const DevColumn& dc = b.cols[(size_t)found];
        if (dc.d_validity != nullptr || dc.kind == ColKind::BOOL || (dc.kind == ColKind::DICT) != !wide) return;
        col = found;
      }
      if (col < 0 || fused_of[(size_t)col] >= 0) return;
      fused_of[(size_t)col] = (int)fused.size();
      fused.push_back(FusedCol{wide, slot, col});
      budget -= w;
    };
    for (int i = 0; i < shape.n_c8; i++) try_slot(true, i);
    for (int i = 0; i < shape.n_c4; i++) try_slot(false, i);
    // (the kernel numbers its outputs 8-byte slots first, then 4-byte slots, in slot order — the order they were tried in)
    for (const FusedCol& f : fused) { if (f.wide) shape.fuse8 |= 1 << f.slot; else shape.fuse4 |= 1 << f.slot; }
  }
  // When it pays (MI355X, 100 M rows, 4 columns): `value > x` 0.86 → 0.79 ms of kernels; with a second, unfused filter column the
  // kernel needs 145 registers (one workgroup per CU) and loses what the saved read gains (0.84 → 0.86), cfg 3's three dictionary
  // leaves 0.94 → 1.00. So: every filter column the predicate reads values of is fused, and they are the full 8 bytes per row.
  // ($FDB_SELECT_ONE_PASS: whenever the kernel can be built — tests, A/B)
  bool one_pass_wanted = !two_pass_env;
  if (one_pass_wanted && std::getenv("FDB_SELECT_ONE_PASS") == nullptr) {
    int fused_bytes = 0, value_slots = 0;
    for (const FusedCol& f : fused) fused_bytes += f.wide ? 8 : 4;
    for (int i = 0; i < shape.n_c8; i++) value_slots += shape.c8[i].has_values ? 1 : 0;
    for (int i = 0; i < shape.n_c4; i++) value_slots += shape.c4[i].has_values ? 1 : 0;
    one_pass_wanted = fused_bytes == 8 && value_slots == (int)fused.size();
  }
  if (!one_pass_wanted) { fused.clear(); std::fill(fused_of.begin(), fused_of.end(), -1); shape.fuse4 = shape.fuse8 = 0; }
  hipFunction_t select_fn = one_pass_wanted ? jit_select_kernel_get(shape) : nullptr;
  hipFunction_t flags_fn = select_fn == nullptr ? jit_flags_get(shape) : nullptr;
  if (select_fn == nullptr) { fused.clear(); std::fill(fused_of.begin(), fused_of.end(), -1); }
  if (select_fn == nullptr && flags_fn == nullptr) return per_record();
  const bool one_pass = select_fn != nullptr;

  // ---- selection bitmap, tile offsets, row counts -----------------------------------------------------------------------------
  uint32_t* d_masks = (uint32_t*)ctx_->dev_alloc((size_t)total_tiles * (FDB_COMPACT_TILE / 8) + 256);
  uint32_t* d_offsets = (uint32_t*)ctx_->dev_alloc((size_t)total_tiles * 4 + 256);
  scratch_.push_back(d_masks); scratch_.push_back(d_offsets);
  auto timed = [&](const std::function<void()>& f) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
    f();
    if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); pending_events_.emplace_back(e0, e1); }
  };
  const size_t stage_off = align_up(lut_lds_max, 16);
  const int first_block = one_pass ? jit_select_block() : 256;
  const size_t first_lds = one_pass ? stage_off + (size_t)(first_block / 64) * jit_select_stage_bytes(shape) : lut_lds_max;
  int per_cu = 1;
  {
    // a wave keeps 4 steps × 64 lanes × 4 rows of every filter column in flight: ≈128 KB per CU (twice jit_select's figure: the
    // waves also spend time on words, counts and stores)
    int waves = row_bytes > 0 ? (131072 / (64 * 4 * 4)) / row_bytes : 8;
    waves = std::max(8, std::min(28, waves));
    per_cu = std::max(1, std::min(jit_blocks_per_cu(one_pass ? select_fn : flags_fn, first_block, first_lds), waves / (first_block / 64)));
    static const int env_per_cu = std::getenv("FDB_FLAGS_BLOCKS_PER_CU") ? std::atoi(std::getenv("FDB_FLAGS_BLOCKS_PER_CU")) : 0;  // (tuning aid)
    if (env_per_cu > 0) per_cu = env_per_cu;
  }
  int64_t grid = (int64_t)(fdb_scan_default_grid(device_) / 2) * per_cu;
  if (grid_override > 0) grid = grid_override;
  if (grid > total_super) grid = total_super;
  unsigned long long* h_base = (unsigned long long*)ctx_->host_alloc((nl + 1 + FDB_SELECT_CTL_WORDS) * 8);
  struct HostFree { Context* c; void* p; ~HostFree() { c->host_free(p); } } hf{ctx_, h_base};
  std::vector<int64_t> totals(nl, 0);
  int launches = 0;
  const unsigned long long* d_rec_base_arg = nullptr;  // (one pass: the offsets are relative to their record already)
  std::vector<char> pre_nullable(n_cols, 0);         // one pass: what the select launch zeroed on its way in (below)
  std::vector<void*> pre_bits(nl, nullptr);
  std::vector<size_t> pre_bits_bytes(nl, 0);
  unsigned long long* pre_nulls = nullptr;
  size_t pre_nulls_words = 0;
  if (one_pass) {
    // outputs of the fused columns: one worst-case block per record, owned by the result from here on
    std::vector<FdbSelectPart> sparts(nl);
    uint32_t epoch = 0;
    unsigned long long ticket_base = 0, arrival_base = 0;
    const size_t status_words = align_up((size_t)total_super, 16);
    unsigned long long* d_ctl = ctx_->select_ctl(align_up(nl, 16) + status_words + 16 * (size_t)total_super, &epoch, &ticket_base, &arrival_base);
    for (size_t k = 0; k < nl; k++) {
      const DeviceBatch& src = *in[live[k]];
      DeviceBatch& o = *out[(size_t)live[k]];
      std::memset(&sparts[k], 0, sizeof(FdbSelectPart));
      sparts[k].total = d_ctl + FDB_SELECT_CTL_WORDS + k;
      if (fused.empty()) continue;
      size_t bytes = 0;
      std::vector<size_t> off(fused.size());
      for (size_t f = 0; f < fused.size(); f++) { off[f] = bytes; bytes += align_up((size_t)src.rows * (fused[f].wide ? 8 : 4) + kTailPad, 256); }
      void* block = device_pool_alloc(device_, bytes);
      o.extra_arenas.push_back(block);
      o.arena_bytes += bytes;
      for (size_t f = 0; f < fused.size(); f++) sparts[k].dst[f] = (unsigned char*)block + off[f];
    }
    // The other columns' validity bitmaps and NULL counters — what the compaction launch ORs / adds into — are zeroed by THIS launch on
    // its way in (FdbSelectArgs::zero) instead of by a launch of their own between the two: they must exist before the row counts do,
    // so a record's bitmaps get a worst-case block of their own (rows / 8 bytes per nullable column; the values still go into an arena
    // of the exact size).
    for (size_t k = 0; k < nl; k++)
      for (size_t c = 0; c < n_cols; c++) if (in[live[k]]->cols[c].d_validity != nullptr) pre_nullable[c] = 1;
    size_t n_rest_pre = 0;
    bool any_nullable_pre = false;
    for (size_t c = 0; c < n_cols; c++) if (fused_of[c] < 0) { n_rest_pre++; any_nullable_pre = any_nullable_pre || pre_nullable[c]; }
    std::vector<unsigned long long> zero_tab;
    if (any_nullable_pre) {
      for (size_t k = 0; k < nl; k++) {
        const DeviceBatch& src = *in[live[k]];
        DeviceBatch& o = *out[(size_t)live[k]];
        size_t bytes = 0;
        for (size_t c = 0; c < n_cols; c++) if (pre_nullable[c]) bytes += align_up((size_t)(src.rows + 7) / 8 + kTailPad, 256);
        void* block = device_pool_alloc(device_, std::max<size_t>(bytes, 256));
        pre_bits[k] = block;
        pre_bits_bytes[k] = std::max<size_t>(bytes, 256);
        // (in FRONT of the fused columns' block: that one stays extra_arenas.back())
        o.extra_arenas.insert(o.extra_arenas.begin(), block);
        o.arena_bytes += pre_bits_bytes[k];
        zero_tab.push_back((unsigned long long)(uintptr_t)block);
        zero_tab.push_back((unsigned long long)pre_bits_bytes[k]);
      }
      pre_nulls_words = nl * std::max<size_t>(n_rest_pre, 1) * 64;
      pre_nulls = (unsigned long long*)ctx_->dev_alloc(align_up(pre_nulls_words * 8, 256));
      scratch_.push_back(pre_nulls);
      zero_tab.push_back((unsigned long long)(uintptr_t)pre_nulls);
      zero_tab.push_back((unsigned long long)align_up(pre_nulls_words * 8, 256));
    }
    FdbSelectArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.ctl = d_ctl;
    sa.ticket_base = ticket_base;
    sa.epoch = epoch;
    sa.stage_off = (uint32_t)stage_off;
    sa.arrival_base = arrival_base;
    sa.status_off = (uint32_t)(FDB_SELECT_CTL_WORDS + align_up(nl, 16));
    sa.place_off = (uint32_t)(sa.status_off + status_words);
    {
      StageScope stage_scope(ctx_);
      sa.sparts = (const FdbSelectPart*)upload(sparts.data(), sparts.size() * sizeof(FdbSelectPart));
      if (!zero_tab.empty()) { sa.zero = (const unsigned long long*)upload(zero_tab.data(), zero_tab.size() * 8); sa.n_zero = (int32_t)(zero_tab.size() / 2); }
    }
    timed([&] { hip_check(jit_select_launch(select_fn, d_parts, (int)parts.size(), total_super, parts[0], (int)grid + 1, first_lds, d_masks, d_offsets, sa, stream_), "select launch"); });
    // (one workgroup more than workers: the scanner; every worker draws exactly one ticket past the end)
    ctx_->select_ctl_drawn((unsigned long long)total_super + (unsigned long long)grid, (unsigned long long)grid + 1);
    launches = 1;
    hip_check(hipMemcpyAsync(h_base, d_ctl + 1, (FDB_SELECT_CTL_WORDS - 1 + nl) * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(row counts)");
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");  // (first host round trip: the other columns' outputs are allocated at their exact sizes)
    if (h_base[0] != 0ull || std::getenv("FDB_TEST_SELECT_STALL") != nullptr) {  // a wait ran into its bound: nothing of this attempt is kept
      ctx_->select_ctl_reset();
      hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
      throw SelectStall();
    }
    for (size_t k = 0; k < nl; k++) totals[k] = (int64_t)h_base[FDB_SELECT_CTL_WORDS - 1 + k];
  } else {
    const int64_t n_blocks = (total_tiles + 1023) / 1024;
    const size_t counts_bytes = align_up((size_t)n_blocks * 8 + (size_t)total_tiles * 4, 256);
    unsigned char* d_counts = (unsigned char*)ctx_->dev_alloc(counts_bytes);
    unsigned long long* d_rec_base = (unsigned long long*)ctx_->dev_alloc((nl + 1) * 8 + 64);
    scratch_.push_back(d_counts); scratch_.push_back(d_rec_base);
    unsigned long long* d_block_sums = (unsigned long long*)d_counts;
    uint32_t* d_tile_counts = (uint32_t*)(d_counts + (size_t)n_blocks * 8);
    const bool two_level = n_blocks > 64;  // (below that every scan workgroup adds up the counts in front of its block itself)
    timed([&] {
      hip_check(jit_flags_launch(flags_fn, d_parts, (int)parts.size(), total_super, parts[0], (int)grid, lut_lds_max, d_masks, d_tile_counts, stream_), "flags launch");
      if (two_level) hip_check(fdb_launch_sel_block_sums(d_tile_counts, total_tiles, d_block_sums, stream_), "block sums launch");
      hip_check(fdb_launch_sel_scan(d_tile_counts, two_level ? d_block_sums : nullptr, total_tiles, d_offsets, d_recs, (int)nl, d_rec_base, stream_), "prefix sums launch");
    });
    launches = two_level ? 3 : 2;
    hip_check(hipMemcpyAsync(h_base, d_rec_base, (nl + 1) * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(row counts)");
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");  // (first host round trip: outputs are allocated at their exact sizes)
    for (size_t k = 0; k < nl; k++) totals[k] = (int64_t)(h_base[k + 1] - h_base[k]);
    d_rec_base_arg = d_rec_base;  // (offsets are global here: compact_multi_kernel subtracts the record's base)
  }

  // ---- outputs: [values of every column | validity bitmaps of every column] per record --------------------------------------
  // (`rest`: the columns compact_multi_kernel writes — all of them, minus what the one-pass kernel has written already)
  std::vector<int> rest;
  for (size_t c = 0; c < n_cols; c++) if (fused_of[c] < 0) rest.push_back((int)c);
  const size_t n_rest = rest.size();
  std::vector<FdbCompactCol> cols(nl * std::max<size_t>(n_rest, 1));
  std::vector<FdbZeroRegion> regions;
  int64_t max_region = 0, any_selected = 0;
  // a column is compacted as nullable if ANY record of the launch has a bitmap for it (the kernel is specialised per column, not
  // per record); the records that have none read an all-ones bitmap
  std::vector<char> nullable(n_cols, 0);
  int64_t ones_rows = 0;
  for (size_t k = 0; k < nl; k++)
    for (size_t c = 0; c < n_cols; c++) if (in[live[k]]->cols[c].d_validity != nullptr) nullable[c] = 1;
  for (size_t k = 0; k < nl; k++)
    for (size_t c = 0; c < n_cols; c++) if (nullable[c] && in[live[k]]->cols[c].d_validity == nullptr) ones_rows = std::max(ones_rows, in[live[k]]->rows);
  uint8_t* d_ones = nullptr;
  if (ones_rows > 0) {
    const size_t ones_bytes = align_up((size_t)(ones_rows + 7) / 8 + kTailPad, 256);
    d_ones = (uint8_t*)ctx_->dev_alloc(ones_bytes);
    scratch_.push_back(d_ones);
    hip_check(hipMemsetAsync(d_ones, 0xFF, ones_bytes, stream_), "hipMemsetAsync(all-valid bitmap)");
  }
  struct Placed { void* values = nullptr; uint8_t* valid = nullptr; };
  std::vector<Placed> placed(nl * n_cols);
  // worst-case blocks whose contents moved into the exact arena (or of records nothing was selected from): back to the pool once the
  // repacking copies have read them — after the wait below, or, when an error unwinds from here on, after a wait of its own
  struct PoolFree {
    int dev; hipStream_t s; std::vector<void*> v; int n = std::uncaught_exceptions();
    ~PoolFree() {
      if (!v.empty() && std::uncaught_exceptions() > n) (void)hipStreamSynchronize(s);
      for (void* p : v) device_pool_free(dev, p);
    }
  } pool_free{device_, stream_, {}};
  std::vector<void*>& repacked = pool_free.v;
  for (size_t k = 0; k < nl; k++) {
    const DeviceBatch& src = *in[live[k]];
    DeviceBatch& o = *out[(size_t)live[k]];
    const int64_t total = totals[k];
    if (total < 0 || total > src.rows) throw Error(FDB_ERR_DEVICE, "internal: selection counts out of range");
    n_selected[live[k]] = total;
    o.rows = total;
    any_selected += total;
    stat_bytes += Rs[k].bytes;
    stat_rows += src.rows;
    // the fused columns stay where the kernel put them unless most of the block is unused
    const bool repack = !fused.empty() && total > 0 && (double)total < 0.4 * (double)src.rows;
    size_t bytes = 0, bits_at = 0;
    std::vector<size_t> val_off(n_cols, 0), bit_off(n_cols, 0);
    for (size_t c = 0; c < n_cols; c++) {
      if (fused_of[c] >= 0 && !repack) continue;
      val_off[c] = bytes;
      bytes += align_up((size_t)total * (src.cols[c].kind == ColKind::DICT ? 4 : 8) + kTailPad, 256);
    }
    bits_at = bytes;
    uint8_t* const bits_block = (uint8_t*)pre_bits[k];  // (one pass with nullable columns: zeroed by the select launch, worst-case sized)
    if (bits_block != nullptr) {
      size_t at = 0;
      for (size_t c = 0; c < n_cols; c++) if (nullable[c]) { bit_off[c] = at; at += align_up((size_t)(src.rows + 7) / 8 + kTailPad, 256); }
    } else {
      for (size_t c = 0; c < n_cols; c++)
        if (nullable[c]) { bit_off[c] = bytes; bytes += align_up(((size_t)total + 7) / 8 + kTailPad, 256); }
    }
    if (total > 0 && bytes > 0) {
      o.arena = device_pool_alloc(device_, std::max<size_t>(bytes, 256));
      o.arena_bytes += std::max<size_t>(bytes, 256);
      if (bytes > bits_at) { regions.push_back(FdbZeroRegion{(unsigned char*)o.arena + bits_at, (int64_t)(bytes - bits_at)}); max_region = std::max<int64_t>(max_region, (int64_t)(bytes - bits_at)); }
    }
    if (bits_block != nullptr && total == 0) {  // nothing selected: the record's result has no buffers at all
      o.extra_arenas.erase(o.extra_arenas.begin());
      o.arena_bytes -= pre_bits_bytes[k];
      repacked.push_back(bits_block);
    }
    if (!fused.empty() && (total == 0 || repack)) {  // the worst-case block is not part of the result
      void* block = o.extra_arenas.back();
      o.extra_arenas.pop_back();
      size_t worst = 0;
      for (size_t f = 0; f < fused.size(); f++) {
        const int w = fused[f].wide ? 8 : 4;
        if (repack) hip_check(hipMemcpyAsync((unsigned char*)o.arena + val_off[(size_t)fused[f].col], (unsigned char*)block + worst, (size_t)total * w, hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync(repack)");
        worst += align_up((size_t)src.rows * w + kTailPad, 256);
      }
      o.arena_bytes -= worst;
      repacked.push_back(block);
    }
    for (size_t c = 0; c < n_cols; c++) {
      Placed& P = placed[k * n_cols + c];
      if (fused_of[c] >= 0 && !repack) {
        // (fused columns sit in the block in `fused` order)
        size_t at = 0;
        for (int f = 0; f < fused_of[c]; f++) at += align_up((size_t)src.rows * (fused[(size_t)f].wide ? 8 : 4) + kTailPad, 256);
        P.values = total > 0 ? (unsigned char*)o.extra_arenas.back() + at : nullptr;
      } else {
        P.values = total > 0 ? (unsigned char*)o.arena + val_off[c] : nullptr;
      }
      P.valid = total > 0 && nullable[c] ? (bits_block != nullptr ? bits_block : (uint8_t*)o.arena) + bit_off[c] : nullptr;
    }
    for (size_t r = 0; r < n_rest; r++) {
      const size_t c = (size_t)rest[r];
      FdbCompactCol& C = cols[k * n_rest + r];
      C.src = src.cols[c].d_values;
      C.width = src.cols[c].kind == ColKind::DICT ? 4 : 8;
      C.nullable = nullable[c];
      C.src_valid = nullable[c] ? (src.cols[c].d_validity != nullptr ? src.cols[c].d_validity : d_ones) : nullptr;
      C.dst = placed[k * n_cols + c].values;
      C.dst_valid = placed[k * n_cols + c].valid;
    }
  }
  std::vector<unsigned long long> h_nulls(nl * std::max<size_t>(n_rest, 1) * 64, 0);
  int any_nullable = 0;
  for (size_t r = 0; r < n_rest; r++) any_nullable |= nullable[(size_t)rest[r]] ? 1 : 0;
  {
    if (any_selected > 0 && n_rest > 0) {
      // NULL counts and validity bitmaps exist only when some column has a bitmap: without one there is nothing to zero, count or copy back
      unsigned long long* d_nulls = nullptr;
      if (any_nullable && pre_nulls != nullptr && pre_nulls_words >= h_nulls.size()) {
        d_nulls = pre_nulls;  // (zeroed by the select launch)
      } else if (any_nullable) {
        d_nulls = (unsigned long long*)ctx_->dev_alloc(h_nulls.size() * 8);
        scratch_.push_back(d_nulls);
        regions.push_back(FdbZeroRegion{d_nulls, (int64_t)(h_nulls.size() * 8)});
        max_region = std::max<int64_t>(max_region, (int64_t)(h_nulls.size() * 8));
      }
      const FdbCompactCol* d_cols = (const FdbCompactCol*)upload(cols.data(), cols.size() * sizeof(FdbCompactCol));
      const FdbZeroRegion* d_regions = regions.empty() ? nullptr : (const FdbZeroRegion*)upload(regions.data(), regions.size() * sizeof(FdbZeroRegion));
      // waves are dealt to the columns in proportion to their bytes per row (a wave stays on its column for the whole launch):
      // 1.5 × the workgroups of 4 waves that are resident at once (a wave's share of tiles is fixed at launch: smaller shares even out
      // the waves that finish late — measured 8 % faster than exactly-resident; handing tiles out dynamically, one ticket per tile or per
      // 8 tiles on a per-column counter, was slower: 2.5 ms and 1.08 ms against 0.90), at least one wave per column, never more waves
      // than a column has tiles
      static const int env_per_cu = std::getenv("FDB_COMPACT_BLOCKS_PER_CU") ? std::atoi(std::getenv("FDB_COMPACT_BLOCKS_PER_CU")) : 0;  // (tuning aid)
      const int64_t budget = (int64_t)(fdb_scan_default_grid(device_) / 2) * (env_per_cu > 0 ? env_per_cu : (fdb_compact_multi_blocks_per_cu(any_nullable) * 3 + 1) / 2) * 4;
      int64_t weight_sum = 0;
      for (size_t r = 0; r < n_rest; r++) weight_sum += cols[r].width;
      std::vector<int32_t> wave_begin(n_rest + 1, 0);
      for (size_t r = 0; r < n_rest; r++) {
        int64_t share = std::max<int64_t>(1, budget * cols[r].width / std::max<int64_t>(weight_sum, 1));
        share = std::min<int64_t>(share, total_tiles);
        wave_begin[r + 1] = wave_begin[r] + (int32_t)share;
      }
      const int32_t* d_wave_begin = (const int32_t*)upload(wave_begin.data(), wave_begin.size() * 4);
      timed([&] {
        if (!regions.empty()) hip_check(fdb_launch_zero_regions(d_regions, (int)regions.size(), max_region, stream_), "zero launch");
        hip_check(fdb_launch_compact_multi(d_recs, (int)nl, d_cols, (int)n_rest, any_nullable, d_wave_begin, wave_begin[n_rest], d_masks, d_offsets, d_rec_base_arg, total_tiles, d_nulls, stream_),
                  "compact launch");
      });
      launches += regions.empty() ? 1 : 2;
      if (any_nullable) hip_check(hipMemcpyAsync(h_nulls.data(), d_nulls, h_nulls.size() * 8, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(null counts)");
    }
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    for (void* p : repacked) device_pool_free(device_, p);
    repacked.clear();
  }
  last_kernel_ = one_pass ? (n_rest > 0 ? "fdb_select_kernel + compact_multi_kernel" : "fdb_select_kernel") : "fdb_flags_kernel + compact_multi_kernel";
  stat_launches += launches;
  for (size_t k = 0; k < nl; k++) {
    const DeviceBatch& src = *in[live[k]];
    DeviceBatch& o = *out[(size_t)live[k]];
    const int64_t total = o.rows;
    for (size_t c = 0; c < n_cols; c++) {
      const Placed& P = placed[k * n_cols + c];
      const int width = src.cols[c].kind == ColKind::DICT ? 4 : 8;
      DevColumn& d = o.cols[c];
      d.length = total;
      unsigned long long nulls = 0;
      size_t r = 0;
      for (; r < n_rest; r++) if ((size_t)rest[r] == c) break;
      if (r < n_rest) for (int q = 0; q < 64; q++) nulls += h_nulls[(k * n_rest + r) * 64 + (size_t)q];
      d.null_count = (int64_t)nulls;
      d.d_values = P.values;
      d.value_bytes = src.cols[c].kind == ColKind::BOOL ? (total + 7) / 8 : total * width;
      if (P.valid != nullptr && d.null_count > 0) { d.d_validity = P.valid; d.validity_bytes = (total + 7) / 8; }
      o.payload_bytes += d.value_bytes + d.validity_bytes;
      // algorithmic bytes of the compaction: every selected value read once and written once, validity likewise
      stat_bytes += 2 * (total * width) + (src.cols[c].d_validity != nullptr ? 2 * ((total + 7) / 8) : 0);
    }
  }
  sync();
  return out;
}

// The record of a resident batch as Arrow in host memory (one device→host copy per buffer).
void export_batch(const DeviceBatch& b, ArrowArray* out, ArrowSchema* out_schema) {
  hip_check(hipSetDevice(b.device), "hipSetDevice");
  const int64_t n = b.rows;
  std::vector<OutColumn> cols;
  for (const DevColumn& c : b.cols) {
    if (c.d_values == nullptr && n > 0)
      throw Error(FDB_ERR_UNSUPPORTED, "export: column type " + c.format + " (" + c.name + ") is not held on the device");
    OutColumn o;
    o.name = c.name;
    o.length = n;
    const int w = c.kind == ColKind::DICT ? 4 : 8;
    o.format = c.kind == ColKind::DICT ? "I" : c.format;
    o.values.resize((size_t)n * w);
    if (n > 0) hip_check(hipMemcpy(o.values.data(), c.d_values, (size_t)n * w, hipMemcpyDeviceToHost), "hipMemcpy(export values)");
    if (c.d_validity != nullptr && n > 0) {
      o.validity.assign((size_t)((n + 63) / 64) * 8, 0);
      hip_check(hipMemcpy(o.validity.data(), c.d_validity, (size_t)(n + 7) / 8, hipMemcpyDeviceToHost), "hipMemcpy(export validity)");
      o.null_count = count_nulls(o.validity.data(), 0, n);
    }
    if (c.kind == ColKind::DICT && c.dict && c.dict->plain) {  // a plain string / binary column leaves as one
      const std::vector<uint8_t> idx_bytes = std::move(o.values);
      set_plain_strings(&o, (const uint32_t*)idx_bytes.data(), o.validity.empty() ? nullptr : o.validity.data(), n, c.dict->values, c.dict->value_format);
    } else if (c.kind == ColKind::DICT) {
      if (!c.dict) throw Error(FDB_ERR_INVALID, "export: dictionary column without its dictionary: " + c.name);
      set_dictionary(&o, c.dict->values, c.dict->value_format);
    } else if (c.kind == ColKind::BOOL) {  // held as int64 1 / 2, Arrow wants bits
      std::vector<uint8_t> bits((size_t)(n + 7) / 8 + 8, 0);
      for (int64_t i = 0; i < n; i++) { int64_t v; std::memcpy(&v, o.values.data() + (size_t)i * 8, 8); if (v >= 2) bits[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7)); }
      o.values = std::move(bits);
    }
    cols.push_back(std::move(o));
  }
  export_record(std::move(cols), n, out, out_schema);
}

void Plan::select(const ArrowArray* array, const ArrowSchema* schema, uint32_t* indices, int64_t capacity, int64_t* n_selected) {
  if (filter_root_ < 0) throw Error(FDB_ERR_STATE, "plan has no filter");
  HostRecordView view;
  view_record(array, schema, &view);
  if (capacity < view.rows) throw Error(FDB_ERR_INVALID, "indices capacity smaller than the record");
  std::function<bool(const std::string&)> want = [this](const std::string& n) {
    for (const ExprNode& e : filter_) if (is_leaf_op(e.op) && e.column == n) return true;
    return false;
  };
  std::unique_ptr<DeviceBatch> b = import_batch(view, device_, &want, stream_);
  *n_selected = 0;
  if (b->rows == 0) return;
  DevBuf d_idx(ctx_, (size_t)b->rows * 4);
  const int64_t n = select_batch(*b, (uint32_t*)d_idx.p, b->rows);
  *n_selected = n;
  if (n) hip_check(hipMemcpy(indices, d_idx.p, (size_t)n * 4, hipMemcpyDeviceToHost), "hipMemcpy(indices)");
}

void Plan::filter(const ArrowArray* array, const ArrowSchema* schema, ArrowArray* out, ArrowSchema* out_schema, int64_t* n_selected) {
  if (filter_root_ < 0) throw Error(FDB_ERR_STATE, "plan has no filter");
  HostRecordView view;
  view_record(array, schema, &view);
  std::unique_ptr<DeviceBatch> b = import_batch(view, device_, nullptr, stream_);
  std::unique_ptr<DeviceBatch> f = filter_batch(*b, n_selected);
  if (*n_selected == 0) return;  // filter.go:264-266
  export_batch(*f, out, out_schema);
}

}  // namespace fdb
