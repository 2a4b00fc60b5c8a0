// fdb_arrow.cpp — see fdb_arrow.h.
#include "fdb_arrow.h"

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string_view>
#include <unordered_set>
#include <memory>
#include <unordered_map>
#include <mutex>

namespace fdb {

namespace {

ColKind kind_of(const std::string& f, bool has_dict) {
  if (has_dict) return ColKind::DICT;
  if (f == "l") return ColKind::I64;
  if (f == "L") return ColKind::U64;
  if (f == "g") return ColKind::F64;
  if (f == "b") return ColKind::BOOL;
  if (f == "u" || f == "z" || f == "U" || f == "Z") return ColKind::STR;
  return ColKind::OTHER;
}

int index_width_of(const std::string& f) {
  if (f.empty()) return 0;
  switch (f[0]) {
    case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': return 4;
    case 'l': case 'L': return 8;
  }
  return 0;
}


// Content hash of a dictionary: the entries' lengths and the entries' bytes laid end to end, both eight bytes at a time. (Until
// round 4 this was FNV-1a, one byte per multiply: a chain that receives 1 024-row records pays the hash of BOTH its dictionaries
// once per record — 17 KB of path names took ≈17 µs of a 33 µs Callback.) The value-by-value and the whole-span entry points give
// the same result: Arrow dictionaries are contiguous (offsets never decrease), so read_dictionary feeds one span.
struct DictHasher {
  uint64_t hl = 0x243F6A8885A308D3ull, hb = 0x13198A2E03707344ull, tail = 0, total = 0;
  int tail_n = 0;
  static uint64_t mix(uint64_t h, uint64_t w) { h = (h ^ w) * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
  void add_len(uint64_t len) { hl = mix(hl, len); }
  void add_bytes(const char* p, size_t n) {
    total += n;
    while (tail_n != 0 && n > 0) { tail |= (uint64_t)(unsigned char)*p++ << (8 * tail_n); n--; if (++tail_n == 8) { hb = mix(hb, tail); tail = 0; tail_n = 0; } }
    for (; n >= 8; n -= 8, p += 8) { uint64_t w; std::memcpy(&w, p, 8); hb = mix(hb, w); }
    for (; n > 0; n--) { tail |= (uint64_t)(unsigned char)*p++ << (8 * tail_n); tail_n++; }
  }
  uint64_t finish() const { uint64_t h = mix(hb, tail ^ ((uint64_t)tail_n << 56)); h = mix(h, total); return mix(h, hl); }
};

// Interning tables of read_dictionary / encode_plain: content hash → weak reference. Candidates are compared OUTSIDE the lock (a
// full-content memcmp under a process-wide mutex would serialise concurrent scan chains), and expired entries are swept whenever
// the table has doubled since the last sweep, so a long-running process that sees an endless stream of distinct dictionaries
// (one per part / row group) keeps the table proportional to the LIVE dictionaries. HostDicts are allocated with `new` (not
// make_shared), so their memory goes when the last strong reference does, whatever weak references remain.
struct InternTable {
  std::mutex mu;
  std::unordered_multimap<uint64_t, std::weak_ptr<HostDict>> live;
  size_t sweep_at = 1024;
  std::vector<std::shared_ptr<HostDict>> candidates(uint64_t h) {
    std::vector<std::shared_ptr<HostDict>> out;
    std::lock_guard<std::mutex> lk(mu);
    auto range = live.equal_range(h);
    for (auto it = range.first; it != range.second;) {
      std::shared_ptr<HostDict> other = it->second.lock();
      if (!other) { it = live.erase(it); continue; }
      out.push_back(std::move(other));
      ++it;
    }
    return out;
  }
  void insert(uint64_t h, const std::shared_ptr<HostDict>& d) {
    std::lock_guard<std::mutex> lk(mu);
    if (live.size() >= sweep_at) {
      for (auto it = live.begin(); it != live.end();) { if (it->second.expired()) it = live.erase(it); else ++it; }
      sweep_at = std::max<size_t>(1024, live.size() * 2);
    }
    live.emplace(h, d);
  }
};

InternTable& dict_table() { static InternTable t; return t; }  // real dictionaries (read_dictionary, make_dictionary)

}  // namespace

// A dictionary from values that did not arrive as an Arrow array (a Parquet dictionary page): same content hash, same interning
// as read_dictionary, so a part decoded from Parquet and one imported from Arrow share their HostDict when the values agree.
std::shared_ptr<HostDict> make_dictionary(std::vector<std::string>&& values, const std::string& value_format) {
  DictHasher hs;
  for (const std::string& v : values) { hs.add_len(v.size()); hs.add_bytes(v.data(), v.size()); }
  const uint64_t h = hs.finish();
  InternTable& table = dict_table();
  for (const std::shared_ptr<HostDict>& other : table.candidates(h))
    if (!other->plain && other->value_format == value_format && other->values == values) return other;
  std::shared_ptr<HostDict> d(new HostDict());
  d->value_format = value_format;
  d->hash = h;
  d->values = std::move(values);
  {
    bool fit = true;
    size_t span = 0;
    for (const std::string& v : d->values) { fit = fit && v.size() <= 0xFFFFFFFFull; span += v.size(); }
    if (fit) {
      d->lens.reserve(d->values.size());
      d->concat.reserve(span);
      for (const std::string& v : d->values) { d->lens.push_back((uint32_t)v.size()); d->concat.append(v); }
    }
  }
  std::unordered_set<std::string_view> seen;
  seen.reserve(d->values.size() * 2);
  for (const std::string& v : d->values)
    if (!seen.insert(std::string_view(v)).second) d->unique = false;
  table.insert(h, d);
  return d;
}

namespace {
// What the C data interface lets a consumer verify without knowing buffer sizes: no negative lengths / offsets / counts, a buffer
// table wherever buffers are announced, child tables and children wherever children are announced, a dictionary array wherever the
// schema has a dictionary. Everything below dereferences these pointers; a producer's bug must come back as an error code.
void check_array_shape(const ArrowArray* a, const ArrowSchema* s, const std::string& what, int depth) {
  if (a == nullptr || s == nullptr) throw Error(FDB_ERR_INVALID, "missing array or schema: " + what);
  if (depth > 8) throw Error(FDB_ERR_UNSUPPORTED, "array nested too deeply: " + what);
  if (a->length < 0 || a->offset < 0) throw Error(FDB_ERR_INVALID, "negative length or offset: " + what);
  if (a->n_buffers < 0 || a->n_children < 0 || s->n_children < 0) throw Error(FDB_ERR_INVALID, "negative buffer or child count: " + what);
  if (a->n_buffers > 0 && a->buffers == nullptr) throw Error(FDB_ERR_INVALID, "buffers announced without a buffer table: " + what);
  if (a->n_children != s->n_children) throw Error(FDB_ERR_INVALID, "schema/array children mismatch: " + what);
  if (a->n_children > 0 && (a->children == nullptr || s->children == nullptr)) throw Error(FDB_ERR_INVALID, "children announced without a child table: " + what);
  for (int64_t i = 0; i < a->n_children; i++) check_array_shape(a->children[i], s->children[i], what, depth + 1);
  if ((s->dictionary != nullptr) != (a->dictionary != nullptr)) throw Error(FDB_ERR_INVALID, "dictionary column without a dictionary array: " + what);
  if (s->dictionary != nullptr) check_array_shape(a->dictionary, s->dictionary, what, depth + 1);
}
}  // namespace

void view_record(const ArrowArray* array, const ArrowSchema* schema, HostRecordView* out) {
  if (array == nullptr || schema == nullptr) throw Error(FDB_ERR_INVALID, "null record");
  check_array_shape(array, schema, "record", 0);
  if (schema->format == nullptr || std::strcmp(schema->format, "+s") != 0)
    throw Error(FDB_ERR_INVALID, "record batch must be exported as a struct array (format \"+s\")");
  if (array->n_children != schema->n_children) throw Error(FDB_ERR_INVALID, "schema/array children mismatch");
  if (array->offset != 0) throw Error(FDB_ERR_INVALID, "sliced struct records are not supported; slice the columns instead");
  out->rows = array->length;
  out->cols.clear();
  out->cols.reserve((size_t)array->n_children);
  for (int64_t i = 0; i < array->n_children; i++) {
    const ArrowSchema* cs = schema->children[i];
    const ArrowArray* ca = array->children[i];
    HostColView c;
    c.name = cs->name ? cs->name : "";
    c.format = cs->format ? cs->format : "";
    c.kind = kind_of(c.format, cs->dictionary != nullptr);
    c.length = ca->length;
    c.offset = ca->offset;
    c.array = ca;
    c.schema = cs;
    if (ca->length != array->length) throw Error(FDB_ERR_INVALID, "column length differs from record length: " + c.name);
    c.validity = ca->n_buffers > 0 ? (const uint8_t*)ca->buffers[0] : nullptr;
    c.null_count = ca->null_count;
    if (c.validity == nullptr) c.null_count = 0;
    else if (c.null_count < 0) c.null_count = count_nulls(c.validity, c.offset, c.length);
    if (c.kind == ColKind::DICT) {
      c.index_width = index_width_of(c.format);
      if (c.index_width == 0) throw Error(FDB_ERR_INVALID, "unsupported dictionary index type " + c.format);
      const std::string df = cs->dictionary->format ? cs->dictionary->format : "";
      if (!(df == "u" || df == "z" || df == "U" || df == "Z")) c.kind = ColKind::OTHER;  // non-string dictionaries
    }
    if (c.kind == ColKind::I64 || c.kind == ColKind::U64 || c.kind == ColKind::F64 || c.kind == ColKind::DICT ||
        c.kind == ColKind::BOOL) {
      if (ca->n_buffers < 2) throw Error(FDB_ERR_INVALID, "missing values buffer: " + c.name);
      c.values = ca->buffers[1];
      if (c.values == nullptr && c.length > 0) throw Error(FDB_ERR_INVALID, "NULL values buffer in a column with rows: " + c.name);
      if (c.kind == ColKind::DICT && (ca->dictionary == nullptr || cs->dictionary == nullptr)) throw Error(FDB_ERR_INVALID, "dictionary column without a dictionary array: " + c.name);
    }
    out->cols.push_back(std::move(c));
  }
}

std::shared_ptr<HostDict> read_dictionary(const HostColView& col) {
  const ArrowArray* da = col.array->dictionary;
  if (da == nullptr || col.schema->dictionary == nullptr || col.schema->dictionary->format == nullptr)
    throw Error(FDB_ERR_INVALID, "dictionary column without a dictionary array: " + col.name);
  const std::string df = col.schema->dictionary->format;
  const std::string value_format = (df == "u" || df == "U") ? "u" : "z";
  const int64_t n = da->length, off = da->offset;
  if (n < 0 || off < 0) throw Error(FDB_ERR_INVALID, "dictionary with a negative length / offset: " + col.name);
  if (n > 0 && (da->n_buffers < 3 || da->buffers == nullptr || da->buffers[1] == nullptr)) throw Error(FDB_ERR_INVALID, "dictionary without an offsets buffer: " + col.name);
  const char* data = da->n_buffers >= 3 ? (const char*)da->buffers[2] : nullptr;
  const bool wide = !(df == "u" || df == "z");
  const void* offsets = da->n_buffers >= 2 ? da->buffers[1] : nullptr;  // (only read when n > 0)
  const int32_t* o32 = (const int32_t*)offsets;
  const int64_t* o64 = (const int64_t*)offsets;
  auto begin = [&](int64_t i) -> int64_t { return wide ? o64[off + i] : (int64_t)o32[off + i]; };
  // A thread's recent dictionaries (see below), asked FIRST and by content alone: a record whose dictionary is one of them — the usual
  // case, record after record of one part — is recognised by two memcmps (its raw offsets, its bytes: ≈ 0.4 µs for 1 024 entries of 12
  // bytes) instead of three passes over the entries and a serial multiply chain over every byte (≈ 4 µs: more than half of what a
  // 1 024-row record's Callback cost). Offsets equal to those of a dictionary that was validated are valid.
  struct Recent { uint64_t hash = 0; std::shared_ptr<std::shared_ptr<HostDict>> holder; std::vector<int32_t> raw_off; };
  static thread_local Recent recent[16];
  static thread_local unsigned recent_next = 0;
  static const bool l1 = std::getenv("FDB_NO_DICT_L1") == nullptr;
  static const bool l1_fast = l1 && std::getenv("FDB_NO_DICT_L1_FAST") == nullptr;  // (A/B aid)
  if (l1_fast && !wide && n > 0)
    for (unsigned k = 0; k < 16; k++) {
      const Recent& r = recent[(recent_next + 15 - k) % 16];  // (most recent first)
      if (!r.holder || r.raw_off.size() != (size_t)n + 1) continue;
      const HostDict& o = **r.holder;
      if (o.plain || o.value_format != value_format || std::memcmp(o32 + off, r.raw_off.data(), ((size_t)n + 1) * 4) != 0) continue;
      const int64_t s0 = r.raw_off[0], sp = (int64_t)r.raw_off[(size_t)n] - s0;
      if ((int64_t)o.concat.size() != sp || (sp > 0 && (data == nullptr || std::memcmp(data + s0, o.concat.data(), (size_t)sp) != 0))) continue;
      return std::shared_ptr<HostDict>(r.holder, r.holder->get());
    }
  // offsets must not decrease, and a dictionary with bytes needs a data buffer (a NULL entry reads as "", like arrow-go's
  // Binary.Value of a null slot)
  for (int64_t i = 0; i < n; i++)
    if (begin(i + 1) < begin(i) || begin(i) < 0) throw Error(FDB_ERR_INVALID, "dictionary with decreasing offsets: " + col.name);
  if (n > 0 && begin(n) > begin(0) && data == nullptr) throw Error(FDB_ERR_INVALID, "dictionary without a data buffer: " + col.name);
  // content hash straight over the Arrow buffers: the lengths, then the bytes of all entries as ONE span
  DictHasher hs;
  for (int64_t i = 0; i < n; i++) hs.add_len((uint64_t)(begin(i + 1) - begin(i)));
  if (n > 0 && begin(n) > begin(0)) hs.add_bytes(data + begin(0), (size_t)(begin(n) - begin(0)));
  const uint64_t h = hs.finish();
  // Dictionaries with identical content are shared: the parts of one table usually carry the same dictionary, and one object
  // for all of them turns every later "same dictionary?" test (key-id LUT cache, LUT de-duplication across the records of a
  // launch) into a pointer compare — and a record whose dictionary is already known costs one pass over its bytes here,
  // no string allocations.
  InternTable& table = dict_table();
  const int64_t span0 = n > 0 ? begin(0) : 0, span = n > 0 ? begin(n) - span0 : 0;
  static thread_local std::vector<uint32_t> lens;
  lens.resize((size_t)n);
  bool lens_fit = true;
  for (int64_t i = 0; i < n; i++) { const int64_t len = begin(i + 1) - begin(i); lens_fit = lens_fit && len <= 0xFFFFFFFFll; lens[(size_t)i] = (uint32_t)len; }
  auto same_content = [&](const HostDict& other) {
    if (other.plain || other.value_format != value_format || (int64_t)other.values.size() != n) return false;
    bool same = lens_fit && other.lens.size() == (size_t)n && (int64_t)other.concat.size() == span;
    if (same && n > 0) same = std::memcmp(other.lens.data(), lens.data(), (size_t)n * 4) == 0 && (span == 0 || std::memcmp(other.concat.data(), data + span0, (size_t)span) == 0);
    return same;
  };
  // A thread's recent dictionaries, in front of the process-wide table. With N chains pushing small records (N goroutines, 1 024-row
  // records: the reference's batch floor) every record of every chain asked the table — one mutex — for the same few HostDicts and took
  // a reference to them: the mutex and, worse, the reference count of the ONE shared object bounced between N cores, ≈ 1.7 µs per record
  // whatever N (0.60 G rows/s at 8 chains, 0.63 at 32: round 4). A hit here touches neither: the returned pointer ALIASES the interned
  // object (same address: "same dictionary?" stays a pointer compare everywhere) but counts its references in a control block that
  // belongs to this thread's cache entry, which in turn holds the interned object alive. ($FDB_NO_DICT_L1: A/B aid)
  if (l1)
    for (Recent& r : recent)
      if (r.holder && r.hash == h && same_content(**r.holder)) return std::shared_ptr<HostDict>(r.holder, r.holder->get());
  auto remember = [&](const std::shared_ptr<HostDict>& d) -> std::shared_ptr<HostDict> {
    // (bounded: an entry keeps its dictionary alive for the thread's lifetime, so only small ones are cached — ≤ 16 × 256 KiB per pushing
    // thread; with a bigger dictionary the pass over its bytes dwarfs the mutex and the reference count this cache exists to avoid)
    if (!l1 || span + n * 4 > (int64_t)(256 << 10)) return d;
    Recent& r = recent[recent_next++ % 16];
    r.hash = h;
    r.holder = std::make_shared<std::shared_ptr<HostDict>>(d);
    r.raw_off.clear();
    if (!wide && n > 0 && lens_fit) r.raw_off.assign(o32 + off, o32 + off + n + 1);  // (what the content-only lookup above compares)
    return std::shared_ptr<HostDict>(r.holder, d.get());
  };
  for (const std::shared_ptr<HostDict>& other : table.candidates(h))
    if (same_content(*other)) return remember(other);
  std::shared_ptr<HostDict> d(new HostDict());
  d->value_format = value_format;
  d->hash = h;
  d->values.resize((size_t)n);
  for (int64_t i = 0; i < n; i++) d->values[(size_t)i].assign(data + begin(i), (size_t)(begin(i + 1) - begin(i)));
  if (lens_fit) { d->lens = lens; if (span > 0) d->concat.assign(data + span0, (size_t)span); }
  std::unordered_set<std::string_view> seen;
  seen.reserve(d->values.size() * 2);
  for (const std::string& v : d->values)
    if (!seen.insert(std::string_view(v)).second) d->unique = false;
  table.insert(h, d);
  return remember(d);
}

std::shared_ptr<HostDict> encode_plain(const HostColView& col, std::vector<uint32_t>* idx) {
  const ArrowArray* a = col.array;
  const bool wide = col.format == "U" || col.format == "Z";
  const int64_t n = col.length, off = col.offset;
  if (n > 0 && (a->n_buffers < 3 || a->buffers == nullptr || a->buffers[1] == nullptr)) throw Error(FDB_ERR_INVALID, "string column without offsets: " + col.name);
  const char* data = a->n_buffers >= 3 ? (const char*)a->buffers[2] : nullptr;
  const void* offsets = a->n_buffers >= 2 ? a->buffers[1] : nullptr;  // (only read when n > 0)
  const int32_t* o32 = (const int32_t*)offsets;
  const int64_t* o64 = (const int64_t*)offsets;
  auto begin = [&](int64_t i) -> int64_t { return wide ? o64[off + i] : (int64_t)o32[off + i]; };
  std::shared_ptr<HostDict> d(new HostDict());
  d->value_format = col.format;  // the column's own type, large or not: key columns and filter output keep it
  d->plain = true;
  idx->assign((size_t)n, 0u);
  // views into the record's own bytes while encoding; the distinct values are copied once at the end
  std::unordered_map<std::string_view, uint32_t> ids;
  std::vector<std::string_view> order;
  for (int64_t i = 0; i < n; i++) {
    if (col.null_count > 0 && col.validity != nullptr && !((col.validity[(off + i) >> 3] >> ((off + i) & 7)) & 1)) continue;
    const int64_t b0 = begin(i), b1 = begin(i + 1);
    if (b1 < b0 || b0 < 0) throw Error(FDB_ERR_INVALID, "string column with decreasing offsets: " + col.name);
    if (b1 > b0 && data == nullptr) throw Error(FDB_ERR_INVALID, "string column without a data buffer: " + col.name);
    const std::string_view v(b1 > b0 ? data + b0 : "", (size_t)(b1 - b0));
    auto it = ids.find(v);
    if (it == ids.end()) {
      if (order.size() >= 0xFFFFFFFEull) throw Error(FDB_ERR_UNSUPPORTED, "more than 2^32 distinct values in string column " + col.name);
      it = ids.emplace(v, (uint32_t)order.size()).first;
      order.push_back(v);
    }
    (*idx)[(size_t)i] = it->second;
  }
  DictHasher hs;
  for (const std::string_view& v : order) { hs.add_len(v.size()); hs.add_bytes(v.data(), v.size()); }
  d->hash = hs.finish() ^ 0x9E3779B97F4A7C15ull;  // (never equal to the hash of a real dictionary with the same entries)
  // share with an earlier record's encoding when the distinct values came out the same (same order): downstream caches
  // (key-id LUTs, truth tables) are keyed by the dictionary object
  static InternTable table;
  for (const std::shared_ptr<HostDict>& other : table.candidates(d->hash)) {
    bool same = other->plain && other->value_format == d->value_format && other->values.size() == order.size();
    for (size_t i = 0; i < order.size() && same; i++) same = std::string_view(other->values[i]) == order[i];
    if (same) return other;
  }
  d->values.reserve(order.size());
  for (const std::string_view& v : order) d->values.emplace_back(v);
  table.insert(d->hash, d);
  return d;
}

const std::vector<int32_t>& HostDict::arrow_offsets() const {
  std::call_once(offsets_once_, [this] {
    arrow_offsets_.resize(values.size() + 1);
    uint64_t off = 0;
    for (size_t i = 0; i < values.size(); i++) { arrow_offsets_[i] = (int32_t)off; off += values[i].size(); }
    arrow_offsets_[values.size()] = (int32_t)off;
  });
  return arrow_offsets_;
}

const std::vector<uint32_t>& HostDict::sorted_ranks() const {
  std::call_once(ranks_once_, [this] {
    std::vector<uint32_t> order(values.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    if (!std::is_sorted(values.begin(), values.end())) std::stable_sort(order.begin(), order.end(), [this](uint32_t x, uint32_t y) { return values[x] < values[y]; });
    sorted_ranks_.resize(values.size());
    for (size_t r = 0; r < order.size(); r++) sorted_ranks_[order[r]] = (uint32_t)r;
  });
  return sorted_ranks_;
}

template <typename S>
void set_dictionary(OutColumn* oc, const std::vector<S>& values, const std::string& value_format) {
  const bool large = value_format == "U" || value_format == "Z";
  uint64_t total = 0;
  for (const S& v : values) total += v.size();
  if (!large && total > 0x7FFFFFFFull) throw Error(FDB_ERR_UNSUPPORTED, "group column " + oc->name + ": more than 2 GiB of distinct key bytes in one dictionary");
  oc->is_dict = true;
  oc->dict_format = value_format;
  oc->dict_offsets.clear();
  oc->dict_offsets64.clear();
  if (large) oc->dict_offsets64.resize(values.size() + 1); else oc->dict_offsets.resize(values.size() + 1);
  oc->dict_data.clear();
  oc->dict_data.reserve((size_t)total);
  uint64_t off = 0;
  for (size_t v = 0; v <= values.size(); v++) {
    if (large) oc->dict_offsets64[v] = (int64_t)off; else oc->dict_offsets[v] = (int32_t)off;
    if (v == values.size()) break;
    oc->dict_data.insert(oc->dict_data.end(), values[v].begin(), values[v].end());
    off += values[v].size();
  }
}

template <typename S>
void set_plain_strings(OutColumn* oc, const uint32_t* idx, const uint8_t* valid_bits, int64_t n, const std::vector<S>& values,
                       const std::string& value_format) {
  auto valid = [&](int64_t i) { return valid_bits == nullptr || ((valid_bits[i >> 3] >> (i & 7)) & 1); };  // (no bitmap: no NULLs)
  uint64_t total = 0;
  for (int64_t i = 0; i < n; i++) if (valid(i)) total += values[idx[i]].size();
  // the reference starts a new record when a key builder passes 2 GiB (aggregate.go:426-468); one record here, 64-bit offsets
  const bool utf8 = value_format == "u" || value_format == "U";
  const bool large = total > 0x7FFFFFFFull || value_format == "U" || value_format == "Z";
  std::vector<uint8_t> offsets((size_t)(n + 1) * (large ? 8 : 4));
  std::vector<char> data;
  data.reserve((size_t)total);
  uint64_t off = 0;
  for (int64_t i = 0; i <= n; i++) {
    if (large) { const int64_t o = (int64_t)off; std::memcpy(offsets.data() + (size_t)i * 8, &o, 8); }
    else { const int32_t o = (int32_t)off; std::memcpy(offsets.data() + (size_t)i * 4, &o, 4); }
    if (i < n && valid(i)) { const S& v = values[idx[i]]; data.insert(data.end(), v.begin(), v.end()); off += v.size(); }
  }
  oc->is_dict = false;
  oc->is_str = true;
  oc->format = utf8 ? (large ? "U" : "u") : (large ? "Z" : "z");
  oc->values = std::move(offsets);
  oc->ext_values = nullptr;
  oc->str_data = std::move(data);
}

template void set_dictionary<std::string>(OutColumn*, const std::vector<std::string>&, const std::string&);
template void set_dictionary<std::string_view>(OutColumn*, const std::vector<std::string_view>&, const std::string&);
template void set_plain_strings<std::string>(OutColumn*, const uint32_t*, const uint8_t*, int64_t, const std::vector<std::string>&, const std::string&);
template void set_plain_strings<std::string_view>(OutColumn*, const uint32_t*, const uint8_t*, int64_t, const std::vector<std::string_view>&, const std::string&);

int64_t count_nulls(const uint8_t* validity, int64_t offset, int64_t length) {
  int64_t set = 0;
  for (int64_t i = 0; i < length; i++) set += (validity[(offset + i) >> 3] >> ((offset + i) & 7)) & 1;
  return length - set;
}

void copy_bits(const uint8_t* src, int64_t offset, int64_t length, uint8_t* dst) {
  const int64_t nbytes = (length + 7) / 8;
  if (nbytes == 0) return;
  if ((offset & 7) == 0) {
    std::memcpy(dst, src + (offset >> 3), (size_t)nbytes);
  } else {
    const int sh = (int)(offset & 7);
    const uint8_t* s = src + (offset >> 3);
    const int64_t src_bytes = (offset + length + 7) / 8 - (offset >> 3);
    for (int64_t i = 0; i < nbytes; i++) {
      uint32_t lo = s[i];
      uint32_t hi = (i + 1 < src_bytes) ? s[i + 1] : 0;
      dst[i] = (uint8_t)((lo >> sh) | (hi << (8 - sh)));
    }
  }
  const int tail = (int)(length & 7);
  if (tail) dst[nbytes - 1] &= (uint8_t)((1u << tail) - 1u);
}

namespace {
inline bool large_offsets(const OutColumn& c) { return c.format == "U" || c.format == "Z"; }
inline int64_t str_offset(const OutColumn& c, int64_t i) {
  if (large_offsets(c)) { int64_t o; std::memcpy(&o, c.values.data() + (size_t)i * 8, 8); return o; }
  int32_t o; std::memcpy(&o, c.values.data() + (size_t)i * 4, 4); return o;
}
}  // namespace

int64_t plain_string_bytes(const OutColumn& c, int64_t i) { return str_offset(c, i + 1) - str_offset(c, i); }
int64_t plain_string_total(const OutColumn& c, int64_t n) { return str_offset(c, n) - str_offset(c, 0); }

std::vector<OutColumn> slice_columns(const std::vector<OutColumn>& cols, int64_t start, int64_t len) {
  std::vector<OutColumn> out;
  out.reserve(cols.size());
  for (const OutColumn& c : cols) {
    OutColumn o;
    o.name = c.name; o.format = c.format; o.length = len;
    const uint8_t* vbits = c.ext_validity != nullptr ? c.ext_validity : (c.validity.empty() ? nullptr : c.validity.data());
    if (vbits != nullptr && c.null_count > 0 && len > 0) {
      o.validity.assign((size_t)(len + 7) / 8, 0);
      copy_bits(vbits, start, len, o.validity.data());
      o.null_count = count_nulls(o.validity.data(), 0, len);
      if (o.null_count == 0) o.validity.clear();
    }
    const uint8_t* vals = c.ext_values != nullptr ? c.ext_values : c.values.data();
    if (c.is_str) {
      const int64_t b0 = str_offset(c, start), b1 = str_offset(c, start + len);
      const bool utf8 = c.format == "u" || c.format == "U";
      const bool large = b1 - b0 > 0x7FFFFFFFll;
      o.values.assign((size_t)(len + 1) * (large ? 8 : 4), 0);
      for (int64_t i = 0; i <= len; i++) {
        const int64_t v = str_offset(c, start + i) - b0;
        if (large) std::memcpy(o.values.data() + (size_t)i * 8, &v, 8);
        else { const int32_t v32 = (int32_t)v; std::memcpy(o.values.data() + (size_t)i * 4, &v32, 4); }
      }
      o.is_str = true;
      o.format = utf8 ? (large ? "U" : "u") : (large ? "Z" : "z");
      o.str_data.assign(c.str_data.begin() + (ptrdiff_t)b0, c.str_data.begin() + (ptrdiff_t)b1);
    } else if (c.format == "b") {
      o.values.assign((size_t)(len + 7) / 8 + 8, 0);
      if (len > 0) copy_bits(vals, start, len, o.values.data());
    } else {
      const size_t w = c.format == "I" ? 4 : 8;
      if (len > 0) o.values.assign(vals + (size_t)start * w, vals + (size_t)(start + len) * w);
    }
    if (c.is_dict) {
      o.is_dict = true; o.dict_format = c.dict_format;
      o.dict_offsets = c.dict_offsets; o.dict_offsets64 = c.dict_offsets64; o.dict_data = c.dict_data; o.dict_ref = c.dict_ref;
    }
    out.push_back(std::move(o));
  }
  return out;
}

// ---- export --------------------------------------------------------------------------------------------

namespace {

struct Holder {
  std::vector<OutColumn> cols;
  // C structs handed out
  std::vector<ArrowArray> child_arrays;
  std::vector<ArrowArray*> child_array_ptrs;
  std::vector<ArrowArray> dict_arrays;
  std::vector<std::vector<const void*>> buffers;  // per child + per dict + top
  std::vector<ArrowSchema> child_schemas;
  std::vector<ArrowSchema*> child_schema_ptrs;
  std::vector<ArrowSchema> dict_schemas;
  int refs = 2;  // array + schema
};

void noop_release_array(ArrowArray* a) { a->release = nullptr; }
void noop_release_schema(ArrowSchema* s) { s->release = nullptr; }

void drop(Holder* h) {
  if (--h->refs == 0) delete h;
}

void release_top_array(ArrowArray* a) {
  Holder* h = (Holder*)a->private_data;
  for (auto& c : h->child_arrays) {
    if (c.release) c.release(&c);
  }
  a->release = nullptr;
  drop(h);
}

void release_top_schema(ArrowSchema* s) {
  Holder* h = (Holder*)s->private_data;
  for (auto& c : h->child_schemas) {
    if (c.release) c.release(&c);
  }
  s->release = nullptr;
  drop(h);
}

}  // namespace

void export_record(std::vector<OutColumn>&& cols_in, int64_t rows, ArrowArray* out, ArrowSchema* out_schema) {
  Holder* h = new Holder();
  h->cols = std::move(cols_in);
  const size_t n = h->cols.size();
  h->child_arrays.resize(n);
  h->child_array_ptrs.resize(n);
  h->dict_arrays.resize(n);
  h->child_schemas.resize(n);
  h->child_schema_ptrs.resize(n);
  h->dict_schemas.resize(n);
  h->buffers.resize(2 * n + 1);
  for (size_t i = 0; i < n; i++) {
    OutColumn& c = h->cols[i];
    ArrowArray& a = h->child_arrays[i];
    std::memset(&a, 0, sizeof(a));
    a.length = c.length;
    a.null_count = c.null_count;
    a.offset = 0;
    a.n_buffers = 2;
    std::vector<const void*>& b = h->buffers[i];
    b.resize(2);
    static const uint64_t kEmpty = 0;
    // each buffer is either external (inside the record's shared block) or the column's own vector
    const void* vbits = c.ext_validity != nullptr ? (const void*)c.ext_validity : (c.validity.empty() ? nullptr : (const void*)c.validity.data());
    b[0] = c.null_count > 0 ? vbits : nullptr;
    b[1] = c.ext_values != nullptr ? (const void*)c.ext_values : (c.values.empty() ? (const void*)&kEmpty : (const void*)c.values.data());
    if (c.is_str) {
      static const char kNoBytes = 0;
      a.n_buffers = 3;
      b.push_back(c.str_data.empty() ? (const void*)&kNoBytes : (const void*)c.str_data.data());
    }
    a.buffers = b.data();
    a.release = noop_release_array;
    ArrowSchema& s = h->child_schemas[i];
    std::memset(&s, 0, sizeof(s));
    s.format = c.format.c_str();
    s.name = c.name.c_str();
    s.flags = ARROW_FLAG_NULLABLE;
    s.release = noop_release_schema;
    if (c.is_dict) {
      ArrowArray& d = h->dict_arrays[i];
      std::memset(&d, 0, sizeof(d));
      const bool wide = !c.dict_offsets64.empty();
      d.length = c.dict_entries();
      d.null_count = 0;
      d.n_buffers = 3;
      std::vector<const void*>& db = h->buffers[n + i];
      db.resize(3);
      db[0] = nullptr;
      static const char kNoData = 0;
      if (c.dict_ref) {  // (the interned dictionary's own buffers: the holder keeps the reference until the consumer releases the record)
        db[1] = (const void*)c.dict_ref->arrow_offsets().data();
        db[2] = c.dict_ref->concat.empty() ? (const void*)&kNoData : (const void*)c.dict_ref->concat.data();
      } else {
        db[1] = wide ? (const void*)c.dict_offsets64.data() : (const void*)c.dict_offsets.data();
        db[2] = c.dict_data.empty() ? (const void*)&kNoData : (const void*)c.dict_data.data();
      }
      d.buffers = db.data();
      d.release = noop_release_array;
      a.dictionary = &d;
      ArrowSchema& ds = h->dict_schemas[i];
      std::memset(&ds, 0, sizeof(ds));
      ds.format = c.dict_format.c_str();
      ds.name = "";
      ds.flags = ARROW_FLAG_NULLABLE;
      ds.release = noop_release_schema;
      s.dictionary = &ds;
    }
    h->child_array_ptrs[i] = &a;
    h->child_schema_ptrs[i] = &s;
  }
  std::memset(out, 0, sizeof(*out));
  out->length = rows;
  out->null_count = 0;
  out->n_buffers = 1;
  std::vector<const void*>& tb = h->buffers[2 * n];
  tb.assign(1, nullptr);
  out->buffers = tb.data();
  out->n_children = (int64_t)n;
  out->children = h->child_array_ptrs.data();
  out->release = release_top_array;
  out->private_data = h;
  std::memset(out_schema, 0, sizeof(*out_schema));
  out_schema->format = "+s";
  out_schema->name = "";
  out_schema->n_children = (int64_t)n;
  out_schema->children = h->child_schema_ptrs.data();
  out_schema->release = release_top_schema;
  out_schema->private_data = h;
}

void roundtrip_record(const HostRecordView& view, ArrowArray* out, ArrowSchema* out_schema) {
  std::vector<OutColumn> cols;
  for (const HostColView& c : view.cols) {
    OutColumn o;
    o.name = c.name;
    o.length = c.length;
    o.null_count = c.null_count;
    if (c.null_count > 0) {
      o.validity.assign((size_t)(c.length + 7) / 8 + 8, 0);
      copy_bits(c.validity, c.offset, c.length, o.validity.data());
    }
    switch (c.kind) {
      case ColKind::I64: case ColKind::U64: case ColKind::F64:
        o.format = c.format;
        o.values.resize((size_t)c.length * 8);
        if (c.length > 0) std::memcpy(o.values.data(), (const unsigned char*)c.values + (size_t)c.offset * 8, (size_t)c.length * 8);
        break;
      case ColKind::BOOL:
        o.format = "b";
        o.values.assign((size_t)(c.length + 7) / 8 + 8, 0);
        if (c.length > 0) copy_bits((const uint8_t*)c.values, c.offset, c.length, o.values.data());
        break;
      case ColKind::DICT: {
        const std::shared_ptr<HostDict> d = read_dictionary(c);
        o.format = "I";
        o.values.resize((size_t)c.length * 4);
        uint32_t* idx = (uint32_t*)o.values.data();
        for (int64_t i = 0; i < c.length; i++) {
          switch (c.index_width) {
            case 1: idx[i] = ((const uint8_t*)c.values)[c.offset + i]; break;
            case 2: idx[i] = ((const uint16_t*)c.values)[c.offset + i]; break;
            case 4: idx[i] = ((const uint32_t*)c.values)[c.offset + i]; break;
            default: idx[i] = (uint32_t)((const uint64_t*)c.values)[c.offset + i]; break;
          }
        }
        set_dictionary(&o, d->values, d->value_format);
        break;
      }
      case ColKind::STR: {
        std::vector<uint32_t> idx;
        const std::shared_ptr<HostDict> d = encode_plain(c, &idx);
        set_plain_strings(&o, idx.data(), o.validity.empty() ? nullptr : o.validity.data(), c.length, d->values, d->value_format);
        break;
      }
      default:
        throw Error(FDB_ERR_UNSUPPORTED, "column type " + c.format + " (" + c.name + ") is not supported");
    }
    cols.push_back(std::move(o));
  }
  export_record(std::move(cols), view.rows, out, out_schema);
}

}  // namespace fdb
