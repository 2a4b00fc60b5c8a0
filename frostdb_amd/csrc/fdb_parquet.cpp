// fdb_parquet.cpp — Parquet column chunks decoded straight into HBM-resident columns.
//
// ≙ pqarrow/arrow.go:711-823 (writeColumnToArray) + pqarrow/writer/writer.go:391-405 + parquet-go's page decoders: today the
// reference decodes every page on the CPU and appends dictionary values PER ROW into an Arrow builder; the survey calls that
// producer the real end-to-end bottleneck. Here the host only reads what is O(pages) or O(runs): thrift page headers, the
// dictionary page (strings never reach the device), and the HEADERS of the RLE / bit-packed hybrid runs (definition levels,
// dictionary indices). The page bytes go to HBM as they are, and the per-row work — definition levels → validity bitmap, value
// rank of every row, dictionary index / PLAIN value of every row — runs in fdb_kernels.hip (pq_* kernels). The result is a
// DeviceBatch with exactly the layout import_batch produces, so every plan entry point takes it.
//
// Covered (what pqarrow/convert/convert.go:28-102 maps to Arrow — UTF8 / BYTE_ARRAY, Int(64) signed and unsigned, BOOLEAN, DOUBLE — in the
// encodings and codecs a FrostDB schema can ask for, schema.proto:54-86): BOOLEAN PLAIN / RLE; INT64 PLAIN / DELTA_BINARY_PACKED; DOUBLE PLAIN;
// BYTE_ARRAY dictionary-encoded, PLAIN, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY; UNCOMPRESSED, SNAPPY, GZIP, BROTLI, ZSTD, LZ4(_RAW).
// Repeated (list) columns are refused. First slice, as it was built (what FrostDB's default layouts produce, dynparquet/schema.go:508-560): flat schemas; INT64 / DOUBLE columns with
// PLAIN data pages; BYTE_ARRAY columns with a PLAIN dictionary page + RLE_DICTIONARY data pages (→ dictionary<uint32, binary>,
// pqarrow/convert/convert.go:64-70); required or optional (max definition level 1); data pages V1 and V2; INT64 also DELTA_BINARY_PACKED (the reference's default for struct-tag schemas, internal/records/record_builder.go:146-148); pages UNCOMPRESSED, or SNAPPY / GZIP / ZSTD / LZ4_RAW (inflated on the host while the page headers are walked).
// Anything else (DELTA_* encodings, compressed pages, dictionary fallback to PLAIN, nested columns) is FDB_ERR_UNSUPPORTED.
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <list>
#include <memory>
#include <tuple>
#include <type_traits>
#include <mutex>
#include <functional>
#include <deque>
#include <condition_variable>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <cstdlib>
#include <cstring>

#include "fdb_hostpool.h"
#include "fdb_context.h"
#include "fdb_kernels.h"
#include "fdb_plan_internal.h"

namespace fdb {

namespace {

// ---- thrift compact protocol, as much as a PageHeader needs (parquet-format/src/main/thrift/parquet.thrift: PageHeader) ---------
struct Thrift {
  const uint8_t* p;
  const uint8_t* end;
  void need(size_t n) const { if ((size_t)(end - p) < n) throw Error(FDB_ERR_INVALID, "parquet: page header truncated"); }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      need(1);
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return v;
    }
    throw Error(FDB_ERR_INVALID, "parquet: varint too long");
  }
  int64_t zigzag() { const uint64_t v = varint(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
  // `in_collection`: a bool is one byte as a list / set / map element (as a struct field its value lives in the field header).
  // Every collection element occupies at least one byte, so a count larger than what is left of the header is a damaged header —
  // without that check a list of 2^60 zero-byte elements keeps a thread busy for ever.
  void skip(int type, int depth = 0, bool in_collection = false) {
    if (depth > 16) throw Error(FDB_ERR_INVALID, "parquet: page header nested too deeply");
    switch (type) {
      case 1: case 2: if (in_collection) { need(1); p += 1; } return;
      case 3: need(1); p += 1; return;              // byte
      case 4: case 5: case 6: (void)zigzag(); return;
      case 7: need(8); p += 8; return;              // double
      case 8: { const uint64_t n = varint(); need((size_t)std::min<uint64_t>(n, (uint64_t)1 << 40)); p += n; return; }  // binary
      case 9: case 10: {                            // list / set
        need(1);
        const uint8_t h = *p++;
        uint64_t n = h >> 4;
        if (n == 15) n = varint();
        if (n > (uint64_t)(end - p)) throw Error(FDB_ERR_INVALID, "parquet: page header truncated");
        for (uint64_t i = 0; i < n; i++) skip(h & 0x0F, depth + 1, true);
        return;
      }
      case 11: {                                    // map
        const uint64_t n = varint();
        if (n == 0) return;
        need(1);
        const uint8_t kv = *p++;
        if (n > (uint64_t)(end - p) / 2) throw Error(FDB_ERR_INVALID, "parquet: page header truncated");
        for (uint64_t i = 0; i < n; i++) { skip(kv >> 4, depth + 1, true); skip(kv & 0x0F, depth + 1, true); }
        return;
      }
      case 12: skip_struct(depth + 1); return;
      default: throw Error(FDB_ERR_INVALID, "parquet: unknown thrift type in a page header");
    }
  }
  // Calls f(field id, type) for every field; f returns true if it consumed the value.
  template <typename F>
  void read_struct(F&& f, int depth = 0) {
    int16_t last = 0;
    for (;;) {
      need(1);
      const uint8_t h = *p++;
      if (h == 0) return;
      const int type = h & 0x0F;
      int16_t id;
      if ((h >> 4) != 0) id = (int16_t)(last + (h >> 4)); else id = (int16_t)zigzag();
      last = id;
      if (!f(id, type)) skip(type, depth);
    }
  }
  void skip_struct(int depth) { read_struct([](int16_t, int) { return false; }, depth); }
};

enum { PQ_DATA_PAGE = 0, PQ_DICTIONARY_PAGE = 2, PQ_DATA_PAGE_V2 = 3 };
enum { ENC_PLAIN = 0, ENC_PLAIN_DICTIONARY = 2, ENC_RLE = 3, ENC_DELTA_BINARY_PACKED = 5, ENC_DELTA_LENGTH_BYTE_ARRAY = 6, ENC_DELTA_BYTE_ARRAY = 7, ENC_RLE_DICTIONARY = 8 };
enum { PT_BOOLEAN = 0, PT_INT64 = 2, PT_DOUBLE = 5, PT_BYTE_ARRAY = 6 };  // parquet.thrift Type

struct PageHeader {
  int32_t type = -1, uncompressed = 0, compressed = 0;
  int32_t num_values = 0, encoding = -1, def_encoding = ENC_RLE;
  int32_t v2_num_nulls = 0, v2_def_bytes = 0, v2_rep_bytes = 0;
  bool v2_compressed = true;
};

PageHeader read_page_header(Thrift& t) {
  PageHeader h;
  t.read_struct([&](int16_t id, int type) {
    if (id == 1 && type == 5) { h.type = (int32_t)t.zigzag(); return true; }
    if (id == 2 && type == 5) { h.uncompressed = (int32_t)t.zigzag(); return true; }
    if (id == 3 && type == 5) { h.compressed = (int32_t)t.zigzag(); return true; }
    if (id == 5 && type == 12) {  // DataPageHeader
      t.read_struct([&](int16_t f, int ty) {
        if (f == 1 && ty == 5) { h.num_values = (int32_t)t.zigzag(); return true; }
        if (f == 2 && ty == 5) { h.encoding = (int32_t)t.zigzag(); return true; }
        if (f == 3 && ty == 5) { h.def_encoding = (int32_t)t.zigzag(); return true; }
        return false;
      }, 1);
      return true;
    }
    if (id == 7 && type == 12) {  // DictionaryPageHeader
      t.read_struct([&](int16_t f, int ty) {
        if (f == 1 && ty == 5) { h.num_values = (int32_t)t.zigzag(); return true; }
        if (f == 2 && ty == 5) { h.encoding = (int32_t)t.zigzag(); return true; }
        return false;
      }, 1);
      return true;
    }
    if (id == 8 && type == 12) {  // DataPageHeaderV2
      t.read_struct([&](int16_t f, int ty) {
        if (f == 1 && ty == 5) { h.num_values = (int32_t)t.zigzag(); return true; }
        if (f == 2 && ty == 5) { h.v2_num_nulls = (int32_t)t.zigzag(); return true; }
        if (f == 4 && ty == 5) { h.encoding = (int32_t)t.zigzag(); return true; }
        if (f == 5 && ty == 5) { h.v2_def_bytes = (int32_t)t.zigzag(); return true; }
        if (f == 6 && ty == 5) { h.v2_rep_bytes = (int32_t)t.zigzag(); return true; }
        if (f == 7 && (ty == 1 || ty == 2)) { h.v2_compressed = ty == 1; return true; }
        return false;
      }, 1);
      return true;
    }
    return false;
  });
  return h;
}

// The per-run / per-miniblock tables a parse produces for the device (cfg 2's `labels.code`: ≈ 200 k runs × 24 bytes per 5 M rows). They
// cross PCIe next to the chunk, so once a table outgrows 32 KiB it moves to the pinned pool: out of a pageable std::vector the copy went
// through the runtime's staging buffers ON THE ISSUING THREAD (≈ 2 ms per row group, more than the chunk's own copy). Without a device
// (parsing must work on a CPU-only box) pinned memory cannot be had and the table stays where it is.
template <typename T>
class HostTable {
 public:
  HostTable() = default;
  HostTable(const HostTable&) = delete;
  HostTable& operator=(const HostTable&) = delete;
  HostTable(HostTable&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), pinned_(o.pinned_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
  HostTable& operator=(HostTable&& o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; pinned_ = o.pinned_; o.p_ = nullptr; o.n_ = o.cap_ = 0; } return *this; }
  ~HostTable() { release(); }
  void push_back(const T& v) { if (n_ == cap_) grow(); p_[n_++] = v; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  const T* data() const { return p_; }
  bool pinned() const { return pinned_; }

 private:
  static_assert(std::is_trivially_copyable<T>::value, "tables of plain structs");
  void release() { if (p_ == nullptr) return; if (pinned_) pinned_pool_free(p_); else std::free(p_); p_ = nullptr; }
  void grow() {
    const size_t cap = cap_ == 0 ? 256 : cap_ * 2;
    T* q = nullptr;
    bool pin = false;
    static const bool no_pinned = std::getenv("FDB_PARQUET_NO_PINNED") != nullptr;  // (measurement aid, as for the image)
    if (cap * sizeof(T) > ((size_t)32 << 10) && !no_pinned) {
      // (a pooled block is ≥ 2 MiB: the table's next doublings stay inside it)
      try { q = (T*)pinned_pool_alloc(std::max(cap * sizeof(T), (size_t)2 << 20)); pin = true; } catch (const Error&) { (void)hipGetLastError(); q = nullptr; }
    }
    size_t got = cap;
    if (pin) got = std::max(cap * sizeof(T), (size_t)2 << 20) / sizeof(T);
    else { q = (T*)std::malloc(cap * sizeof(T)); if (q == nullptr) throw Error(FDB_ERR_OOM, "parquet: out of host memory"); }
    if (n_) std::memcpy(q, p_, n_ * sizeof(T));
    release();
    p_ = q; cap_ = got; pinned_ = pin;
  }
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
  bool pinned_ = false;
};

// Walks the run HEADERS of an RLE / bit-packed hybrid stream of `n_values` values of `bw` bits in chunk[off, off + len) and
// appends one FdbPqRun per run, numbered from `first` on. Bit-packed payloads are skipped, not read — except when `count_ones`
// is given (definition levels, bw = 1): then the set bits are counted (popcount over the payload bytes).
void scan_runs(const uint8_t* chunk, size_t off, size_t len, int bw, int64_t n_values, int64_t first, HostTable<FdbPqRun>* runs, int64_t* count_ones) {
  Thrift t{chunk + off, chunk + off + len};
  int64_t done = 0;
  const int vbytes = (bw + 7) / 8;
  while (done < n_values) {
    // (a header is one byte for runs of < 64 values / < 64 groups — nearly all of them: the general varint loop only for the rest)
    uint64_t h;
    if (t.p < t.end && *t.p < 0x80) h = *t.p++; else h = t.varint();
    FdbPqRun r;
    r.start = first + done;
    if (h & 1) {
      const uint64_t groups = h >> 1;
      if (groups == 0) throw Error(FDB_ERR_INVALID, "parquet: empty bit-packed run");
      // a run holds groups × 8 values, the last group padded: more groups than the values still missing is a damaged header —
      // checked BEFORE multiplying (a varint of 2^59 groups would wrap `bytes` / `count` and pass the length test below)
      if (groups > ((uint64_t)(n_values - done) + 7) / 8) throw Error(FDB_ERR_INVALID, "parquet: bit-packed run holds more values than its page");
      const uint64_t bytes = groups * (uint64_t)bw;
      int64_t count = (int64_t)(groups * 8);
      if (count > n_values - done) count = n_values - done;  // (the last group is padded)
      // the final run of a stream may be cut short (writers need not pad it to whole groups)
      const size_t have = (size_t)(t.end - t.p);
      if (have < bytes && (int64_t)(have * 8 / (size_t)std::max(bw, 1)) < count) throw Error(FDB_ERR_INVALID, "parquet: bit-packed run past the end of its page");
      r.kind = 1; r.bit_width = (uint32_t)bw; r.payload = (uint64_t)(t.p - chunk) * 8;
      if (count_ones != nullptr) {
        const int64_t full = count / 8;
        int64_t ones = 0, i = 0;
        for (; i + 8 <= full; i += 8) { uint64_t w; std::memcpy(&w, t.p + i, 8); ones += __builtin_popcountll(w); }
        for (; i < full; i++) ones += __builtin_popcount(t.p[i]);
        if (count % 8) ones += __builtin_popcount(t.p[full] & ((1u << (count % 8)) - 1u));
        *count_ones += ones;
      }
      t.p += std::min<size_t>(have, (size_t)bytes);
      done += count;
    } else {
      int64_t count = (int64_t)(h >> 1);
      if (count == 0) throw Error(FDB_ERR_INVALID, "parquet: empty RLE run");
      t.need((size_t)vbytes);
      uint64_t v = 0;
      if (vbytes == 1) v = *t.p; else for (int i = 0; i < vbytes; i++) v |= (uint64_t)t.p[i] << (8 * i);
      t.p += vbytes;
      if (count > n_values - done) count = n_values - done;
      r.kind = 0; r.bit_width = 0; r.payload = v;
      if (count_ones != nullptr && (v & 1)) *count_ones += count;
      done += count;
    }
    runs->push_back(r);
  }
}

// ---- page decompression (host): the device decodes VALUES; inflating a page is a byte-serial job the host does while it walks
// the page headers anyway. The decompressed pages of a chunk are laid end to end in one image that goes to HBM instead of the
// file's bytes. Codec numbers are parquet.thrift's CompressionCodec.
enum { CODEC_NONE = 0, CODEC_SNAPPY = 1, CODEC_GZIP = 2, CODEC_BROTLI = 4, CODEC_LZ4_HADOOP = 5, CODEC_ZSTD = 6, CODEC_LZ4_RAW = 7 };

bool snappy_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {  // the Snappy block format (format_description.txt)
  size_t ip = 0, op = 0;
  uint64_t len = 0;
  for (int shift = 0;; shift += 7) {  // preamble: uncompressed length as a varint
    if (ip >= n || shift > 35) return false;
    const uint8_t b = src[ip++];
    len |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) break;
  }
  if (len != cap) return false;
  while (ip < n) {
    const uint8_t tag = src[ip++];
    if ((tag & 3) == 0) {  // literal
      size_t l = (size_t)(tag >> 2) + 1;
      if (l > 60) {
        const size_t extra = l - 60;
        if (ip + extra > n) return false;
        l = 0;
        for (size_t i = 0; i < extra; i++) l |= (size_t)src[ip + i] << (8 * i);
        l += 1;
        ip += extra;
      }
      if (ip + l > n || op + l > cap) return false;
      std::memcpy(dst + op, src + ip, l);
      ip += l; op += l;
      continue;
    }
    size_t l, off;
    if ((tag & 3) == 1) { if (ip + 1 > n) return false; l = 4 + ((tag >> 2) & 7); off = ((size_t)(tag >> 5) << 8) | src[ip]; ip += 1; }
    else if ((tag & 3) == 2) { if (ip + 2 > n) return false; l = (size_t)(tag >> 2) + 1; off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8); ip += 2; }
    else { if (ip + 4 > n) return false; l = (size_t)(tag >> 2) + 1; off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8) | ((size_t)src[ip + 2] << 16) | ((size_t)src[ip + 3] << 24); ip += 4; }
    if (off == 0 || off > op || op + l > cap) return false;
    if (off >= l) std::memcpy(dst + op, dst + op - off, l);  // disjoint
    else if (off >= 8) { for (size_t i = 0; i < l; i += 8) std::memcpy(dst + op + i, dst + op - off + i, std::min<size_t>(8, l - i)); }  // a pattern of ≥ 8 bytes: 8 at a time
    else for (size_t i = 0; i < l; i++) dst[op + i] = dst[op - off + i];  // short pattern repeated: byte by byte
    op += l;
  }
  return op == cap;
}

// The first `want` bytes of a Snappy page (the definition levels at the head of a V1 page whose values are inflated on the device):
// the same walk, stopped as soon as `want` bytes exist. `cap` = the page's announced uncompressed size.
bool snappy_prefix(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t want) {
  size_t ip = 0, op = 0;
  uint64_t len = 0;
  for (int shift = 0;; shift += 7) {
    if (ip >= n || shift > 35) return false;
    const uint8_t b = src[ip++];
    len |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) break;
  }
  if (len != cap || want > cap) return false;
  while (ip < n && op < want) {
    const uint8_t tag = src[ip++];
    if ((tag & 3) == 0) {
      size_t l = (size_t)(tag >> 2) + 1;
      if (l > 60) {
        const size_t extra = l - 60;
        if (ip + extra > n) return false;
        l = 0;
        for (size_t i = 0; i < extra; i++) l |= (size_t)src[ip + i] << (8 * i);
        l += 1;
        ip += extra;
      }
      if (ip + l > n || op + l > cap) return false;
      std::memcpy(dst + op, src + ip, std::min(l, want - op));
      ip += l; op += l;
      continue;
    }
    size_t l, off;
    if ((tag & 3) == 1) { if (ip + 1 > n) return false; l = 4 + ((tag >> 2) & 7); off = ((size_t)(tag >> 5) << 8) | src[ip]; ip += 1; }
    else if ((tag & 3) == 2) { if (ip + 2 > n) return false; l = (size_t)(tag >> 2) + 1; off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8); ip += 2; }
    else { if (ip + 4 > n) return false; l = (size_t)(tag >> 2) + 1; off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8) | ((size_t)src[ip + 2] << 16) | ((size_t)src[ip + 3] << 24); ip += 4; }
    if (off == 0 || off > op || op + l > cap) return false;
    for (size_t i = 0; i < l && op + i < want; i++) dst[op + i] = dst[op - off + i];
    op += l;
  }
  return op >= want;
}

// May the device's decoder take this Snappy page? It keeps the page's last 64 KiB of output in an LDS ring, so a copy that reaches
// further back than that (offset > 65 472: legal Snappy — a 4-byte-offset element, or a block longer than 64 KiB as klauspost/compress
// writes them for parquet-go, go.mod) is beyond it. Walks the element tags only (literals are skipped, nothing is copied): a page of
// literals has a handful of them. A malformed stream also answers "no": the host's inflate then reports it.
bool snappy_device_ok(const uint8_t* src, size_t n) {
  size_t ip = 0;
  for (int shift = 0;; shift += 7) {
    if (ip >= n || shift > 35) return false;
    if (!(src[ip++] & 0x80)) break;
  }
  while (ip < n) {
    const uint8_t tag = src[ip++];
    if ((tag & 3) == 0) {
      size_t l = (size_t)(tag >> 2) + 1;
      if (l > 60) {
        const size_t extra = l - 60;
        if (ip + extra > n) return false;
        l = 0;
        for (size_t i = 0; i < extra; i++) l |= (size_t)src[ip + i] << (8 * i);
        l += 1;
        ip += extra;
      }
      if (l > n - ip) return false;
      ip += l;
      continue;
    }
    size_t off;
    if ((tag & 3) == 1) { if (ip + 1 > n) return false; off = ((size_t)(tag >> 5) << 8) | src[ip]; ip += 1; }
    else if ((tag & 3) == 2) { if (ip + 2 > n) return false; off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8); ip += 2; }
    else { if (ip + 4 > n) return false; off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8) | ((size_t)src[ip + 2] << 16) | ((size_t)src[ip + 3] << 24); ip += 4; }
    if (off > 65472) return false;
  }
  return true;
}

typedef int (*lz4_fn)(const char*, char*, int, int);
typedef size_t (*zstd_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_err_fn)(size_t);
lz4_fn lz4_decompress() {
  static lz4_fn f = [] { void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL); return h ? (lz4_fn)dlsym(h, "LZ4_decompress_safe") : (lz4_fn) nullptr; }();
  return f;
}
std::pair<zstd_fn, zstd_err_fn> zstd_decompress() {
  static std::pair<zstd_fn, zstd_err_fn> f = [] {
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    return h ? std::make_pair((zstd_fn)dlsym(h, "ZSTD_decompress"), (zstd_err_fn)dlsym(h, "ZSTD_isError")) : std::make_pair((zstd_fn) nullptr, (zstd_err_fn) nullptr);
  }();
  return f;
}

typedef int (*brotli_fn)(size_t, const uint8_t*, size_t*, uint8_t*);
brotli_fn brotli_decompress() {
  static brotli_fn f = [] { void* h = dlopen("libbrotlidec.so.1", RTLD_NOW | RTLD_LOCAL); return h ? (brotli_fn)dlsym(h, "BrotliDecoderDecompress") : (brotli_fn) nullptr; }();
  return f;
}

// dst[0, cap) = the decompressed bytes of src[0, n); throws on a codec that is not available or on corrupt input.
void inflate_page(int codec, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  if (cap == 0) return;
  switch (codec) {
    case CODEC_SNAPPY:
      if (!snappy_raw(src, n, dst, cap)) throw Error(FDB_ERR_INVALID, "parquet: corrupt Snappy page");
      return;
    case CODEC_GZIP: {
      z_stream z;
      std::memset(&z, 0, sizeof(z));
      if (inflateInit2(&z, 15 + 32) != Z_OK) throw Error(FDB_ERR_OOM, "parquet: zlib init failed");
      z.next_in = const_cast<Bytef*>(src); z.avail_in = (uInt)n; z.next_out = dst; z.avail_out = (uInt)cap;
      const int rc = inflate(&z, Z_FINISH);
      const size_t got = cap - z.avail_out;
      inflateEnd(&z);
      if (rc != Z_STREAM_END || got != cap) throw Error(FDB_ERR_INVALID, "parquet: corrupt GZIP page");
      return;
    }
    case CODEC_ZSTD: {
      auto f = zstd_decompress();
      if (f.first == nullptr || f.second == nullptr) throw Error(FDB_ERR_UNSUPPORTED, "parquet: ZSTD pages need libzstd.so.1 on this host");
      const size_t got = f.first(dst, cap, src, n);
      if (f.second(got) || got != cap) throw Error(FDB_ERR_INVALID, "parquet: corrupt ZSTD page");
      return;
    }
    case CODEC_BROTLI: {
      brotli_fn f = brotli_decompress();
      if (f == nullptr) throw Error(FDB_ERR_UNSUPPORTED, "parquet: BROTLI pages need libbrotlidec.so.1 on this host");
      size_t got = cap;
      if (f(n, src, &got, dst) != 1 || got != cap) throw Error(FDB_ERR_INVALID, "parquet: corrupt BROTLI page");
      return;
    }
    case CODEC_LZ4_RAW: case CODEC_LZ4_HADOOP: {
      lz4_fn f = lz4_decompress();
      if (f == nullptr) throw Error(FDB_ERR_UNSUPPORTED, "parquet: LZ4 pages need liblz4.so.1 on this host");
      if (f((const char*)src, (char*)dst, (int)n, (int)cap) == (int)cap) return;
      // the deprecated "LZ4" codec may carry Hadoop's framing: [uncompressed size BE32][compressed size BE32][block] …
      size_t ip = 0, op = 0;
      while (ip + 8 <= n && op < cap) {
        const size_t ul = ((size_t)src[ip] << 24) | ((size_t)src[ip + 1] << 16) | ((size_t)src[ip + 2] << 8) | src[ip + 3];
        const size_t cl = ((size_t)src[ip + 4] << 24) | ((size_t)src[ip + 5] << 16) | ((size_t)src[ip + 6] << 8) | src[ip + 7];
        ip += 8;
        if (ip + cl > n || op + ul > cap || f((const char*)src + ip, (char*)dst + op, (int)cl, (int)ul) != (int)ul) break;
        ip += cl; op += ul;
      }
      if (op != cap) throw Error(FDB_ERR_INVALID, "parquet: corrupt LZ4 page");
      return;
    }
    default:
      throw Error(FDB_ERR_UNSUPPORTED, "parquet: compression codec " + std::to_string(codec) + " is not supported (UNCOMPRESSED, SNAPPY, GZIP, BROTLI, ZSTD, LZ4_RAW are)");
  }
}

// DELTA_BINARY_PACKED on the host (parquet-format Encodings.md): only the LENGTH streams of the DELTA byte-array encodings are
// decoded here (one integer per value; the values of INT64 columns are unpacked and summed on the device). Reads `count` values
// — the header's own total must say the same — and leaves `d.p` behind the last miniblock that holds one.
std::vector<int64_t> delta_bp_host(Thrift& d, int64_t count) {
  const uint64_t block = d.varint(), n_mini = d.varint(), total = d.varint();
  uint64_t cur = (uint64_t)d.zigzag();
  if (block == 0 || block % 128 != 0 || n_mini == 0 || block % n_mini != 0 || (block / n_mini) % 32 != 0 || block > (1u << 24))
    throw Error(FDB_ERR_INVALID, "parquet: malformed DELTA_BINARY_PACKED header");
  if ((int64_t)total != count) throw Error(FDB_ERR_INVALID, "parquet: DELTA_BINARY_PACKED value count differs from the page's");
  std::vector<int64_t> out;
  if (count == 0) return out;
  out.reserve((size_t)count);
  out.push_back((int64_t)cur);
  const uint64_t vpm = block / n_mini;
  uint64_t left = total - 1;
  while (left > 0) {
    const uint64_t min_delta = (uint64_t)d.zigzag();
    d.need((size_t)n_mini);
    const uint8_t* widths = d.p;
    d.p += n_mini;
    for (uint64_t m = 0; m < n_mini && left > 0; m++) {
      const uint32_t w = widths[m];
      if (w > 64) throw Error(FDB_ERR_INVALID, "parquet: DELTA_BINARY_PACKED bit width > 64");
      const size_t bytes = (size_t)(vpm / 8) * w;
      d.need(bytes);
      const uint64_t take = std::min<uint64_t>(left, vpm);
      for (uint64_t i = 0; i < take; i++) {
        uint64_t v = 0;
        if (w != 0) {
          const uint64_t bit = i * w;
          const uint8_t* q = d.p + (bit >> 3);
          const uint32_t sh = (uint32_t)(bit & 7);
          // (≤ 9 bytes hold the value; byte-wise so that nothing past the miniblock is read)
          const size_t nb = (size_t)((sh + w + 7) / 8);
          unsigned __int128 acc = 0;
          for (size_t k = 0; k < nb; k++) acc |= (unsigned __int128)q[k] << (8 * k);
          v = (uint64_t)(acc >> sh);
          if (w < 64) v &= ((uint64_t)1 << w) - 1;
        }
        cur += min_delta + v;  // (wraps like the reference's decoder)
        out.push_back((int64_t)cur);
      }
      d.p += bytes;
      left -= take;
    }
  }
  return out;
}

// The non-NULL values of a DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY page, one std::string each.
// DELTA_LENGTH_BYTE_ARRAY: <lengths, DELTA_BINARY_PACKED> <all bytes back to back>.
// DELTA_BYTE_ARRAY: <prefix lengths, DELTA_BINARY_PACKED> <suffixes, DELTA_LENGTH_BYTE_ARRAY>; value i = first prefix[i] bytes of value i − 1 + suffix i.
void delta_byte_array_host(const uint8_t* p, size_t len, int64_t count, bool with_prefixes, std::vector<std::string>* out) {
  Thrift d{p, p + len};
  std::vector<int64_t> prefixes;
  if (with_prefixes) prefixes = delta_bp_host(d, count);
  const std::vector<int64_t> lengths = delta_bp_host(d, count);
  out->clear();
  out->reserve((size_t)count);
  std::string prev;
  for (int64_t i = 0; i < count; i++) {
    const int64_t l = lengths[(size_t)i];
    if (l < 0 || (size_t)l > (size_t)(d.end - d.p)) throw Error(FDB_ERR_INVALID, "parquet: DELTA byte-array page truncated");
    std::string v;
    if (with_prefixes) {
      const int64_t pl = prefixes[(size_t)i];
      if (pl < 0 || (size_t)pl > prev.size()) throw Error(FDB_ERR_INVALID, "parquet: DELTA_BYTE_ARRAY prefix longer than the previous value");
      v.assign(prev, 0, (size_t)pl);
    }
    v.append((const char*)d.p, (size_t)l);
    d.p += l;
    if (with_prefixes) prev = v;
    out->push_back(std::move(v));
  }
}

// The image of a compressed chunk's inflated pages: sized by a first walk over the page headers, in pinned memory when the process
// has a device (the copy engine reads it directly; the pool re-uses blocks, so no page faults after the first row group), in plain
// memory otherwise (parsing must work — and refuse bad chunks — without a GPU).
struct Image {
  uint8_t* p = nullptr;
  size_t cap = 0, used = 0;
  bool pinned = false;
  Image() = default;
  Image(const Image&) = delete;
  Image& operator=(const Image&) = delete;
  Image(Image&& o) noexcept : p(o.p), cap(o.cap), used(o.used), pinned(o.pinned) { o.p = nullptr; o.cap = o.used = 0; }
  Image& operator=(Image&& o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; used = o.used; pinned = o.pinned; o.p = nullptr; o.cap = o.used = 0; } return *this; }
  ~Image() { release(); }
  void release() { if (p == nullptr) return; if (pinned) pinned_pool_free(p); else std::free(p); p = nullptr; }
  void allocate(size_t bytes) {
    release();
    cap = bytes; used = 0;
    static const bool no_pinned = std::getenv("FDB_PARQUET_NO_PINNED") != nullptr;  // (measurement aid)
    try { if (no_pinned) throw Error(FDB_ERR_DEVICE, "off"); p = (uint8_t*)pinned_pool_alloc(bytes); pinned = true; }
    catch (const Error&) { (void)hipGetLastError(); p = (uint8_t*)std::malloc(std::max<size_t>(bytes, 1)); pinned = false; if (p == nullptr) throw Error(FDB_ERR_OOM, "parquet: out of host memory"); }
  }
  bool empty() const { return used == 0; }
  const uint8_t* data() const { return p; }
  size_t size() const { return used; }
};

struct ParsedChunk {
  Image image;                             // compressed chunks: the decompressed page bodies end to end (what goes to HBM); else empty
  std::shared_ptr<HostDict> dict;          // BYTE_ARRAY columns
  HostTable<FdbPqRun> def_runs;            // optional columns: one entry per run, row-numbered
  HostTable<FdbPqRun> idx_runs;            // dictionary-encoded columns: rank-numbered
  std::vector<FdbPqPlainPage> plain_pages; // PLAIN fixed-width columns
  std::vector<FdbPqDeltaPage> delta_pages; // DELTA_BINARY_PACKED INT64 columns
  HostTable<FdbPqDeltaMini> delta_minis;
  int64_t non_null = 0;
  uint32_t max_index_bits = 0;
  // Pages inflated on the DEVICE (snappy_decode_kernel): SNAPPY pages of PLAIN fixed-width values whose compressed size says they are
  // literals — inflating those on the host is a copy of the whole page into the image that the device can do at memory speed, while
  // match-heavy pages (levels, indices, deltas) are byte-serial work the host's threads are better at. Their place in the
  // image stays empty on the host (but for a V1 page's definition levels, which the host needs): the image is shipped in `host_spans`.
  struct DevPage { size_t raw_off, comp, at, len; };  // compressed bytes [raw_off, raw_off + comp) of the chunk → image bytes [at, at + len)
  std::vector<DevPage> dev_pages;
  std::vector<std::pair<size_t, size_t>> host_spans;  // [begin, end) of the image that crosses PCIe (only filled when dev_pages is not empty)
};

// A page whose body has to be inflated into the chunk's image: `at` is fixed by the first walk over the page headers, so the pages
// of every chunk of a row group can be inflated side by side (they are independent) before anything is parsed.
struct InflateJob { int codec; const uint8_t* raw; size_t comp, prefix, body_len; uint8_t* dst; };

// First walk over a chunk's page headers: validates the page chain, decides whether the chunk needs an image — compressed pages, or
// BYTE_ARRAY data pages that are not dictionary-encoded (their values are dictionary-encoded HERE, one hash probe per value like the
// reference's own BinaryDictionaryBuilder.Append, pqarrow/writer/writer.go:391-405, and the indices are appended to the image as a
// 32-bit-wide bit-packed run, which is all the device needs) — sizes it, and lists the inflate jobs of its data pages.
void plan_chunk(const fdb_parquet_chunk& c, int64_t n_rows, ParsedChunk* out, std::vector<InflateJob>* jobs) {
  if (c.data == nullptr || c.n_bytes <= 0) throw Error(FDB_ERR_INVALID, std::string("parquet: empty column chunk for ") + (c.name ? c.name : "?"));
  if (c.optional != 0 && c.optional != 1) throw Error(FDB_ERR_UNSUPPORTED, "parquet: nested / repeated columns are not supported (max definition level > 1)");
  const bool is_bytes = c.physical_type == PT_BYTE_ARRAY, is_fixed8 = c.physical_type == PT_INT64 || c.physical_type == PT_DOUBLE, is_bool = c.physical_type == PT_BOOLEAN;
  if (!is_bytes && !is_fixed8 && !is_bool)
    throw Error(FDB_ERR_UNSUPPORTED, "parquet: only BOOLEAN, INT64, DOUBLE and BYTE_ARRAY columns are decoded (what pqarrow/convert/convert.go maps to Arrow)");
  bool use_image = c.codec != CODEC_NONE;
  Thrift w{c.data, c.data + c.n_bytes};
  size_t need = 0, extra = 0;
  int64_t values = 0;
  struct Pg { const uint8_t* raw; size_t comp, prefix, body_len; bool packed; bool device; bool v1_levels; };
  // ($FDB_PARQUET_HOST_INFLATE: every page on the host, as before round 4)
  static const bool device_inflate = std::getenv("FDB_PARQUET_HOST_INFLATE") == nullptr;
  std::vector<Pg> pages;
  while (w.p < w.end && values < n_rows) {
    const PageHeader h = read_page_header(w);
    if (h.compressed < 0 || h.uncompressed < 0 || (size_t)(w.end - w.p) < (size_t)h.compressed) throw Error(FDB_ERR_INVALID, "parquet: page runs past the end of the column chunk");
    const uint8_t* raw = w.p;
    w.p += (size_t)h.compressed;
    if (h.type != PQ_DATA_PAGE && h.type != PQ_DATA_PAGE_V2) continue;
    // (a damaged header must not size the image: no page holds more values than the row group has rows)
    if (h.num_values < 0 || values + h.num_values > n_rows) throw Error(FDB_ERR_INVALID, "parquet: pages hold more values than the row group has rows");
    values += h.num_values;
    const size_t prefix = h.type == PQ_DATA_PAGE_V2 ? (size_t)h.v2_def_bytes + (size_t)h.v2_rep_bytes : 0;
    if (prefix > (size_t)h.compressed || prefix > (size_t)h.uncompressed) throw Error(FDB_ERR_INVALID, "parquet: levels run past the page");
    const bool packed = c.codec != CODEC_NONE && (h.type != PQ_DATA_PAGE_V2 || h.v2_compressed);
    const size_t comp_body = (size_t)h.compressed - prefix, plain_body = (size_t)h.uncompressed - prefix;
    static const size_t device_min = std::getenv("FDB_PARQUET_DEVICE_MIN_BYTES") ? (size_t)std::atoll(std::getenv("FDB_PARQUET_DEVICE_MIN_BYTES")) : ((size_t)32 << 10);  // (tuning aid; 256 KiB until round 6: with every row group of a call on the host threads at once, inflating 1 000 literal pages of 48 KB per row group there was the host part's largest item — 3.2 ms against 1.3)
    const bool on_device = device_inflate && packed && c.codec == CODEC_SNAPPY && is_fixed8 && h.encoding == ENC_PLAIN && plain_body >= device_min &&
                           comp_body * 10 >= plain_body * 9 && plain_body < ((size_t)1 << 31) && snappy_device_ok(raw + prefix, comp_body);
    pages.push_back(Pg{raw, (size_t)h.compressed, prefix, (size_t)h.uncompressed, packed, on_device, on_device && h.type == PQ_DATA_PAGE && c.optional != 0});
    need += (size_t)h.uncompressed + 8;  // (+8: keeps every 64-bit window of the device's readers inside the image)
    if (is_bytes && h.encoding != ENC_RLE_DICTIONARY && h.encoding != ENC_PLAIN_DICTIONARY) { use_image = true; extra += (size_t)h.num_values * 4 + 16; }
  }
  if (!use_image) return;
  out->image.allocate(need + extra + 64);
  size_t at = 0;
  for (const Pg& g : pages) {
    uint8_t* dst = out->image.p + at;
    std::memcpy(dst, g.raw, g.prefix);  // a V2 page keeps its levels uncompressed in front of the (possibly) compressed values
    if (g.device) {
      if (g.v1_levels) {  // <4-byte length> <RLE levels> at the head of the inflated body: the host reads them, the device inflates them again with the values
        uint32_t dl = 0;
        if (g.body_len < 4 || !snappy_prefix(g.raw, g.comp, dst, g.body_len, 4)) throw Error(FDB_ERR_INVALID, "parquet: corrupt Snappy page");
        std::memcpy(&dl, dst, 4);
        if ((size_t)dl + 4 > g.body_len) throw Error(FDB_ERR_INVALID, "parquet: definition levels run past the page");
        if (!snappy_prefix(g.raw, g.comp, dst, g.body_len, (size_t)dl + 4)) throw Error(FDB_ERR_INVALID, "parquet: corrupt Snappy page");
      }
      out->dev_pages.push_back(ParsedChunk::DevPage{(size_t)(g.raw - c.data) + g.prefix, g.comp - g.prefix, at + g.prefix, g.body_len - g.prefix});
      if (g.prefix > 0) out->host_spans.emplace_back(at, at + g.prefix);
    }
    else if (g.packed) { jobs->push_back(InflateJob{c.codec, g.raw, g.comp, g.prefix, g.body_len, dst}); out->host_spans.emplace_back(at, at + g.body_len + 8); }
    else {
      out->host_spans.emplace_back(at, at + g.body_len + 8);
      if (g.comp != g.body_len) throw Error(FDB_ERR_INVALID, "parquet: uncompressed page with differing sizes");
      std::memcpy(dst + g.prefix, g.raw + g.prefix, g.body_len - g.prefix);
    }
    at += g.body_len + 8;
  }
  out->image.used = at;
  if (out->dev_pages.empty()) out->host_spans.clear();
  else {  // neighbours merged: one copy command per stretch of host pages
    std::vector<std::pair<size_t, size_t>> merged;
    for (const auto& sp : out->host_spans) {
      if (!merged.empty() && merged.back().second == sp.first) merged.back().second = sp.second;
      else merged.push_back(sp);
    }
    out->host_spans.swap(merged);
  }
}

inline void run_inflate(const InflateJob& j) { inflate_page(j.codec, j.raw + j.prefix, j.comp - j.prefix, j.dst + j.prefix, j.body_len - j.prefix); }

// Second walk: page by page — bodies in place (uncompressed chunks) or in the image, where plan_chunk / the inflate jobs put them.
void parse_chunk(const fdb_parquet_chunk& c, int64_t n_rows, ParsedChunk* outp) {
  ParsedChunk& out = *outp;
  const bool is_bytes = c.physical_type == PT_BYTE_ARRAY, is_fixed8 = c.physical_type == PT_INT64 || c.physical_type == PT_DOUBLE, is_bool = c.physical_type == PT_BOOLEAN;
  const bool use_image = out.image.p != nullptr;
  std::vector<std::string> dict_values;                    // the chunk's dictionary: its dictionary page, then values of pages without dictionary indices
  std::unordered_map<std::string, uint32_t> dict_lookup;   // built when the first such page arrives
  bool lookup_ready = false;
  const uint8_t* base = use_image ? out.image.data() : c.data;  // what run / page offsets are relative to
  Thrift t{c.data, c.data + c.n_bytes};
  int64_t rows_done = 0, rank_done = 0;
  size_t image_at = 0;  // next data page's place in the image
  bool have_dict = false;
  std::vector<std::string> page_values;
  while (t.p < t.end && rows_done < n_rows) {
    const PageHeader h = read_page_header(t);
    if (h.compressed < 0 || h.uncompressed < 0 || (size_t)(t.end - t.p) < (size_t)h.compressed) throw Error(FDB_ERR_INVALID, "parquet: page runs past the end of the column chunk");
    if (c.codec == CODEC_NONE && h.compressed != h.uncompressed) throw Error(FDB_ERR_INVALID, "parquet: page sizes disagree in an UNCOMPRESSED chunk");
    const uint8_t* raw = t.p;  // the page's bytes in the file
    t.p += (size_t)h.compressed;
    if (h.type == PQ_DICTIONARY_PAGE) {
      if (!is_bytes) throw Error(FDB_ERR_UNSUPPORTED, "parquet: dictionary-encoded numeric columns are not supported on the device path");
      if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICTIONARY) throw Error(FDB_ERR_UNSUPPORTED, "parquet: dictionary page encoding");
      if (have_dict) throw Error(FDB_ERR_INVALID, "parquet: two dictionary pages in one column chunk");
      std::vector<uint8_t> dict_tmp;
      const uint8_t* body = raw;
      size_t body_len = (size_t)h.compressed;
      if (c.codec != CODEC_NONE) {  // (small: inflated here)
        body_len = (size_t)h.uncompressed;
        dict_tmp.resize(body_len + 8);
        inflate_page(c.codec, raw, (size_t)h.compressed, dict_tmp.data(), body_len);
        body = dict_tmp.data();
      }
      if (h.num_values < 0 || (size_t)h.num_values > body_len / 4) throw Error(FDB_ERR_INVALID, "parquet: dictionary page truncated");  // (every value has a 4-byte length)
      dict_values.reserve((size_t)h.num_values);
      size_t o = 0;
      for (int32_t i = 0; i < h.num_values; i++) {
        if (o + 4 > body_len) throw Error(FDB_ERR_INVALID, "parquet: dictionary page truncated");
        uint32_t len; std::memcpy(&len, body + o, 4); o += 4;
        if (o + len > body_len) throw Error(FDB_ERR_INVALID, "parquet: dictionary page truncated");
        dict_values.emplace_back((const char*)body + o, len); o += len;
      }
      have_dict = true;
      continue;
    }
    if (h.type != PQ_DATA_PAGE && h.type != PQ_DATA_PAGE_V2) continue;  // (index pages etc.)
    if (h.num_values < 0 || rows_done + h.num_values > n_rows) throw Error(FDB_ERR_INVALID, "parquet: pages hold more values than the row group has rows");
    size_t body_off = (size_t)(raw - c.data), body_len = (size_t)h.compressed;
    if (use_image) {
      body_off = image_at; body_len = (size_t)h.uncompressed;
      image_at += body_len + 8;
      if (image_at > out.image.cap) throw Error(FDB_ERR_INVALID, "parquet: pages hold more values than the row group has rows");
    }
    size_t voff = body_off, vlen = body_len;  // the values part of the page
    int64_t page_non_null = h.num_values;
    if (h.type == PQ_DATA_PAGE_V2) {
      if (h.v2_rep_bytes != 0) throw Error(FDB_ERR_UNSUPPORTED, "parquet: repeated columns are not supported");
      if ((size_t)h.v2_def_bytes > body_len) throw Error(FDB_ERR_INVALID, "parquet: definition levels run past the page");
      if (c.optional) {
        if (h.v2_def_bytes > 0) { page_non_null = 0; scan_runs(base, voff, (size_t)h.v2_def_bytes, 1, h.num_values, rows_done, &out.def_runs, &page_non_null); }
        else out.def_runs.push_back(FdbPqRun{rows_done, 1ull, 0u, 0u});
      }
      voff += (size_t)h.v2_def_bytes; vlen -= (size_t)h.v2_def_bytes;
    } else if (c.optional) {
      if (h.def_encoding != ENC_RLE) throw Error(FDB_ERR_UNSUPPORTED, "parquet: definition levels must be RLE-encoded");
      if (vlen < 4) throw Error(FDB_ERR_INVALID, "parquet: data page without definition levels");
      uint32_t dl; std::memcpy(&dl, base + voff, 4);
      if ((size_t)dl + 4 > vlen) throw Error(FDB_ERR_INVALID, "parquet: definition levels run past the page");
      page_non_null = 0;
      scan_runs(base, voff + 4, dl, 1, h.num_values, rows_done, &out.def_runs, &page_non_null);
      voff += 4 + dl; vlen -= 4 + dl;
    }
    if (is_bool) {
      // PLAIN: one bit per value, LSB first; RLE (data page V2): <4-byte length> + the RLE / bit-packed hybrid at bit width 1
      if (h.encoding == ENC_PLAIN) {
        if ((size_t)(page_non_null + 7) / 8 > vlen) throw Error(FDB_ERR_INVALID, "parquet: PLAIN BOOLEAN page shorter than its values");
        if (page_non_null > 0) out.idx_runs.push_back(FdbPqRun{rank_done, (uint64_t)voff * 8u, 1u, 1u});
      } else if (h.encoding == ENC_RLE) {
        if (vlen < 4) throw Error(FDB_ERR_INVALID, "parquet: RLE BOOLEAN page without its length");
        uint32_t rl; std::memcpy(&rl, base + voff, 4);
        if ((size_t)rl + 4 > vlen) throw Error(FDB_ERR_INVALID, "parquet: RLE BOOLEAN values run past the page");
        if (page_non_null > 0) scan_runs(base, voff + 4, rl, 1, page_non_null, rank_done, &out.idx_runs, nullptr);
      } else throw Error(FDB_ERR_UNSUPPORTED, "parquet: BOOLEAN pages must be PLAIN or RLE");
    } else if (is_fixed8) {
      if (h.encoding == ENC_DELTA_BINARY_PACKED && c.physical_type == PT_INT64) {
        // <block size> <miniblocks per block> <total value count> <first value> then per block <min delta> <bit widths> <miniblocks>
        // (parquet-format Encodings.md): only the headers are read here, the deltas are unpacked and summed on the device
        if (!out.plain_pages.empty()) throw Error(FDB_ERR_UNSUPPORTED, "parquet: PLAIN and DELTA_BINARY_PACKED pages in one column chunk");
        if (page_non_null > 0) {
          Thrift d{base + voff, base + voff + vlen};
          const uint64_t block = d.varint(), n_mini = d.varint(), count = d.varint();
          const uint64_t first = (uint64_t)d.zigzag();
          if (block == 0 || block % 128 != 0 || n_mini == 0 || block % n_mini != 0 || (block / n_mini) % 32 != 0 || block > (1u << 24))
            throw Error(FDB_ERR_INVALID, "parquet: malformed DELTA_BINARY_PACKED header");
          if ((int64_t)count != page_non_null) throw Error(FDB_ERR_INVALID, "parquet: DELTA_BINARY_PACKED value count differs from the page's");
          const uint64_t vpm = block / n_mini;
          FdbPqDeltaPage P{rank_done, page_non_null, first, (int32_t)out.delta_minis.size(), (int32_t)vpm};
          uint64_t left = count - 1;  // deltas still to be located
          while (left > 0) {
            const uint64_t min_delta = (uint64_t)d.zigzag();
            if ((size_t)(d.end - d.p) < n_mini) throw Error(FDB_ERR_INVALID, "parquet: DELTA_BINARY_PACKED block truncated");
            const uint8_t* widths = d.p;
            d.p += n_mini;
            for (uint64_t m = 0; m < n_mini && left > 0; m++) {  // (miniblocks past the last value have no body)
              const uint32_t w = widths[m];
              if (w > 64) throw Error(FDB_ERR_INVALID, "parquet: DELTA_BINARY_PACKED bit width > 64");
              const size_t bytes = (size_t)(vpm / 8) * w;
              if ((size_t)(d.end - d.p) < bytes) throw Error(FDB_ERR_INVALID, "parquet: DELTA_BINARY_PACKED miniblock truncated");
              out.delta_minis.push_back(FdbPqDeltaMini{(uint64_t)(d.p - base) * 8u, min_delta, w, 0u});
              d.p += bytes;
              left -= std::min<uint64_t>(left, vpm);
            }
          }
          if (out.delta_minis.size() > 0x7FFFFFF0u) throw Error(FDB_ERR_UNSUPPORTED, "parquet: too many DELTA_BINARY_PACKED miniblocks");
          out.delta_pages.push_back(P);
        }
      } else {
        if (h.encoding != ENC_PLAIN) throw Error(FDB_ERR_UNSUPPORTED, "parquet: INT64 pages must be PLAIN or DELTA_BINARY_PACKED, DOUBLE pages PLAIN");
        if (!out.delta_pages.empty()) throw Error(FDB_ERR_UNSUPPORTED, "parquet: PLAIN and DELTA_BINARY_PACKED pages in one column chunk");
        if ((size_t)page_non_null * 8 > vlen) throw Error(FDB_ERR_INVALID, "parquet: PLAIN page shorter than its values");
        out.plain_pages.push_back(FdbPqPlainPage{rank_done, (int64_t)voff});
      }
    } else if (h.encoding == ENC_RLE_DICTIONARY || h.encoding == ENC_PLAIN_DICTIONARY) {
      if (!have_dict) throw Error(FDB_ERR_INVALID, "parquet: dictionary-encoded page without a dictionary page");
      if (page_non_null > 0) {
        if (vlen < 1) throw Error(FDB_ERR_INVALID, "parquet: dictionary-index page without a bit width");
        const int bw = base[voff];
        if (bw > 32) throw Error(FDB_ERR_INVALID, "parquet: dictionary index bit width > 32");
        out.max_index_bits = std::max<uint32_t>(out.max_index_bits, (uint32_t)bw);
        if (bw == 0) out.idx_runs.push_back(FdbPqRun{rank_done, 0ull, 0u, 0u});  // every index is 0
        else scan_runs(base, voff + 1, vlen - 1, bw, page_non_null, rank_done, &out.idx_runs, nullptr);
      }
    } else if (h.encoding == ENC_PLAIN || h.encoding == ENC_DELTA_LENGTH_BYTE_ARRAY || h.encoding == ENC_DELTA_BYTE_ARRAY) {
      // values without dictionary indices — a writer without dictionaries, its dictionary fallback half way through a chunk, or
      // the DELTA byte-array encodings a FrostDB schema may ask for (schema.proto:62-65): every non-NULL value → an index into
      // the chunk's (growing) dictionary, the indices to the device as one 32-bit-wide bit-packed run
      if (!lookup_ready) {
        dict_lookup.reserve(dict_values.size() * 2 + 1024);
        for (size_t i = 0; i < dict_values.size(); i++) dict_lookup.emplace(dict_values[i], (uint32_t)i);
        lookup_ready = true;
      }
      const size_t at = (out.image.used + 3) & ~(size_t)3;
      if (!use_image || at + (size_t)page_non_null * 4 + 8 > out.image.cap) throw Error(FDB_ERR_INVALID, "parquet: pages hold more values than the row group has rows");
      uint32_t* idx = reinterpret_cast<uint32_t*>(out.image.p + at);
      auto intern = [&](std::string&& v) -> uint32_t {
        auto it = dict_lookup.find(v);
        if (it == dict_lookup.end()) {
          if (dict_values.size() >= 0xFFFFFFF0u) throw Error(FDB_ERR_UNSUPPORTED, "parquet: more than 2^32 distinct values in a column chunk");
          it = dict_lookup.emplace(v, (uint32_t)dict_values.size()).first;
          dict_values.push_back(std::move(v));
        }
        return it->second;
      };
      if (h.encoding == ENC_PLAIN) {  // <length><bytes> per value
        size_t o = voff;
        const size_t end = voff + vlen;
        for (int64_t i = 0; i < page_non_null; i++) {
          if (o + 4 > end) throw Error(FDB_ERR_INVALID, "parquet: PLAIN BYTE_ARRAY page truncated");
          uint32_t len; std::memcpy(&len, base + o, 4); o += 4;
          if ((size_t)len > end - o) throw Error(FDB_ERR_INVALID, "parquet: PLAIN BYTE_ARRAY page truncated");
          idx[i] = intern(std::string((const char*)base + o, len)); o += len;
        }
      } else {
        delta_byte_array_host(base + voff, vlen, page_non_null, h.encoding == ENC_DELTA_BYTE_ARRAY, &page_values);
        for (int64_t i = 0; i < page_non_null; i++) idx[i] = intern(std::move(page_values[(size_t)i]));
      }
      out.image.used = at + (size_t)page_non_null * 4 + 8;
      if (page_non_null > 0) { out.idx_runs.push_back(FdbPqRun{rank_done, (uint64_t)at * 8u, 32u, 1u}); out.max_index_bits = 32; }
    } else {
      throw Error(FDB_ERR_UNSUPPORTED, "parquet: BYTE_ARRAY pages must be PLAIN, (PLAIN_ / RLE_)DICTIONARY, DELTA_LENGTH_BYTE_ARRAY or DELTA_BYTE_ARRAY");
    }
    rows_done += h.num_values;
    rank_done += page_non_null;
  }
  if (rows_done != n_rows) throw Error(FDB_ERR_INVALID, std::string("parquet: column chunk ") + (c.name ? c.name : "?") + " holds " + std::to_string(rows_done) + " values, the row group has " + std::to_string(n_rows) + " rows");
  out.non_null = rank_done;
  if (is_bytes && rank_done > 0 && dict_values.empty()) throw Error(FDB_ERR_INVALID, std::string("parquet: column chunk ") + (c.name ? c.name : "?") + " has values but an empty dictionary");
  if (is_bytes) out.dict = make_dictionary(std::move(dict_values), c.utf8 ? "u" : "z");
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t kTailPad = 256;

std::atomic<int64_t> g_pq_calls{0}, g_pq_host_us{0}, g_pq_device_us{0}, g_pq_file_bytes{0}, g_pq_out_bytes{0};

}  // namespace

void parquet_stats(int64_t* calls, double* host_ms, double* device_ms, int64_t* file_bytes, int64_t* out_bytes) {
  if (calls) *calls = g_pq_calls.load();
  if (host_ms) *host_ms = (double)g_pq_host_us.load() / 1000.0;
  if (device_ms) *device_ms = (double)g_pq_device_us.load() / 1000.0;
  if (file_bytes) *file_bytes = g_pq_file_bytes.load();
  if (out_bytes) *out_bytes = g_pq_out_bytes.load();
}

// Row groups → resident batches, ONE call: the page headers of every chunk of every row group are walked first (host threads), the
// chunks that cross PCIe as they are start crossing it at once — all of them queued on ONE copy stream, row group after row group, so
// the link never waits for a host phase in between —, the compressed pages of all row groups are inflated and all chunks parsed side by
// side, and the calling thread issues a row group's decode kernels as soon as ITS chunks are parsed, while later row groups are still
// on the host threads. One wait at the end. (Until round 6 a call took one row group and the caller overlapped calls from threads of
// its own: the link idled whenever the calls' host phases coincided, and four calls walked their run headers on four threads instead of
// sixteen.) Device memory: the chunks (or images) of ALL the call's row groups are in HBM at once — callers bound a call by bytes.
std::vector<std::unique_ptr<DeviceBatch>> batches_from_parquet(const fdb_parquet_row_group* groups, int32_t n_groups, int device) {
  if (groups == nullptr || n_groups <= 0) throw Error(FDB_ERR_INVALID, "parquet: no row groups");
  for (int32_t g = 0; g < n_groups; g++)
    if (groups[g].chunks == nullptr || groups[g].n_chunks <= 0 || groups[g].n_rows < 0) throw Error(FDB_ERR_INVALID, "parquet: no column chunks");
  struct Piece { size_t val_off, bit_off; };
  struct Group {
    const fdb_parquet_chunk* chunks; int32_t n_chunks; int64_t n_rows;
    std::vector<ParsedChunk> parsed;
    std::vector<std::vector<InflateJob>> jobs;  // per chunk
    std::vector<uint8_t*> early;                // device copies of chunks that cross PCIe as they are, started before the host part
    std::vector<hipEvent_t> copied;             // per early-copied chunk: its copy is complete
    std::vector<Piece> pieces;
    std::unique_ptr<DeviceBatch> b;
    unsigned long long* h_totals = nullptr;     // per chunk, in the call's pinned block
    std::vector<unsigned long long*> d_totals;
    std::vector<int32_t> flag_at;               // per chunk: its word in the index-check flags, or -1
    std::atomic<int> parse_left{0};
    std::vector<char> parsed_ok;                // per early-copied chunk: its parse is complete (under ready_mu)
    std::vector<double> parse_us;
  };
  std::vector<Group> G((size_t)n_groups);
  size_t all_chunks = 0;
  for (int32_t g = 0; g < n_groups; g++) {
    Group& R = G[(size_t)g];
    R.chunks = groups[g].chunks; R.n_chunks = groups[g].n_chunks; R.n_rows = groups[g].n_rows;
    const size_t n = (size_t)R.n_chunks;
    R.parsed.resize(n); R.jobs.resize(n); R.early.assign(n, nullptr); R.copied.assign(n, nullptr); R.pieces.resize(n);
    R.d_totals.assign(n, nullptr); R.flag_at.assign(n, -1); R.parse_us.assign(n, 0.0); R.parsed_ok.assign(n, 0);
    all_chunks += n;
  }
  const auto t_host0 = std::chrono::steady_clock::now();
  // The context (stream, staging, scratch cache) is taken as soon as the page headers have been walked — see the early copies below — or,
  // when there is nothing to copy early, after the host part as before. Whatever happens, the streams are idle before anything is given back.
  Context* ctx = nullptr;
  hipStream_t copy_stream = nullptr;  // the early copies' own queue: chunk i's decode kernels run while chunk i + 1 still crosses PCIe
  std::vector<hipEvent_t> events;
  hipStream_t image_stream = nullptr; // the kernels of chunks whose image is copied when they are issued, if the call also has early copies (below)
  struct Release {
    Context** c; hipStream_t* cs; hipStream_t* is; std::vector<hipEvent_t>* ev;
    ~Release() {
      if (*c == nullptr) return;
      if (*cs) (void)hipStreamSynchronize(*cs);
      if (*is) (void)hipStreamSynchronize(*is);
      (void)hipStreamSynchronize((*c)->stream);
      for (hipEvent_t e : *ev) if (e) (*c)->put_event(e);
      (*c)->reset_staging();
      Context::release(*c);
    }
  } rel{&ctx, &copy_stream, &image_stream, &events};
  std::vector<void*> scratch;
  // Pinned memory of the call itself: where small tables are staged and where the device writes what the host reads back (non-NULL
  // counts, Snappy verdicts, index checks). A copy to or from PAGEABLE memory makes the issuing thread wait for the stream — one such
  // copy per chunk and the row groups of a call ran one after the other however they were queued.
  struct PinnedBump {
    std::vector<void*> blocks;
    uint8_t* cur = nullptr;
    size_t left = 0;
    void* take(size_t bytes) {
      bytes = (bytes + 63) & ~(size_t)63;
      if (bytes > left) {
        const size_t want = std::max(bytes, (size_t)2 << 20);
        void* p = pinned_pool_alloc(want);
        blocks.push_back(p);
        cur = (uint8_t*)p; left = want;
      }
      void* r = cur;
      cur += bytes; left -= bytes;
      return r;
    }
  } pinned;
  struct FreeScratch {
    Context** c; hipStream_t* cs; hipStream_t* is; std::vector<void*>* v; PinnedBump* pb;
    ~FreeScratch() {
      if (*c != nullptr) {
        if (*cs) (void)hipStreamSynchronize(*cs);
        if (*is) (void)hipStreamSynchronize(*is);
        (void)hipStreamSynchronize((*c)->stream);
        for (void* p : *v) (*c)->dev_free(p);
      }
      for (void* p : pb->blocks) pinned_pool_free(p);
    }
  } fs{&ctx, &copy_stream, &image_stream, &scratch, &pinned};
  static const bool prof = std::getenv("FDB_PROFILE_PARQUET") != nullptr;  // (tuning aid: the host part's phases on stderr)

  // ---- (1) page headers of every chunk (host-only: malformed / unsupported chunks are refused before any device call; the first
  // failure in row-group and column order is reported) ---------------------------------------------------------------------------
  std::vector<std::pair<int32_t, int32_t>> chunk_at;  // flat chunk number → (row group, column)
  chunk_at.reserve(all_chunks);
  size_t chunk_bytes = 0;
  for (int32_t g = 0; g < n_groups; g++)
    for (int32_t i = 0; i < G[(size_t)g].n_chunks; i++) { chunk_at.emplace_back(g, i); chunk_bytes += (size_t)std::max<int64_t>(G[(size_t)g].chunks[i].n_bytes, 0); }
  auto plan_one = [&](size_t k) {
    Group& R = G[(size_t)chunk_at[k].first];
    const size_t i = (size_t)chunk_at[k].second;
    plan_chunk(R.chunks[i], R.n_rows, &R.parsed[i], &R.jobs[i]);
  };
  const bool threads = chunk_bytes >= ((size_t)1 << 20) && all_chunks > 1;
  if (threads && n_groups > 1) HostPool::get().parallel_for(all_chunks, plan_one);
  else for (size_t k = 0; k < all_chunks; k++) plan_one(k);
  const auto tp0 = std::chrono::steady_clock::now();

  // ---- (2) chunks that need no image (uncompressed pages of fixed-width or dictionary-encoded values: the device decodes from the
  // chunk's own bytes) start crossing PCIe NOW, while the host inflates and parses — a row group's 1.3 ms of header walking used to sit
  // in front of its 1.5 ms of copies (round 5; `profiles/README.md`). plan_chunk has already refused structurally damaged chunks; if no
  // device can be had here (a CPU-only box: the host part still has to say what is wrong with the input) the copies happen later.
  bool any_rows = false;
  for (const Group& R : G) any_rows = any_rows || R.n_rows > 0;
  std::vector<std::pair<int32_t, int32_t>> early_order;  // the early copies, in the order they were queued
  if (any_rows && hipSetDevice(device) == hipSuccess) {
    try { ctx = Context::acquire(device); } catch (...) { ctx = nullptr; }
    if (ctx != nullptr) {
      copy_stream = std::getenv("FDB_PQ_EARLY_ON_MAIN") != nullptr ? ctx->stream : ctx->aux_stream(0);  // ($FDB_PQ_EARLY_ON_MAIN: A/B aid)
      for (int32_t g = 0; g < n_groups; g++) {
        Group& R = G[(size_t)g];
        if (R.n_rows <= 0) continue;
        for (int32_t i = 0; i < R.n_chunks; i++) {
          if (!R.parsed[(size_t)i].image.empty() || R.chunks[i].n_bytes <= 0) continue;
          early_order.emplace_back(g, i);
          R.early[(size_t)i] = (uint8_t*)ctx->dev_alloc((size_t)R.chunks[i].n_bytes + 64);
          scratch.push_back(R.early[(size_t)i]);
          hip_check(hipMemcpyAsync(R.early[(size_t)i], R.chunks[i].data, (size_t)R.chunks[i].n_bytes, hipMemcpyHostToDevice, copy_stream), "hipMemcpyAsync(parquet chunk, early)");
          R.copied[(size_t)i] = ctx->get_event();
          events.push_back(R.copied[(size_t)i]);
          hip_check(hipEventRecord(R.copied[(size_t)i], copy_stream), "hipEventRecord(parquet chunk)");
        }
      }
    }
  } else {
    (void)hipGetLastError();
  }

  // ---- (3) host work, all row groups side by side: EVERY compressed page inflated as a task of its own (pages are independent: a chunk
  // of twenty 1 MiB pages used to be one thread's job), every chunk parsed as one task. One task list in row-group order — a row
  // group's pages, then its chunks — handed out in that order: a parse task whose chunk still has pages in other threads' hands waits for
  // them (they were handed out earlier, so they are running, not queued). The walk of the RLE / bit-packed run headers is a serial
  // chain per page and chunk; it is the host part's floor.
  struct Task { int32_t g, i, job; };  // job ≥ 0: inflate page `job` of the chunk; −1: parse the chunk
  std::vector<Task> tasks;
  size_t job_bytes = 0, n_jobs = 0;
  std::vector<std::unique_ptr<std::atomic<int>[]>> pages_left((size_t)n_groups);
  for (int32_t g = 0; g < n_groups; g++) {
    Group& R = G[(size_t)g];
    pages_left[(size_t)g].reset(new std::atomic<int>[(size_t)R.n_chunks]);
    for (int32_t i = 0; i < R.n_chunks; i++) {
      pages_left[(size_t)g][(size_t)i].store((int)R.jobs[(size_t)i].size());
      for (size_t j = 0; j < R.jobs[(size_t)i].size(); j++) { tasks.push_back(Task{g, i, (int32_t)j}); job_bytes += R.jobs[(size_t)i][j].body_len; n_jobs++; }
    }
    for (int32_t i = 0; i < R.n_chunks; i++) tasks.push_back(Task{g, i, -1});
    R.parse_left.store(R.n_chunks);
  }
  std::atomic<bool> failed{false};
  std::mutex ready_mu;
  std::condition_variable ready_cv;
  std::deque<std::pair<int32_t, int32_t>> ready;  // chunks whose parse is complete, in the order they came out
  auto run_task = [&](size_t k) {
    const Task& T = tasks[k];
    Group& R = G[(size_t)T.g];
    std::atomic<int>& left = pages_left[(size_t)T.g][(size_t)T.i];
    if (T.job >= 0) {
      struct Done { std::atomic<int>& l; ~Done() { l.fetch_sub(1, std::memory_order_release); } } done{left};
      run_inflate(R.jobs[(size_t)T.i][(size_t)T.job]);
      return;
    }
    struct Done { std::atomic<int>& l; ~Done() { l.fetch_sub(1, std::memory_order_release); } } done{R.parse_left};
    try {
      while (left.load(std::memory_order_acquire) > 0) std::this_thread::yield();
      if (failed.load(std::memory_order_relaxed)) return;  // (some page or chunk before this one is damaged: its error is the one reported)
      const auto a = std::chrono::steady_clock::now();
      parse_chunk(R.chunks[T.i], R.n_rows, &R.parsed[(size_t)T.i]);
      R.parse_us[(size_t)T.i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
      {
        std::lock_guard<std::mutex> lk(ready_mu);
        if (R.early[(size_t)T.i] != nullptr) R.parsed_ok[(size_t)T.i] = 1; else ready.emplace_back(T.g, T.i);
      }
      ready_cv.notify_one();
    } catch (...) { failed.store(true); { std::lock_guard<std::mutex> lk(ready_mu); } ready_cv.notify_all(); throw; }
  };
  auto run_tasks = [&] {
    auto guarded = [&](size_t k) { try { run_task(k); } catch (...) { failed.store(true); { std::lock_guard<std::mutex> lk(ready_mu); } ready_cv.notify_all(); throw; } };
    if (threads || job_bytes >= ((size_t)1 << 20)) HostPool::get().parallel_for(tasks.size(), guarded);
    else for (size_t k = 0; k < tasks.size(); k++) guarded(k);
  };
  // (several row groups: the task list runs behind a thread of its own and this one issues device work as the chunks come out)
  std::exception_ptr host_error;
  std::thread runner;
  struct Join { std::thread* t; ~Join() { if (t->joinable()) t->join(); } } join{&runner};
  std::atomic<bool> host_done{false};
  std::chrono::steady_clock::time_point t_host1 = t_host0;
  auto rethrow_host_error = [&] {
    if (runner.joinable()) runner.join();
    if (host_error) std::rethrow_exception(host_error);
  };

  // ---- (4) device part, chunk by chunk in the order the parses complete: a row group's 40 MB of PLAIN values (parsed in 50 µs) must not
  // wait for the 1–2 ms run-header walk of the dictionary column next to it before it may cross PCIe ----------------------------------
  hipStream_t stream = nullptr;
  std::list<std::vector<FdbSnappyPage>> snappy_tables;                                       // device-inflated pages: per chunk, the launch's page table …
  std::list<std::tuple<const uint32_t*, size_t, int32_t, int32_t>> snappy_status;            // … and where its verdicts land (pinned host copy, pages, row group, chunk)
  uint32_t* d_flags = nullptr;
  uint32_t* h_flags = nullptr;
  int32_t n_flags = 0;
  auto prepare = [&] {  // the batches and their arenas (sizes follow from the chunks' types alone), the index checks' flags
    hip_check(hipSetDevice(device), "hipSetDevice");
    if (ctx == nullptr) ctx = Context::acquire(device);
    stream = ctx->stream;
    d_flags = (uint32_t*)ctx->dev_alloc(all_chunks * 4 + 64);
    scratch.push_back(d_flags);
    hip_check(hipMemsetAsync(d_flags, 0, all_chunks * 4, stream), "hipMemsetAsync");
    h_flags = (uint32_t*)pinned.take(all_chunks * 4);
    std::memset(h_flags, 0, all_chunks * 4);
    for (Group& R : G) {
      R.h_totals = (unsigned long long*)pinned.take((size_t)R.n_chunks * 8);
      std::memset(R.h_totals, 0, (size_t)R.n_chunks * 8);
      R.b.reset(new DeviceBatch());
      R.b->device = device;
      R.b->rows = R.n_rows;
      size_t total = 0;
      for (int32_t i = 0; i < R.n_chunks; i++) {
        const fdb_parquet_chunk& c = R.chunks[i];
        const size_t w = c.physical_type == 6 ? 4 : 8;
        R.pieces[(size_t)i].val_off = total;
        total += align_up((size_t)R.n_rows * w + kTailPad, 256);
        R.pieces[(size_t)i].bit_off = total;
        if (c.optional) total += align_up((size_t)((R.n_rows + 31) / 32) * 4 + kTailPad, 256);
      }
      if (R.n_rows > 0) { R.b->arena = device_pool_alloc(device, std::max<size_t>(total, 256)); R.b->arena_bytes = std::max<size_t>(total, 256); }
    }
  };
  auto issue_chunk = [&](int32_t g, int32_t i, hipStream_t stream) {  // (`stream`: the one this chunk's kernels and tables go to)
    Group& R = G[(size_t)g];
    const int64_t n_rows = R.n_rows;
    if (n_rows <= 0) return;
    const fdb_parquet_chunk* chunks = R.chunks;
    DeviceBatch* b = R.b.get();
    const int64_t n_words = (n_rows + 31) / 32;
    {

      const fdb_parquet_chunk& c = chunks[i];
      const ParsedChunk& P = R.parsed[(size_t)i];
      // the chunk's bytes as they are (or the image of its decompressed pages), padded so that 8-byte windows at the very end stay
      // inside the allocation
      const uint8_t* src = P.image.empty() ? c.data : P.image.data();
      const size_t src_bytes = P.image.empty() ? (size_t)c.n_bytes : P.image.size();
      const bool copied_early = P.image.empty() && R.early[(size_t)i] != nullptr;
      if (copied_early) hip_check(hipStreamWaitEvent(stream, R.copied[(size_t)i], 0), "hipStreamWaitEvent(parquet chunk)");  // this chunk's kernels wait for ITS copy only
      uint8_t* d_chunk = copied_early ? R.early[(size_t)i] : (uint8_t*)ctx->dev_alloc(src_bytes + 64);
      if (!copied_early) scratch.push_back(d_chunk);
      // the chunk's bytes (or its image) go to the copy queue; the kernels' stream waits for an event behind them (fence), so the link
      // stays busy while earlier chunks' kernels run
      hipStream_t cs = copy_stream != nullptr ? copy_stream : stream;
      bool unfenced = false;
      auto h2d = [&](void* d, const void* host, size_t bytes, const char* what) {
        hip_check(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, cs), what);
        unfenced = true;
      };
      auto fence = [&] {
        if (unfenced && cs != stream) {
          hipEvent_t e = ctx->get_event();
          events.push_back(e);
          hip_check(hipEventRecord(e, cs), "hipEventRecord(parquet copies)");
          hip_check(hipStreamWaitEvent(stream, e, 0), "hipStreamWaitEvent(parquet copies)");
        }
        unfenced = false;
      };
      auto to_device = [&](const void* host, size_t bytes, bool host_is_pinned = false) -> void* {
        void* d = ctx->dev_alloc(std::max<size_t>(bytes, 16));
        scratch.push_back(d);
        if (bytes == 0) return d;
        if (!host_is_pinned) { void* st = pinned.take(bytes); std::memcpy(st, host, bytes); host = st; }  // (small tables: through the call's pinned block)
        // (the tables of a chunk that was copied early ride on the kernels' stream: on the copy queue they would wait behind every later
        // row group's early copy; a chunk whose image goes to the copy queue now sends its tables along)
        if (copied_early) hip_check(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(parquet tables)");
        else h2d(d, host, bytes, "hipMemcpyAsync(parquet tables)");
        return d;
      };
      if (P.dev_pages.empty()) {
        if (src_bytes && !copied_early) h2d(d_chunk, src, src_bytes, "hipMemcpyAsync(parquet chunk)");
      } else {
        // the host's part of the image, stretch by stretch (plus whatever the parse appended behind the pages), then the compressed bytes
        // of the pages the device inflates — one copy from the caller's chunk — and one launch that puts them where the image has holes
        size_t spans_end = 0;
        for (const auto& sp : P.host_spans) {
          const size_t e = std::min(sp.second, src_bytes);
          if (e > sp.first) h2d(d_chunk + sp.first, src + sp.first, e - sp.first, "hipMemcpyAsync(parquet chunk)");
        }
        for (const ParsedChunk::DevPage& q : P.dev_pages) spans_end = std::max(spans_end, q.at + q.len + 8);
        for (const auto& sp : P.host_spans) spans_end = std::max(spans_end, sp.second);
        if (src_bytes > spans_end) h2d(d_chunk + spans_end, src + spans_end, src_bytes - spans_end, "hipMemcpyAsync(parquet chunk)");
        size_t lo = (size_t)-1, hi = 0;
        for (const ParsedChunk::DevPage& q : P.dev_pages) { lo = std::min(lo, q.raw_off); hi = std::max(hi, q.raw_off + q.comp); }
        uint8_t* d_raw = (uint8_t*)ctx->dev_alloc(hi - lo + 64);
        scratch.push_back(d_raw);
        h2d(d_raw, c.data + lo, hi - lo, "hipMemcpyAsync(compressed pages)");
        std::vector<FdbSnappyPage> table;
        for (const ParsedChunk::DevPage& q : P.dev_pages) table.push_back(FdbSnappyPage{(uint64_t)(q.raw_off - lo), (uint64_t)q.at, (uint32_t)q.comp, (uint32_t)q.len});
        snappy_tables.push_back(std::move(table));  // (kept alive until the copy below has read it)
        const std::vector<FdbSnappyPage>& T = snappy_tables.back();
        const FdbSnappyPage* d_table = (const FdbSnappyPage*)to_device(T.data(), T.size() * sizeof(FdbSnappyPage));
        uint32_t* d_status = (uint32_t*)ctx->dev_alloc(T.size() * 4 + 16);
        scratch.push_back(d_status);
        fence();
        hip_check(fdb_launch_snappy_decode(d_raw, d_table, (int32_t)T.size(), d_chunk, d_status, stream), "snappy decode");
        uint32_t* h_status = (uint32_t*)pinned.take(T.size() * 4);
        std::memset(h_status, 0xFF, T.size() * 4);  // (a verdict that never arrives is not "ok")
        snappy_status.emplace_back(h_status, T.size(), g, i);
        hip_check(hipMemcpyAsync(h_status, d_status, T.size() * 4, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(snappy status)");
      }
      uint32_t* d_valid = nullptr;
      uint32_t* d_prefix = nullptr;
      if (c.optional) {
        if (P.def_runs.empty()) throw Error(FDB_ERR_INVALID, "parquet: optional column without definition levels");
        const FdbPqRun* d_runs = (const FdbPqRun*)to_device(P.def_runs.data(), P.def_runs.size() * sizeof(FdbPqRun), P.def_runs.pinned());
        d_valid = (uint32_t*)((unsigned char*)b->arena + R.pieces[(size_t)i].bit_off);
        d_prefix = (uint32_t*)ctx->dev_alloc((size_t)(n_words + 4 + n_words / 1024 + 8) * 4);  // (+ the scan's per-1024 sums)
        scratch.push_back(d_prefix);
        R.d_totals[(size_t)i] = (unsigned long long*)ctx->dev_alloc(64);
        scratch.push_back(R.d_totals[(size_t)i]);
        fence();
        hip_check(fdb_launch_pq_validity(d_chunk, d_runs, (int32_t)P.def_runs.size(), n_rows, d_valid, d_prefix, stream), "parquet validity");
        hip_check(fdb_launch_exclusive_scan(d_prefix, n_words, d_prefix + n_words + 4, R.d_totals[(size_t)i], stream), "parquet rank scan");
      }
      void* d_out = (unsigned char*)b->arena + R.pieces[(size_t)i].val_off;
      if (c.physical_type == 6) {
        const FdbPqRun* d_idx = (const FdbPqRun*)to_device(P.idx_runs.data(), P.idx_runs.size() * sizeof(FdbPqRun), P.idx_runs.pinned());
        if (P.non_null > 0 && P.idx_runs.empty()) throw Error(FDB_ERR_INVALID, "parquet: values without index runs");
        if (P.idx_runs.empty()) hip_check(hipMemsetAsync(d_out, 0, (size_t)n_rows * 4, stream), "hipMemsetAsync");
        else { fence(); hip_check(fdb_launch_pq_decode(1, d_chunk, d_valid, d_prefix, nullptr, 0, d_idx, (int32_t)P.idx_runs.size(), n_rows, d_out, stream), "parquet decode"); }
        if (P.non_null > 0) {  // (bit width 0 included: index 0 of an EMPTY dictionary is out of range too)
          // indices are validated like any imported dictionary column's (a corrupt page must not become an out-of-bounds LUT read);
          // a NULL row's index is whatever the decode left there, so the check reads the bitmap whenever the column is optional
          const size_t dict_len = P.dict ? P.dict->values.size() : 0;
          R.flag_at[(size_t)i] = n_flags;
          hip_check(fdb_launch_validate_indices((const uint32_t*)d_out, (const uint8_t*)d_valid, n_rows, (uint32_t)std::min<size_t>(dict_len, 0xFFFFFFFFu), d_flags + n_flags, stream), "index check");
          n_flags++;
        }
      } else if (c.physical_type == PT_BOOLEAN) {
        // bits through the run tables like dictionary indices; the column is held as int64 1 (false) / 2 (true) like every bool column
        const FdbPqRun* d_idx = (const FdbPqRun*)to_device(P.idx_runs.data(), P.idx_runs.size() * sizeof(FdbPqRun), P.idx_runs.pinned());
        if (P.non_null > 0 && P.idx_runs.empty()) throw Error(FDB_ERR_INVALID, "parquet: values without runs");
        if (P.idx_runs.empty()) hip_check(hipMemsetAsync(d_out, 0, (size_t)n_rows * 8, stream), "hipMemsetAsync");
        else { fence(); hip_check(fdb_launch_pq_decode(2, d_chunk, d_valid, d_prefix, nullptr, 0, d_idx, (int32_t)P.idx_runs.size(), n_rows, d_out, stream), "parquet decode"); }
      } else {
        if (!P.delta_pages.empty()) {
          // DELTA_BINARY_PACKED: the non-NULL values are decoded densely (rank order) — straight into the column when it is
          // required (rank = row), else into a scratch array that the row kernel then reads like one PLAIN page
          const FdbPqDeltaPage* d_dp = (const FdbPqDeltaPage*)to_device(P.delta_pages.data(), P.delta_pages.size() * sizeof(FdbPqDeltaPage));
          const FdbPqDeltaMini* d_dm = (const FdbPqDeltaMini*)to_device(P.delta_minis.data(), P.delta_minis.size() * sizeof(FdbPqDeltaMini), P.delta_minis.pinned());
          unsigned long long* dense = (unsigned long long*)d_out;
          if (c.optional) { dense = (unsigned long long*)ctx->dev_alloc((size_t)P.non_null * 8 + 64); scratch.push_back(dense); }
          else if (P.non_null != n_rows) throw Error(FDB_ERR_INVALID, "parquet: required column with fewer values than rows");
          fence();
        hip_check(fdb_launch_pq_delta(d_chunk, d_dp, (int32_t)P.delta_pages.size(), d_dm, dense, stream), "parquet delta decode");
          if (c.optional) {
            static const FdbPqPlainPage kWhole{0, 0};  // (static: the copy below is asynchronous)
            const FdbPqPlainPage* d_one = (const FdbPqPlainPage*)to_device(&kWhole, sizeof(kWhole));
            fence();
            hip_check(fdb_launch_pq_decode(0, (const uint8_t*)dense, d_valid, d_prefix, d_one, 1, nullptr, 0, n_rows, d_out, stream), "parquet decode");
          }
        } else {
          const FdbPqPlainPage* d_pages = (const FdbPqPlainPage*)to_device(P.plain_pages.data(), P.plain_pages.size() * sizeof(FdbPqPlainPage));
          if (P.plain_pages.empty()) hip_check(hipMemsetAsync(d_out, 0, (size_t)n_rows * 8, stream), "hipMemsetAsync");
          else { fence(); hip_check(fdb_launch_pq_decode(0, d_chunk, d_valid, d_prefix, d_pages, (int32_t)P.plain_pages.size(), nullptr, 0, n_rows, d_out, stream), "parquet decode"); }
        }
      }
      if (R.d_totals[(size_t)i] != nullptr)
        hip_check(hipMemcpyAsync(&R.h_totals[(size_t)i], R.d_totals[(size_t)i], 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(non-null count)");
    }
  };
  if (ctx != nullptr) prepare();
  // the host part: behind a thread of its own when there is a device to feed meanwhile, else right here (a CPU-only box gets as far as
  // the parse errors, then the device error)
  if (ctx != nullptr && all_chunks > 1) {
    runner = std::thread([&] {
      try { run_tasks(); } catch (...) { host_error = std::current_exception(); }
      t_host1 = std::chrono::steady_clock::now();
      { std::lock_guard<std::mutex> lk(ready_mu); host_done.store(true, std::memory_order_release); }
      ready_cv.notify_all();
    });
  } else {
    run_tasks();
    t_host1 = std::chrono::steady_clock::now();
    host_done.store(true);
    if (ctx == nullptr) prepare();
  }
  // Which chunk next: the stream runs a chunk's kernels when ITS copy has arrived, and in the order they were issued — so chunks that were
  // copied early are issued in the order of the copy queue (issued as they are parsed, a 40 MB PLAIN chunk whose copy is the call's last
  // would sit in front of every dictionary chunk's kernels), and chunks whose image is copied when they are issued go in the order their
  // parses complete — to a stream of their own when the call has both kinds (their copies queue up behind every early one).
  if (ctx != nullptr && !early_order.empty() && early_order.size() < all_chunks) {
    image_stream = ctx->aux_stream(1);
    hipEvent_t e = ctx->get_event();
    events.push_back(e);
    hip_check(hipEventRecord(e, stream), "hipEventRecord(parquet flags)");  // (the index checks' flags are zeroed on the main stream)
    hip_check(hipStreamWaitEvent(image_stream, e, 0), "hipStreamWaitEvent(parquet flags)");
  }
  size_t next_early = 0;
  for (size_t issued = 0; issued < all_chunks; issued++) {
    std::pair<int32_t, int32_t> k;
    bool is_early = false;
    {
      std::unique_lock<std::mutex> lk(ready_mu);
      auto early_ready = [&] { return next_early < early_order.size() && G[(size_t)early_order[next_early].first].parsed_ok[(size_t)early_order[next_early].second] != 0; };
      ready_cv.wait(lk, [&] { return early_ready() || !ready.empty() || failed.load() || host_done.load(); });
      if (failed.load()) break;
      if (early_ready()) { k = early_order[next_early++]; is_early = true; }
      else if (!ready.empty()) { k = ready.front(); ready.pop_front(); }
      else break;  // (host_done with nothing ready: a task failed before it could say so)
    }
    issue_chunk(k.first, k.second, is_early || image_stream == nullptr ? stream : image_stream);
  }
  rethrow_host_error();  // (joins the host threads' runner; a damaged chunk's error wins over anything the device may say)
  if (prof) {
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "[fdb] parquet host part (%d row groups): headers %.0f us, inflate %zu pages (%zu bytes) + parse %.0f us (per chunk:", n_groups, us(t_host0, tp0), n_jobs, job_bytes, us(tp0, t_host1));
    for (const Group& R : G) { for (int32_t i = 0; i < R.n_chunks; i++) std::fprintf(stderr, " %s %.0f", R.chunks[i].name ? R.chunks[i].name : "?", R.parse_us[(size_t)i]); std::fprintf(stderr, " |"); }
    std::fprintf(stderr, ")\n");
  }
  if (stream != nullptr) {
    if (image_stream != nullptr) {
      hipEvent_t e = ctx->get_event();
      events.push_back(e);
      hip_check(hipEventRecord(e, image_stream), "hipEventRecord(parquet decode)");
      hip_check(hipStreamWaitEvent(stream, e, 0), "hipStreamWaitEvent(parquet decode)");
    }
    if (n_flags > 0) hip_check(hipMemcpyAsync(h_flags, d_flags, (size_t)n_flags * 4, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(index checks)");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(parquet decode)");
  }
  for (const auto& st : snappy_status)
    for (size_t q = 0; q < std::get<1>(st); q++)
      if (std::get<0>(st)[q] != 0) {
        const char* nm = G[(size_t)std::get<2>(st)].chunks[std::get<3>(st)].name;
        throw Error(FDB_ERR_INVALID, std::string("parquet: corrupt Snappy page in column ") + (nm ? nm : "?"));
      }

  std::vector<std::unique_ptr<DeviceBatch>> out;
  int64_t fb = 0, ob = 0;
  for (int32_t g = 0; g < n_groups; g++) {
    Group& R = G[(size_t)g];
    if (!R.b) {  // (only when nothing had rows and no device was needed: an empty batch still has its columns)
      R.b.reset(new DeviceBatch());
      R.b->device = device;
      R.b->rows = R.n_rows;
    }
    DeviceBatch* b = R.b.get();
    const int64_t n_rows = R.n_rows;
    for (int32_t i = 0; i < R.n_chunks; i++) {
      const fdb_parquet_chunk& c = R.chunks[i];
      const ParsedChunk& P = R.parsed[(size_t)i];
      DevColumn d;
      d.name = c.name ? c.name : "";
      d.length = n_rows;
      if (c.physical_type == 6) { d.kind = ColKind::DICT; d.format = "I"; d.dict = P.dict ? P.dict : make_dictionary({}, c.utf8 ? "u" : "z"); }
      else if (c.physical_type == PT_INT64 && c.utf8) { d.kind = ColKind::U64; d.format = "L"; }  // logical type Int(64, unsigned) (convert.go:76-82)
      else if (c.physical_type == PT_INT64) { d.kind = ColKind::I64; d.format = "l"; }
      else if (c.physical_type == PT_BOOLEAN) { d.kind = ColKind::BOOL; d.format = "b"; }
      else { d.kind = ColKind::F64; d.format = "g"; }
      const size_t w = c.physical_type == 6 ? 4 : 8;
      if (n_rows > 0) d.d_values = (unsigned char*)b->arena + R.pieces[(size_t)i].val_off;
      d.value_bytes = d.kind == ColKind::BOOL ? (n_rows + 7) / 8 : n_rows * (int64_t)w;
      if (c.optional && n_rows > 0) {
        if ((int64_t)R.h_totals[(size_t)i] != P.non_null) throw Error(FDB_ERR_INVALID, "parquet: definition levels and value counts disagree in column " + d.name);
        d.null_count = n_rows - P.non_null;
        if (d.null_count > 0) { d.d_validity = (uint8_t*)b->arena + R.pieces[(size_t)i].bit_off; d.validity_bytes = (n_rows + 7) / 8; }
      }
      if (R.flag_at[(size_t)i] >= 0 && h_flags != nullptr && h_flags[(size_t)R.flag_at[(size_t)i]] != 0) throw Error(FDB_ERR_INVALID, "parquet: dictionary index out of range in column " + d.name);
      b->payload_bytes += d.value_bytes + d.validity_bytes;
      b->cols.push_back(std::move(d));
      fb += c.n_bytes;
    }
    ob += b->payload_bytes;
    out.push_back(std::move(R.b));
  }
  {  // measurement hooks (fdb_parquet_stats): host = header walk + inflate + dictionary work (until the last chunk is parsed),
     // device = what is left of the call after that: copies + pq_* kernels + the wait
    const auto t_end = std::chrono::steady_clock::now();
    g_pq_calls++;
    g_pq_host_us += std::chrono::duration_cast<std::chrono::microseconds>(t_host1 - t_host0).count();
    g_pq_device_us += std::chrono::duration_cast<std::chrono::microseconds>(t_end - t_host1).count();
    g_pq_file_bytes += fb;
    g_pq_out_bytes += ob;
  }
  return out;
}

std::unique_ptr<DeviceBatch> batch_from_parquet(const fdb_parquet_chunk* chunks, int32_t n_chunks, int64_t n_rows, int device) {
  if (chunks == nullptr || n_chunks <= 0 || n_rows < 0) throw Error(FDB_ERR_INVALID, "parquet: no column chunks");
  const fdb_parquet_row_group g{chunks, n_chunks, n_rows};
  return std::move(batches_from_parquet(&g, 1, device)[0]);
}

}  // namespace fdb
