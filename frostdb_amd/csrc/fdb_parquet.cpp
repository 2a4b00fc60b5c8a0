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
// First slice (what FrostDB's default layouts produce, dynparquet/schema.go:508-560): flat schemas; INT64 / DOUBLE columns with
// PLAIN data pages; BYTE_ARRAY columns with a PLAIN dictionary page + RLE_DICTIONARY data pages (→ dictionary<uint32, binary>,
// pqarrow/convert/convert.go:64-70); required or optional (max definition level 1); data pages V1 and V2; codec UNCOMPRESSED.
// Anything else (DELTA_* encodings, compressed pages, dictionary fallback to PLAIN, nested columns) is FDB_ERR_UNSUPPORTED.
#include <algorithm>
#include <cstring>

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
  void skip(int type, int depth = 0) {
    if (depth > 16) throw Error(FDB_ERR_INVALID, "parquet: page header nested too deeply");
    switch (type) {
      case 1: case 2: return;                       // bool (value lives in the field header)
      case 3: need(1); p += 1; return;              // byte
      case 4: case 5: case 6: (void)zigzag(); return;
      case 7: need(8); p += 8; return;              // double
      case 8: { const uint64_t n = varint(); need((size_t)n); p += n; return; }  // binary
      case 9: case 10: {                            // list / set
        need(1);
        const uint8_t h = *p++;
        uint64_t n = h >> 4;
        if (n == 15) n = varint();
        for (uint64_t i = 0; i < n; i++) skip(h & 0x0F, depth + 1);
        return;
      }
      case 11: {                                    // map
        const uint64_t n = varint();
        if (n == 0) return;
        need(1);
        const uint8_t kv = *p++;
        for (uint64_t i = 0; i < n; i++) { skip(kv >> 4, depth + 1); skip(kv & 0x0F, depth + 1); }
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
enum { ENC_PLAIN = 0, ENC_PLAIN_DICTIONARY = 2, ENC_RLE = 3, ENC_RLE_DICTIONARY = 8 };

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

// Walks the run HEADERS of an RLE / bit-packed hybrid stream of `n_values` values of `bw` bits in chunk[off, off + len) and
// appends one FdbPqRun per run, numbered from `first` on. Bit-packed payloads are skipped, not read — except when `count_ones`
// is given (definition levels, bw = 1): then the set bits are counted (popcount over the payload bytes).
void scan_runs(const uint8_t* chunk, size_t off, size_t len, int bw, int64_t n_values, int64_t first, std::vector<FdbPqRun>* runs, int64_t* count_ones) {
  Thrift t{chunk + off, chunk + off + len};
  int64_t done = 0;
  const int vbytes = (bw + 7) / 8;
  while (done < n_values) {
    const uint64_t h = t.varint();
    FdbPqRun r;
    r.start = first + done;
    if (h & 1) {
      const uint64_t groups = h >> 1;
      if (groups == 0) throw Error(FDB_ERR_INVALID, "parquet: empty bit-packed run");
      const uint64_t bytes = groups * (uint64_t)bw;
      int64_t count = (int64_t)(groups * 8);
      if (count > n_values - done) count = n_values - done;  // (the last group is padded)
      // the final run of a stream may be cut short (writers need not pad it to whole groups)
      const size_t have = (size_t)(t.end - t.p);
      if (have < bytes && (int64_t)(have * 8 / (size_t)std::max(bw, 1)) < count) throw Error(FDB_ERR_INVALID, "parquet: bit-packed run past the end of its page");
      r.kind = 1; r.bit_width = (uint32_t)bw; r.payload = (uint64_t)(t.p - chunk) * 8;
      if (count_ones != nullptr) {
        const int64_t full = count / 8;
        for (int64_t i = 0; i < full; i++) *count_ones += __builtin_popcount(t.p[i]);
        if (count % 8) *count_ones += __builtin_popcount(t.p[full] & ((1u << (count % 8)) - 1u));
      }
      t.p += std::min<size_t>(have, (size_t)bytes);
      done += count;
    } else {
      int64_t count = (int64_t)(h >> 1);
      if (count == 0) throw Error(FDB_ERR_INVALID, "parquet: empty RLE run");
      t.need((size_t)vbytes);
      uint64_t v = 0;
      for (int i = 0; i < vbytes; i++) v |= (uint64_t)t.p[i] << (8 * i);
      t.p += vbytes;
      if (count > n_values - done) count = n_values - done;
      r.kind = 0; r.bit_width = 0; r.payload = v;
      if (count_ones != nullptr && (v & 1)) *count_ones += count;
      done += count;
    }
    runs->push_back(r);
  }
}

struct ParsedChunk {
  std::shared_ptr<HostDict> dict;          // BYTE_ARRAY columns
  std::vector<FdbPqRun> def_runs;          // optional columns: one entry per run, row-numbered
  std::vector<FdbPqRun> idx_runs;          // dictionary-encoded columns: rank-numbered
  std::vector<FdbPqPlainPage> plain_pages; // PLAIN fixed-width columns
  int64_t non_null = 0;
  uint32_t max_index_bits = 0;
};

ParsedChunk parse_chunk(const fdb_parquet_chunk& c, int64_t n_rows) {
  if (c.data == nullptr || c.n_bytes <= 0) throw Error(FDB_ERR_INVALID, std::string("parquet: empty column chunk for ") + (c.name ? c.name : "?"));
  if (c.optional != 0 && c.optional != 1) throw Error(FDB_ERR_UNSUPPORTED, "parquet: nested / repeated columns are not supported (max definition level > 1)");
  const bool is_bytes = c.physical_type == 6, is_fixed8 = c.physical_type == 2 || c.physical_type == 5;
  if (!is_bytes && !is_fixed8) throw Error(FDB_ERR_UNSUPPORTED, "parquet: only INT64, DOUBLE and BYTE_ARRAY columns are decoded on the device");
  ParsedChunk out;
  const uint8_t* base = c.data;
  Thrift t{base, base + c.n_bytes};
  int64_t rows_done = 0, rank_done = 0;
  bool have_dict = false;
  while (t.p < t.end && rows_done < n_rows) {
    const PageHeader h = read_page_header(t);
    if (h.compressed != h.uncompressed) throw Error(FDB_ERR_UNSUPPORTED, "parquet: compressed pages are not supported on the device path (codec must be UNCOMPRESSED)");
    if (h.compressed < 0 || (size_t)(t.end - t.p) < (size_t)h.compressed) throw Error(FDB_ERR_INVALID, "parquet: page runs past the end of the column chunk");
    const uint8_t* body = t.p;
    const size_t body_off = (size_t)(body - base), body_len = (size_t)h.compressed;
    t.p += body_len;
    if (h.type == PQ_DICTIONARY_PAGE) {
      if (!is_bytes) throw Error(FDB_ERR_UNSUPPORTED, "parquet: dictionary-encoded numeric columns are not supported on the device path");
      if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICTIONARY) throw Error(FDB_ERR_UNSUPPORTED, "parquet: dictionary page encoding");
      std::vector<std::string> values;
      values.reserve((size_t)std::max(h.num_values, 0));
      size_t o = 0;
      for (int32_t i = 0; i < h.num_values; i++) {
        if (o + 4 > body_len) throw Error(FDB_ERR_INVALID, "parquet: dictionary page truncated");
        uint32_t len; std::memcpy(&len, body + o, 4); o += 4;
        if (o + len > body_len) throw Error(FDB_ERR_INVALID, "parquet: dictionary page truncated");
        values.emplace_back((const char*)body + o, len); o += len;
      }
      out.dict = make_dictionary(std::move(values), c.utf8 ? "u" : "z");
      have_dict = true;
      continue;
    }
    if (h.type != PQ_DATA_PAGE && h.type != PQ_DATA_PAGE_V2) continue;  // (index pages etc.)
    if (h.num_values < 0 || rows_done + h.num_values > n_rows) throw Error(FDB_ERR_INVALID, "parquet: pages hold more values than the row group has rows");
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
    if (is_fixed8) {
      if (h.encoding != ENC_PLAIN) throw Error(FDB_ERR_UNSUPPORTED, "parquet: INT64 / DOUBLE pages must be PLAIN (DELTA_BINARY_PACKED is not decoded on the device yet)");
      if ((size_t)page_non_null * 8 > vlen) throw Error(FDB_ERR_INVALID, "parquet: PLAIN page shorter than its values");
      out.plain_pages.push_back(FdbPqPlainPage{rank_done, (int64_t)voff});
    } else {
      if (h.encoding != ENC_RLE_DICTIONARY && h.encoding != ENC_PLAIN_DICTIONARY)
        throw Error(FDB_ERR_UNSUPPORTED, "parquet: BYTE_ARRAY pages must be dictionary-encoded (a writer that fell back to PLAIN is not supported on the device path)");
      if (!have_dict) throw Error(FDB_ERR_INVALID, "parquet: dictionary-encoded page without a dictionary page");
      if (page_non_null > 0) {
        if (vlen < 1) throw Error(FDB_ERR_INVALID, "parquet: dictionary-index page without a bit width");
        const int bw = base[voff];
        if (bw > 32) throw Error(FDB_ERR_INVALID, "parquet: dictionary index bit width > 32");
        out.max_index_bits = std::max<uint32_t>(out.max_index_bits, (uint32_t)bw);
        if (bw == 0) out.idx_runs.push_back(FdbPqRun{rank_done, 0ull, 0u, 0u});  // every index is 0
        else scan_runs(base, voff + 1, vlen - 1, bw, page_non_null, rank_done, &out.idx_runs, nullptr);
      }
    }
    rows_done += h.num_values;
    rank_done += page_non_null;
  }
  if (rows_done != n_rows) throw Error(FDB_ERR_INVALID, std::string("parquet: column chunk ") + (c.name ? c.name : "?") + " holds " + std::to_string(rows_done) + " values, the row group has " + std::to_string(n_rows) + " rows");
  out.non_null = rank_done;
  return out;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t kTailPad = 256;

}  // namespace

std::unique_ptr<DeviceBatch> batch_from_parquet(const fdb_parquet_chunk* chunks, int32_t n_chunks, int64_t n_rows, int device) {
  if (chunks == nullptr || n_chunks <= 0 || n_rows < 0) throw Error(FDB_ERR_INVALID, "parquet: no column chunks");
  std::vector<ParsedChunk> parsed;  // (host-only: malformed / unsupported chunks are refused before any device call)
  parsed.reserve((size_t)n_chunks);
  for (int32_t i = 0; i < n_chunks; i++) parsed.push_back(parse_chunk(chunks[i], n_rows));
  hip_check(hipSetDevice(device), "hipSetDevice");

  std::unique_ptr<DeviceBatch> b(new DeviceBatch());
  b->device = device;
  b->rows = n_rows;
  struct Piece { size_t val_off, bit_off; };
  std::vector<Piece> pieces((size_t)n_chunks);
  size_t total = 0;
  for (int32_t i = 0; i < n_chunks; i++) {
    const fdb_parquet_chunk& c = chunks[i];
    const size_t w = c.physical_type == 6 ? 4 : 8;
    pieces[(size_t)i].val_off = total;
    total += align_up((size_t)n_rows * w + kTailPad, 256);
    pieces[(size_t)i].bit_off = total;
    if (c.optional) total += align_up((size_t)((n_rows + 31) / 32) * 4 + kTailPad, 256);
  }
  if (n_rows > 0) { b->arena = device_pool_alloc(device, std::max<size_t>(total, 256)); b->arena_bytes = std::max<size_t>(total, 256); }

  Context* ctx = Context::acquire(device);
  struct Release { Context* c; ~Release() { if (c) { (void)hipStreamSynchronize(c->stream); c->reset_staging(); Context::release(c); } } } rel{ctx};
  hipStream_t stream = ctx->stream;
  std::vector<void*> scratch;
  struct FreeScratch { Context* c; std::vector<void*>* v; ~FreeScratch() { (void)hipStreamSynchronize(c->stream); for (void* p : *v) c->dev_free(p); } } fs{ctx, &scratch};
  const int64_t n_words = (n_rows + 31) / 32;
  std::vector<unsigned long long> h_totals((size_t)n_chunks, 0);
  std::vector<unsigned long long*> d_totals((size_t)n_chunks, nullptr);
  for (int32_t i = 0; i < n_chunks && n_rows > 0; i++) {
    const fdb_parquet_chunk& c = chunks[i];
    const ParsedChunk& P = parsed[(size_t)i];
    // the chunk's bytes as they are, padded so that 8-byte windows at the very end stay inside the allocation
    uint8_t* d_chunk = (uint8_t*)ctx->dev_alloc((size_t)c.n_bytes + 64);
    scratch.push_back(d_chunk);
    hip_check(hipMemcpyAsync(d_chunk, c.data, (size_t)c.n_bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(parquet chunk)");
    auto to_device = [&](const void* host, size_t bytes) -> void* {
      void* d = ctx->dev_alloc(std::max<size_t>(bytes, 16));
      scratch.push_back(d);
      if (bytes) hip_check(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(parquet tables)");
      return d;
    };
    uint32_t* d_valid = nullptr;
    uint32_t* d_prefix = nullptr;
    if (c.optional) {
      if (P.def_runs.empty()) throw Error(FDB_ERR_INVALID, "parquet: optional column without definition levels");
      const FdbPqRun* d_runs = (const FdbPqRun*)to_device(P.def_runs.data(), P.def_runs.size() * sizeof(FdbPqRun));
      d_valid = (uint32_t*)((unsigned char*)b->arena + pieces[(size_t)i].bit_off);
      d_prefix = (uint32_t*)ctx->dev_alloc((size_t)(n_words + 4) * 4);
      scratch.push_back(d_prefix);
      d_totals[(size_t)i] = (unsigned long long*)ctx->dev_alloc(64);
      scratch.push_back(d_totals[(size_t)i]);
      hip_check(fdb_launch_pq_validity(d_chunk, d_runs, (int32_t)P.def_runs.size(), n_rows, d_valid, d_prefix, stream), "parquet validity");
      hip_check(fdb_launch_tile_offsets(d_prefix, n_words, d_totals[(size_t)i], stream), "parquet rank scan");
    }
    void* d_out = (unsigned char*)b->arena + pieces[(size_t)i].val_off;
    if (c.physical_type == 6) {
      const FdbPqRun* d_idx = (const FdbPqRun*)to_device(P.idx_runs.data(), P.idx_runs.size() * sizeof(FdbPqRun));
      if (P.non_null > 0 && P.idx_runs.empty()) throw Error(FDB_ERR_INVALID, "parquet: values without index runs");
      if (P.idx_runs.empty()) hip_check(hipMemsetAsync(d_out, 0, (size_t)n_rows * 4, stream), "hipMemsetAsync");
      else hip_check(fdb_launch_pq_decode(1, d_chunk, d_valid, d_prefix, nullptr, 0, d_idx, (int32_t)P.idx_runs.size(), n_rows, d_out, stream), "parquet decode");
    } else {
      const FdbPqPlainPage* d_pages = (const FdbPqPlainPage*)to_device(P.plain_pages.data(), P.plain_pages.size() * sizeof(FdbPqPlainPage));
      if (P.plain_pages.empty()) hip_check(hipMemsetAsync(d_out, 0, (size_t)n_rows * 8, stream), "hipMemsetAsync");
      else hip_check(fdb_launch_pq_decode(0, d_chunk, d_valid, d_prefix, d_pages, (int32_t)P.plain_pages.size(), nullptr, 0, n_rows, d_out, stream), "parquet decode");
    }
    if (d_totals[(size_t)i] != nullptr)
      hip_check(hipMemcpyAsync(&h_totals[(size_t)i], d_totals[(size_t)i], 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(non-null count)");
  }
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(parquet decode)");

  for (int32_t i = 0; i < n_chunks; i++) {
    const fdb_parquet_chunk& c = chunks[i];
    const ParsedChunk& P = parsed[(size_t)i];
    DevColumn d;
    d.name = c.name ? c.name : "";
    d.length = n_rows;
    if (c.physical_type == 6) { d.kind = ColKind::DICT; d.format = "I"; d.dict = P.dict ? P.dict : make_dictionary({}, c.utf8 ? "u" : "z"); }
    else if (c.physical_type == 2) { d.kind = ColKind::I64; d.format = "l"; }
    else { d.kind = ColKind::F64; d.format = "g"; }
    const size_t w = c.physical_type == 6 ? 4 : 8;
    if (n_rows > 0) d.d_values = (unsigned char*)b->arena + pieces[(size_t)i].val_off;
    d.value_bytes = n_rows * (int64_t)w;
    if (c.optional && n_rows > 0) {
      if ((int64_t)h_totals[(size_t)i] != P.non_null) throw Error(FDB_ERR_INVALID, "parquet: definition levels and value counts disagree in column " + d.name);
      d.null_count = n_rows - P.non_null;
      if (d.null_count > 0) { d.d_validity = (uint8_t*)b->arena + pieces[(size_t)i].bit_off; d.validity_bytes = (n_rows + 7) / 8; }
    }
    if (d.kind == ColKind::DICT && P.max_index_bits > 0 && n_rows > 0) {
      // indices are validated like any imported dictionary column's (a corrupt page must not become an out-of-bounds LUT read)
      uint32_t* d_flag = (uint32_t*)ctx->dev_alloc(64);
      scratch.push_back(d_flag);
      uint32_t flag = 0;
      hip_check(hipMemsetAsync(d_flag, 0, 4, stream), "hipMemsetAsync");
      hip_check(fdb_launch_validate_indices((const uint32_t*)d.d_values, d.d_validity, n_rows, (uint32_t)std::min<size_t>(d.dict->values.size(), 0xFFFFFFFFu), d_flag, stream), "index check");
      hip_check(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
      if (flag != 0) throw Error(FDB_ERR_INVALID, "parquet: dictionary index out of range in column " + d.name);
    }
    b->payload_bytes += d.value_bytes + d.validity_bytes;
    b->cols.push_back(std::move(d));
  }
  return b;
}

}  // namespace fdb
