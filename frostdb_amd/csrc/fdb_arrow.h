// fdb_arrow.h — Arrow C Data Interface plumbing: host views of incoming records and construction of
// outgoing ones. No Arrow library is linked; the ABI structs are all that crosses the boundary.
#pragma once

#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/frostdb_amd.h"

namespace fdb {

struct Error : public std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

enum class ColKind : int32_t { I64 = 1, U64 = 2, F64 = 3, BOOL = 4, STR = 5, DICT = 6, OTHER = 7 };

// The dictionary of one dictionary-encoded column, kept on the host (strings never go to the device).
struct HostDict {
  HostDict() = default;
  HostDict(const HostDict& o) : values(o.values), value_format(o.value_format), hash(o.hash), unique(o.unique), plain(o.plain), lens(o.lens), concat(o.concat) {}  // (the lazily computed caches below are not copied)
  HostDict& operator=(const HostDict& o) {
    if (this != &o) { values = o.values; value_format = o.value_format; hash = o.hash; unique = o.unique; plain = o.plain; lens = o.lens; concat = o.concat; }
    return *this;
  }
  std::vector<std::string> values;
  std::string value_format;  // "z" (binary) or "u" (utf8); large dictionaries are narrowed on import, a plain column keeps "Z" / "U"
  uint64_t hash = 0;         // content hash (lengths + bytes), computed at import
  bool unique = true;        // no two entries hold the same bytes (Arrow allows duplicates)
  bool plain = false;        // not an Arrow dictionary: the distinct values of a plain string / binary column, encoded by encode_plain
  // The entries' lengths and their bytes laid end to end (read_dictionary / make_dictionary): "is this record's dictionary the one
  // we know?" is then two memcmps instead of one per entry.
  std::vector<uint32_t> lens;
  std::string concat;
  // What depends on the CONTENT only is computed once per interned dictionary, not once per plan and Finish (a query creates its plans
  // anew; the dictionaries of a table's parts are shared): the Arrow offsets of the entries (n + 1, into `concat`) and the rank of
  // every entry among the sorted values (bytewise ascending; entries with equal bytes in entry order). 65 532-entry dictionaries on 32
  // group columns: ≈ 4 ms of every Finish before.
  const std::vector<int32_t>& arrow_offsets() const;
  const std::vector<uint32_t>& sorted_ranks() const;
  bool utf8() const { return value_format == "u" || value_format == "U"; }
  bool same_content(const HostDict& o) const { return hash == o.hash && plain == o.plain && values == o.values; }

 private:
  mutable std::once_flag offsets_once_, ranks_once_;
  mutable std::vector<int32_t> arrow_offsets_;
  mutable std::vector<uint32_t> sorted_ranks_;
};

// A borrowed view of one column of an incoming record; valid only while the caller's ArrowArray is.
struct HostColView {
  std::string name;
  std::string format;
  ColKind kind = ColKind::OTHER;
  int64_t length = 0, offset = 0, null_count = 0;
  const uint8_t* validity = nullptr;  // may be nullptr when null_count == 0
  const void* values = nullptr;       // fixed-width values / dictionary indices (element 0 of the buffer)
  int index_width = 0;                // DICT: bytes per index
  const ArrowArray* array = nullptr;
  const ArrowSchema* schema = nullptr;
};

struct HostRecordView {
  int64_t rows = 0;
  std::vector<HostColView> cols;
};

// Throws fdb::Error(FDB_ERR_INVALID) on a malformed record.
void view_record(const ArrowArray* array, const ArrowSchema* schema, HostRecordView* out);
std::shared_ptr<HostDict> read_dictionary(const HostColView& col);
std::shared_ptr<HostDict> make_dictionary(std::vector<std::string>&& values, const std::string& value_format);  // same interning, values from elsewhere
// A plain string / binary column (formats u, z, U, Z) never reaches the device as bytes: its distinct values become a HostDict
// (first-seen order, `plain` set) and `idx` gets one uint32 per row (0 for NULL rows) — from there on the column travels like
// a dictionary column, and the filter / group-key code asks `dict->plain` where the reference treats the two differently.
std::shared_ptr<HostDict> encode_plain(const HostColView& col, std::vector<uint32_t>* idx);
int64_t count_nulls(const uint8_t* validity, int64_t offset, int64_t length);
// Copies `length` bits starting at bit `offset` of `src` to bit 0 of `dst` (dst has (length+7)/8 bytes, zero padded).
void copy_bits(const uint8_t* src, int64_t offset, int64_t length, uint8_t* dst);

// ---- building an outgoing record ----------------------------------------------------------------------
struct OutColumn {
  std::string name;
  std::string format;                 // "l", "g", or the index format "I" for dictionary columns
  int64_t length = 0;
  int64_t null_count = 0;
  std::vector<uint8_t> validity;      // empty ⇒ no validity buffer
  std::vector<uint8_t> values;        // fixed-width values / uint32 indices
  // Big results: the buffers live in one block shared by every column of the record (pinned, filled by ONE device→host
  // copy) instead of the vectors above; `backing` returns the block to its pool when the last column is released.
  const uint8_t* ext_validity = nullptr;
  const uint8_t* ext_values = nullptr;
  std::shared_ptr<void> backing;
  // dictionary columns:
  bool is_dict = false;
  std::string dict_format;            // "z" / "u"
  std::vector<int32_t> dict_offsets;  // n_dict + 1
  std::vector<int64_t> dict_offsets64;  // … for the large formats "Z" / "U" (instead of dict_offsets)
  std::vector<char> dict_data;
  std::shared_ptr<const HostDict> dict_ref;  // instead of the three vectors above: the dictionary IS this interned one (its cached offsets, its bytes)
  int64_t dict_entries() const { return dict_ref ? (int64_t)dict_ref->values.size() : (int64_t)(dict_offsets64.empty() ? dict_offsets.size() : dict_offsets64.size()) - 1; }
  int64_t dict_offset(int64_t i) const { return dict_ref ? dict_ref->arrow_offsets()[(size_t)i] : dict_offsets64.empty() ? dict_offsets[(size_t)i] : dict_offsets64[(size_t)i]; }
  const char* dict_bytes() const { return dict_ref ? dict_ref->concat.data() : dict_data.data(); }
  // plain string / binary columns (format u / z, or U / Z with 64-bit offsets past 2 GiB of data): `values` holds the offsets
  bool is_str = false;
  std::vector<char> str_data;
};

// Fills the dictionary of `oc` (format `value_format`, "u" / "z") with `values`.
template <typename S>
void set_dictionary(OutColumn* oc, const std::vector<S>& values, const std::string& value_format);
// Turns `oc` into a plain string / binary column of `n` rows: row i holds values[idx[i]] where bit i of `valid_bits` is set
// (nullptr: everywhere), NULL elsewhere. `oc`'s validity (own or external) is kept; its value buffer is replaced by offsets + data.
template <typename S>
void set_plain_strings(OutColumn* oc, const uint32_t* idx, const uint8_t* valid_bits, int64_t n, const std::vector<S>& values,
                       const std::string& value_format);

// Moves `cols` into a heap holder and fills `out`/`out_schema` (struct-typed record, `rows` long) whose
// release callbacks free that holder.
void export_record(std::vector<OutColumn>&& cols, int64_t rows, ArrowArray* out, ArrowSchema* out_schema);
// Rows [start, start + len) of a finished record as columns of their own (buffers copied; a plain string column gets offsets that
// start at 0 again — 32-bit ones when the slice's bytes allow, whatever the whole column needed). The several-records Finish
// (aggregate.go:426-468) cuts its result with this.
std::vector<OutColumn> slice_columns(const std::vector<OutColumn>& cols, int64_t start, int64_t len);
// Bytes of value i of a plain string column built by set_plain_strings (32- or 64-bit offsets).
int64_t plain_string_bytes(const OutColumn& c, int64_t i);
int64_t plain_string_total(const OutColumn& c, int64_t n);  // bytes of its first n values

// Host-only: `view` → OutColumns through the same code push / finish use (read_dictionary, encode_plain, copy_bits,
// set_dictionary, set_plain_strings) → exported record. Throws FDB_ERR_UNSUPPORTED for column types the path does not handle.
void roundtrip_record(const HostRecordView& view, ArrowArray* out, ArrowSchema* out_schema);

}  // namespace fdb
