// fdb_capi.cpp — the extern "C" surface declared in include/frostdb_amd.h. Every entry point converts C++
// exceptions into fdb_status codes (never aborts: the Go side wraps chains in recovery.Do, physicalplan.go:142).
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "fdb_comm.h"
#include "fdb_context.h"
#include "fdb_dynamic.h"
#include "fdb_jit.h"
#include "fdb_plan.h"
#include "fdb_regex.h"

// A plan handle: one operator chain. With aggregations over a DynamicColumn (fdb_dynamic.h) `plan` is the family's main plan
// (static aggregations / group keys) and `dyn` holds the children.
struct fdb_plan {
  std::unique_ptr<fdb::DynamicAggs> dyn;
  fdb_plan_desc main_desc;
  fdb::Plan plan;
  fdb_plan(const fdb_plan_desc* d, int dev)
      : dyn(fdb::DynamicAggs::wanted(d) ? new fdb::DynamicAggs(d, dev) : nullptr),
        main_desc(dyn ? dyn->main_desc() : (d ? *d : fdb_plan_desc())),
        plan(d ? &main_desc : nullptr, dev) {}
  // a fresh plan of `proto`'s descriptor (the shard of fdb_plan_exchange)
  explicit fdb_plan(const fdb::Plan& proto) : main_desc(), plan(proto, fdb::Plan::CloneTag()) {}
};
struct fdb_comm { std::unique_ptr<fdb::Comm> c; };
struct fdb_batch { std::unique_ptr<fdb::DeviceBatch> b; };

namespace {
thread_local std::string g_last_error;

template <typename F>
int guard(fdb_plan* p, F&& f) {
  try {
    f();
    return FDB_OK;
  } catch (const fdb::Error& e) {
    if (p) p->plan.error = e.what(); else g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    if (p) p->plan.error = "out of host memory"; else g_last_error = "out of host memory";
    return FDB_ERR_OOM;
  } catch (const std::exception& e) {
    if (p) p->plan.error = e.what(); else g_last_error = e.what();
    return FDB_ERR_INVALID;
  } catch (...) {
    if (p) p->plan.error = "unknown error"; else g_last_error = "unknown error";
    return FDB_ERR_INVALID;
  }
}
// Entry points that expose ONE table (state arrays for the RCCL merges, the hash exchange) do not apply to a family of plans.
void single_table_only(const fdb_plan* p) {
  if (p->dyn) throw fdb::Error(FDB_ERR_UNSUPPORTED, "not available for a plan with aggregations over a dynamic column set (merge such plans with fdb_plan_merge)");
}
}  // namespace

namespace {
template <typename F>
int comm_guard(fdb_comm* c, F&& f) {
  try {
    f();
    return FDB_OK;
  } catch (const fdb::Error& e) {
    if (c && c->c) c->c->error = e.what();
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    g_last_error = "out of host memory";
    return FDB_ERR_OOM;
  } catch (const std::exception& e) {
    if (c && c->c) c->c->error = e.what();
    g_last_error = e.what();
    return FDB_ERR_INVALID;
  }
}
int wrap_all(std::vector<std::unique_ptr<fdb::Comm>>&& v, fdb_comm** out) {
  for (size_t i = 0; i < v.size(); i++) { out[i] = new fdb_comm(); out[i]->c = std::move(v[i]); }
  return FDB_OK;
}
}  // namespace

namespace fdb { void widen_indices(const void* src, int width, uint32_t* dst, size_t n); void widen_indices_mapped(const void* src, int width, const uint32_t* table, size_t table_len, uint32_t* dst, size_t n); }  // fdb_widen.cc

extern "C" {

const char* fdb_version(void) { return "frostdb_amd 0.1.0 (gfx950)"; }
const char* fdb_last_error(void) { return g_last_error.c_str(); }

int fdb_device_count(int* n_devices) {
  return guard(nullptr, [&] {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
    *n_devices = n;
  });
}

int fdb_plan_explain(const fdb_plan_desc* desc, char* buf, int64_t capacity, int64_t* needed) {
  return guard(nullptr, [&] {
    std::string s;
    if (fdb::DynamicAggs::wanted(desc)) {
      fdb::DynamicAggs dyn(desc, 0);
      const fdb_plan_desc md = dyn.main_desc();
      fdb::Plan plan(&md, 0, /*explain_only=*/true);
      s = dyn.draw(plan);
    } else {
      fdb::Plan plan(desc, 0, /*explain_only=*/true);
      s = plan.draw();
    }
    if (needed != nullptr) *needed = (int64_t)s.size() + 1;
    if (buf != nullptr && capacity > 0) {
      const size_t n = std::min((size_t)capacity - 1, s.size());
      std::memcpy(buf, s.data(), n);
      buf[n] = 0;
    }
  });
}

int fdb_selftest_widen(const void* src, int32_t width, uint32_t* dst, int64_t n) {
  return guard(nullptr, [&] {
    if ((width != 1 && width != 2 && width != 4 && width != -2 && width != -4 && width != -12 && width != -14) || n < 0 || ((src == nullptr || dst == nullptr) && n > 0))
      throw fdb::Error(FDB_ERR_INVALID, "widen: width 1, 2, 4 (bytes) or -2, -4 (bits) and non-null buffers");
    if (width == -12 || width == -14) {  // the rank → index road of the same widths, through the table t[r] = 3 r + 5 of 4 / 16 entries
      uint32_t table[16];
      for (uint32_t r = 0; r < 16; r++) table[r] = 3 * r + 5;
      fdb::widen_indices_mapped(src, width + 10, table, width == -12 ? 4 : 16, dst, (size_t)n);
      return;
    }
    fdb::widen_indices(src, width, dst, (size_t)n);
  });
}

int fdb_arrow_roundtrip(struct ArrowArray* batch, struct ArrowSchema* schema, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  return guard(nullptr, [&] {
    if (out == nullptr || out_schema == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null output");
    fdb::HostRecordView view;
    fdb::view_record(batch, schema, &view);
    fdb::roundtrip_record(view, out, out_schema);
  });
}

int fdb_read_ceiling(int device, int64_t bytes, int32_t reps, double* gb_per_s) {
  return guard(nullptr, [&] {
    if (gb_per_s == nullptr || bytes < (1 << 20) || reps < 1) throw fdb::Error(FDB_ERR_INVALID, "fdb_read_ceiling: bytes >= 1 MiB, reps >= 1");
    *gb_per_s = 0;
    bytes &= ~(int64_t)4095;
    if (hipSetDevice(device) != hipSuccess) throw fdb::Error(FDB_ERR_DEVICE, "hipSetDevice failed");
    void* buf = nullptr;
    unsigned long long* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&] { if (buf) (void)hipFree(buf); if (out) (void)hipFree(out); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
    auto ck = [&](hipError_t e, const char* what) { if (e != hipSuccess) { cleanup(); throw fdb::Error(e == hipErrorOutOfMemory ? FDB_ERR_OOM : FDB_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); } };
    ck(hipMalloc(&buf, (size_t)bytes), "hipMalloc");
    ck(hipMalloc((void**)&out, 64), "hipMalloc");
    ck(hipMemset(buf, 0, (size_t)bytes), "hipMemset");
    ck(hipEventCreate(&e0), "hipEventCreate");
    ck(hipEventCreate(&e1), "hipEventCreate");
    ck(fdb_launch_stream_read(buf, bytes, out, nullptr), "launch");  // warm-up
    float best = 0;
    for (int r = 0; r < reps; r++) {
      ck(hipEventRecord(e0, nullptr), "hipEventRecord");
      ck(fdb_launch_stream_read(buf, bytes, out, nullptr), "launch");
      ck(hipEventRecord(e1, nullptr), "hipEventRecord");
      ck(hipEventSynchronize(e1), "hipEventSynchronize");
      float ms = 0;
      ck(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
      if (ms > 0 && (best == 0 || ms < best)) best = ms;
    }
    cleanup();
    if (best > 0) *gb_per_s = (double)bytes / (best * 1e-3) / 1e9;
  });
}

int fdb_plan_create(const fdb_plan_desc* desc, int device, fdb_plan** out) {
  return guard(nullptr, [&] {
    if (out == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null out pointer");
    if (desc != nullptr && desc->ordered != 0 && fdb::DynamicAggs::wanted(desc))
      throw fdb::Error(FDB_ERR_UNSUPPORTED, "OrderedAggregate over a dynamic column set is not supported");
    *out = new fdb_plan(desc, device);
  });
}

int fdb_plan_push(fdb_plan* plan, struct ArrowArray* batch, struct ArrowSchema* schema) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { if (plan->dyn) plan->dyn->push(plan->plan, batch, schema); else plan->plan.push(batch, schema); });
}

int fdb_plan_push_many(fdb_plan* plan, struct ArrowArray* const* batches, struct ArrowSchema* const* schemas, int32_t n, int32_t* n_pushed) {
  if (n_pushed) *n_pushed = 0;
  if (!plan || n < 0 || (n > 0 && (!batches || !schemas))) return FDB_ERR_INVALID;
  for (int32_t i = 0; i < n; i++) {
    const int rc = fdb_plan_push(plan, batches[i], schemas[i]);
    if (rc != 0) return rc;
    if (n_pushed) *n_pushed = i + 1;
  }
  return 0;
}

int fdb_plan_push_batch(fdb_plan* plan, const fdb_batch* batch) {
  if (!plan || !batch) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (plan->dyn) { const fdb::DeviceBatch* b = batch->b.get(); plan->dyn->push_batches(plan->plan, &b, 1); return; }
    plan->plan.settle();
    plan->plan.push_batch(*batch->b);
  });
}

int fdb_plan_push_batches(fdb_plan* plan, const fdb_batch* const* batches, int32_t n) {
  if (!plan || (n > 0 && !batches)) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    std::vector<const fdb::DeviceBatch*> v;
    for (int32_t i = 0; i < n; i++) {
      if (batches[i] == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null batch");
      v.push_back(batches[i]->b.get());
    }
    if (plan->dyn) { plan->dyn->push_batches(plan->plan, v.data(), (int)v.size()); return; }
    plan->plan.settle();
    plan->plan.push_batches(v.data(), (int)v.size());
  });
}

int fdb_plan_finish(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema, int64_t* n_rows) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (plan->dyn) { plan->dyn->finish(plan->plan, out, out_schema, n_rows); return; }
    plan->plan.settle();
    plan->plan.finish(out, out_schema, n_rows);
  });
}

int fdb_plan_finish_next(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema, int64_t* n_rows, int32_t* emitted) {
  if (!plan || !out || !out_schema || !emitted) return FDB_ERR_INVALID;
  *emitted = 0;
  return guard(plan, [&] {
    std::memset(out, 0, sizeof(*out));
    std::memset(out_schema, 0, sizeof(*out_schema));
    *emitted = plan->plan.finish_next(out, out_schema, n_rows) ? 1 : 0;  // (a family of plans behind a dynamic aggregation emits one record: nothing pending)
  });
}

int fdb_plan_merge(fdb_plan* dst, fdb_plan* src) {
  if (!dst || !src) return FDB_ERR_INVALID;
  return guard(dst, [&] {
    if ((dst->dyn != nullptr) != (src->dyn != nullptr)) throw fdb::Error(FDB_ERR_INVALID, "plans have different aggregations");
    if (dst->dyn) { dst->dyn->merge_from(dst->plan, *src->dyn, src->plan); return; }
    src->plan.settle();
    dst->plan.settle();
    dst->plan.merge_from(src->plan);
  });
}

int fdb_plan_group_schema(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  if (!plan || !out || !out_schema) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.group_schema(out, out_schema); });
}

int fdb_plan_seed_groups(fdb_plan* plan, struct ArrowArray* schema_record, struct ArrowSchema* schema) {
  if (!plan || !schema_record || !schema) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.seed_groups(schema_record, schema); });
}

int fdb_plan_hash_export(fdb_plan* src, fdb_plan* layout, int32_t n_parts, void** dev_rows, int64_t* counts, int32_t* row_words32) {
  if (!src || !layout || !dev_rows || !counts || !row_words32) return FDB_ERR_INVALID;
  return guard(src, [&] { single_table_only(src); single_table_only(layout); src->plan.settle(); layout->plan.settle(); src->plan.hash_export(layout->plan, n_parts, dev_rows, counts, row_words32); });
}

int fdb_plan_hash_import(fdb_plan* plan, const void* dev_rows, int64_t n_rows) {
  if (!plan || (n_rows > 0 && !dev_rows)) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.hash_import(dev_rows, n_rows); });
}

int fdb_plan_filter(fdb_plan* plan, struct ArrowArray* batch, struct ArrowSchema* schema, struct ArrowArray* out,
                    struct ArrowSchema* out_schema, int64_t* n_selected) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { plan->plan.filter(batch, schema, out, out_schema, n_selected); });
}

int fdb_plan_select(fdb_plan* plan, struct ArrowArray* batch, struct ArrowSchema* schema, uint32_t* indices, int64_t capacity,
                    int64_t* n_selected) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { plan->plan.select(batch, schema, indices, capacity, n_selected); });
}

int fdb_plan_filter_batch(fdb_plan* plan, const fdb_batch* batch, fdb_batch** out, int64_t* n_selected) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (batch == nullptr || !batch->b || out == nullptr || n_selected == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<fdb::DeviceBatch> r = plan->plan.filter_batch(*batch->b, n_selected);
    *out = new fdb_batch{std::move(r)};
  });
}

int fdb_plan_finish_batch(fdb_plan* plan, fdb_batch** out, int64_t* n_rows) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (out == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    *out = nullptr;
    single_table_only(plan);
    std::unique_ptr<fdb::DeviceBatch> r = plan->plan.finish_batch(n_rows);
    *out = new fdb_batch{std::move(r)};
  });
}

int fdb_plan_filter_batches(fdb_plan* plan, const fdb_batch* const* batches, int32_t n, fdb_batch** out, int64_t* n_selected) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (n < 0 || (n > 0 && (batches == nullptr || out == nullptr || n_selected == nullptr))) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    std::vector<const fdb::DeviceBatch*> bs;
    for (int32_t i = 0; i < n; i++) {
      out[i] = nullptr;
      if (batches[i] == nullptr || !batches[i]->b) throw fdb::Error(FDB_ERR_INVALID, "null batch");
      bs.push_back(batches[i]->b.get());
    }
    std::vector<std::unique_ptr<fdb::DeviceBatch>> rs = plan->plan.filter_batches(bs.data(), n, n_selected);
    for (int32_t i = 0; i < n; i++) out[i] = new fdb_batch{std::move(rs[(size_t)i])};
  });
}

int fdb_plan_select_batch(fdb_plan* plan, const fdb_batch* batch, uint32_t* dev_indices, int64_t capacity, int64_t* n_selected) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (batch == nullptr || !batch->b || dev_indices == nullptr || n_selected == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    *n_selected = plan->plan.select_batch(*batch->b, dev_indices, capacity);
  });
}

int fdb_batch_export(const fdb_batch* batch, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  return guard(nullptr, [&] {
    if (batch == nullptr || !batch->b || out == nullptr || out_schema == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    fdb::export_batch(*batch->b, out, out_schema);
  });
}

const char* fdb_plan_draw(fdb_plan* plan) { return !plan ? "" : plan->dyn ? plan->dyn->draw(plan->plan) : plan->plan.draw(); }
const char* fdb_plan_last_error(const fdb_plan* plan) { return plan ? plan->plan.error.c_str() : g_last_error.c_str(); }
void fdb_plan_close(fdb_plan* plan) { delete plan; }

int fdb_plan_num_groups(fdb_plan* plan, int64_t* n_groups) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (plan->dyn) { *n_groups = plan->dyn->num_groups(plan->plan); return; }
    plan->plan.settle();
    *n_groups = plan->plan.num_groups();
  });
}

int fdb_plan_partial_keys(fdb_plan* plan, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.partial_keys(out, out_schema); });
}

int fdb_plan_partial_state(fdb_plan* plan, int32_t agg, void* dst, int64_t capacity_bytes) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.partial_state(agg, dst, capacity_bytes); });
}

int fdb_plan_state_signature(fdb_plan* plan, uint64_t* signature, int64_t* n_slots) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); *signature = plan->plan.state_signature(n_slots); });
}

int fdb_plan_state_pointers(fdb_plan* plan, void** base, int64_t* array_stride, int64_t* n_slots) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.state_pointers(base, array_stride, n_slots); });
}

int fdb_plan_state_read(fdb_plan* plan, int32_t array, void* dst, int64_t capacity_bytes) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.state_read(array, dst, capacity_bytes); });
}

int fdb_plan_state_write(fdb_plan* plan, int32_t array, const void* src, int64_t bytes) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); plan->plan.state_write(array, src, bytes); });
}

int fdb_plan_state_arrays(fdb_plan* plan, int32_t* n_arrays) {
  if (!plan || !n_arrays) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); *n_arrays = plan->plan.num_state_arrays(); });
}

int fdb_plan_state_array_op(fdb_plan* plan, int32_t array, int32_t* op) {
  if (!plan || !op) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); *op = plan->plan.state_array_op(array); });
}

int fdb_plan_agg_type(fdb_plan* plan, int32_t agg, char* format_out) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] { single_table_only(plan); plan->plan.settle(); *format_out = plan->plan.agg_format(agg); });
}

int fdb_batch_import(struct ArrowArray* batch, struct ArrowSchema* schema, int device, fdb_batch** out) {
  return guard(nullptr, [&] {
    fdb::HostRecordView view;
    fdb::view_record(batch, schema, &view);
    std::unique_ptr<fdb_batch> b(new fdb_batch());
    b->b = fdb::import_batch(view, device, nullptr, nullptr);
    *out = b.release();
  });
}

int fdb_batch_from_parquet(const fdb_parquet_chunk* chunks, int32_t n_chunks, int64_t n_rows, int device, fdb_batch** out) {
  return guard(nullptr, [&] {
    if (out == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null output");
    *out = nullptr;
    std::unique_ptr<fdb_batch> b(new fdb_batch());
    b->b = fdb::batch_from_parquet(chunks, n_chunks, n_rows, device);
    *out = b.release();
  });
}

int fdb_batches_from_parquet(const fdb_parquet_row_group* groups, int32_t n_groups, int device, fdb_batch** out) {
  return guard(nullptr, [&] {
    if (out == nullptr || n_groups <= 0) throw fdb::Error(FDB_ERR_INVALID, "null output");
    for (int32_t g = 0; g < n_groups; g++) out[g] = nullptr;
    std::vector<std::unique_ptr<fdb::DeviceBatch>> bs = fdb::batches_from_parquet(groups, n_groups, device);
    std::vector<std::unique_ptr<fdb_batch>> hs;
    for (auto& b : bs) { hs.emplace_back(new fdb_batch()); hs.back()->b = std::move(b); }
    for (int32_t g = 0; g < n_groups; g++) out[g] = hs[(size_t)g].release();
  });
}

int fdb_snappy_decode_pages(const uint8_t* src, int64_t src_bytes, const fdb_snappy_page* pages, int32_t n_pages, uint8_t* dst, int64_t dst_bytes,
                            int device, uint32_t* status, double* kernel_ms) {
  static_assert(sizeof(fdb_snappy_page) == sizeof(FdbSnappyPage), "fdb_snappy_page mirrors FdbSnappyPage");
  return guard(nullptr, [&] {
    if (n_pages < 0 || src_bytes < 0 || dst_bytes < 0 || (n_pages > 0 && (pages == nullptr || status == nullptr))) throw fdb::Error(FDB_ERR_INVALID, "snappy: bad arguments");
    for (int32_t i = 0; i < n_pages; i++)  // every page stays inside the buffers (the kernel checks a page against its own lengths only)
      if (pages[i].src_off > (uint64_t)src_bytes || pages[i].src_len > (uint64_t)src_bytes - pages[i].src_off || pages[i].dst_off > (uint64_t)dst_bytes ||
          pages[i].dst_len > (uint64_t)dst_bytes - pages[i].dst_off)
        throw fdb::Error(FDB_ERR_INVALID, "snappy: page " + std::to_string(i) + " lies outside the buffers");
    if (kernel_ms) *kernel_ms = 0.0;
    if (n_pages == 0) return;
    fdb::hip_check(hipSetDevice(device), "hipSetDevice");
    struct Dev { void* p = nullptr; ~Dev() { if (p) (void)hipFree(p); } } d_src, d_dst, d_pages, d_status;
    struct Ev { hipEvent_t e = nullptr; ~Ev() { if (e) (void)hipEventDestroy(e); } } e0, e1;
    const size_t pad = 64;  // (the kernel's 16-byte moves never start past a page's last byte, but may end up to 15 bytes behind it)
    fdb::hip_check(hipMalloc(&d_src.p, (size_t)src_bytes + pad), "hipMalloc");
    fdb::hip_check(hipMalloc(&d_dst.p, (size_t)dst_bytes + pad), "hipMalloc");
    fdb::hip_check(hipMalloc(&d_pages.p, (size_t)n_pages * sizeof(FdbSnappyPage)), "hipMalloc");
    fdb::hip_check(hipMalloc(&d_status.p, (size_t)n_pages * 4), "hipMalloc");
    fdb::hip_check(hipMemcpy(d_src.p, src, (size_t)src_bytes, hipMemcpyHostToDevice), "hipMemcpy");
    fdb::hip_check(hipMemcpy(d_pages.p, pages, (size_t)n_pages * sizeof(FdbSnappyPage), hipMemcpyHostToDevice), "hipMemcpy");
    fdb::hip_check(hipEventCreate(&e0.e), "hipEventCreate");
    fdb::hip_check(hipEventCreate(&e1.e), "hipEventCreate");
    fdb::hip_check(hipEventRecord(e0.e, nullptr), "hipEventRecord");
    fdb::hip_check(fdb_launch_snappy_decode((const uint8_t*)d_src.p, (const FdbSnappyPage*)d_pages.p, n_pages, (uint8_t*)d_dst.p, (uint32_t*)d_status.p, nullptr), "snappy launch");
    fdb::hip_check(hipEventRecord(e1.e, nullptr), "hipEventRecord");
    fdb::hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    float ms = 0.f;
    fdb::hip_check(hipEventElapsedTime(&ms, e0.e, e1.e), "hipEventElapsedTime");
    if (kernel_ms) *kernel_ms = (double)ms;
    fdb::hip_check(hipMemcpy(status, d_status.p, (size_t)n_pages * 4, hipMemcpyDeviceToHost), "hipMemcpy");
    if (dst_bytes > 0) fdb::hip_check(hipMemcpy(dst, d_dst.p, (size_t)dst_bytes, hipMemcpyDeviceToHost), "hipMemcpy");
  });
}

int64_t fdb_batch_num_rows(const fdb_batch* batch) { return batch ? batch->b->rows : 0; }
int64_t fdb_batch_device_bytes(const fdb_batch* batch) { return batch ? (int64_t)batch->b->arena_bytes : 0; }
void fdb_batch_release(fdb_batch* batch) { delete batch; }

int fdb_plan_stats(fdb_plan* plan, int64_t* algorithmic_bytes, double* kernel_ms, int64_t* n_launches, int64_t* rows_scanned) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    plan->plan.settle();  // queued small records count once they are scanned
    if (plan->plan.timing) plan->plan.sync();  // (event pairs are read once their kernels have run: a rank whose merge emitted no record has not waited yet)
    if (algorithmic_bytes) *algorithmic_bytes = plan->plan.stat_bytes;
    if (kernel_ms) *kernel_ms = plan->plan.stat_ms;
    if (n_launches) *n_launches = plan->plan.stat_launches;
    if (rows_scanned) *rows_scanned = plan->plan.stat_rows;
  });
}

int fdb_regex_match(const char* pattern, int64_t pattern_len, const uint8_t* value, int64_t value_len, int32_t* matched) {
  return guard(nullptr, [&] {
    if (pattern == nullptr || pattern_len < 0 || value_len < 0 || (value == nullptr && value_len > 0) || matched == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    std::string why;
    std::shared_ptr<const fdb::Regex> re = fdb::Regex::compile(std::string(pattern, (size_t)pattern_len), &why);
    if (!re) throw fdb::Error(FDB_ERR_INVALID, why);
    *matched = re->match((const char*)value, (size_t)value_len) ? 1 : 0;
  });
}

int fdb_parquet_stats(int64_t* calls, double* host_ms, double* device_ms, int64_t* file_bytes, int64_t* out_bytes) {
  return guard(nullptr, [&] { fdb::parquet_stats(calls, host_ms, device_ms, file_bytes, out_bytes); });
}

int fdb_jit_stats(int64_t* n_compiled, double* compile_ms, int64_t* n_disk_loads) {
  return guard(nullptr, [&] { fdb::jit_stats(n_compiled, compile_ms, n_disk_loads); });
}

int fdb_plan_merge_ms(fdb_plan* plan, double* merge_ms) {
  if (!plan || !merge_ms) return FDB_ERR_INVALID;
  return guard(plan, [&] { if (plan->plan.timing) plan->plan.sync(); *merge_ms = plan->plan.stat_merge_ms; });
}

int fdb_plan_set_timing(fdb_plan* plan, int32_t enabled) {
  if (!plan) return FDB_ERR_INVALID;
  plan->plan.timing = enabled != 0;
  return FDB_OK;
}

int fdb_plan_stream(fdb_plan* plan, void** stream_out) {
  if (!plan) return FDB_ERR_INVALID;
  *stream_out = (void*)plan->plan.stream();
  return FDB_OK;
}

// Tuning knobs used by bench.py's variant sweeps (not part of the Go binding).
int fdb_plan_set_tuning(fdb_plan* plan, int32_t rows_per_thread, int32_t grid_blocks) {
  if (!plan) return FDB_ERR_INVALID;
  if (rows_per_thread == 0 || rows_per_thread == 4 || rows_per_thread == 8) plan->plan.rows_per_thread = rows_per_thread;
  plan->plan.grid_override = grid_blocks & 0xFFFFF;
  plan->plan.ablate = (grid_blocks >> 20) & 0xF;  // bench --ablate rides in the high bits (tuning aid only)
  plan->plan.use_partials = ((grid_blocks >> 24) & 1) == 0;
  plan->plan.sub_tiles = (grid_blocks >> 25) & 7;
  return FDB_OK;
}

int fdb_plan_set_deterministic(fdb_plan* plan, int32_t enabled) {
  if (!plan) return FDB_ERR_INVALID;
  plan->plan.deterministic = enabled != 0;
  return FDB_OK;
}

const char* fdb_plan_last_kernel(fdb_plan* plan) {
  if (!plan) return "";
  (void)guard(plan, [&] { plan->plan.settle(); });
  return plan->plan.last_kernel();
}

// ---- cross-GPU merge ---------------------------------------------------------------------------------------------------------

int fdb_comm_unique_id(uint8_t id[FDB_COMM_ID_BYTES]) {
  return comm_guard(nullptr, [&] {
    if (id == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null id");
    fdb::rccl_unique_id(id);
  });
}

int fdb_comm_init_rank(const uint8_t id[FDB_COMM_ID_BYTES], int32_t n_ranks, int32_t rank, int device, fdb_comm** out) {
  return comm_guard(nullptr, [&] {
    if (id == nullptr || out == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<fdb::Comm> c = fdb::rccl_init_rank(id, n_ranks, rank, device);
    *out = new fdb_comm();
    (*out)->c = std::move(c);
  });
}

int fdb_comm_init_all(const int* devices, int32_t n, fdb_comm** out) {
  return comm_guard(nullptr, [&] {
    if (devices == nullptr || out == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    for (int32_t i = 0; i < n && i < FDB_MAX_PARTS; i++) out[i] = nullptr;  // (nothing dangling if creation fails)
    wrap_all(fdb::rccl_init_all(devices, n), out);
  });
}

int fdb_comm_init_local(const int* devices, int32_t n, fdb_comm** out) {
  return comm_guard(nullptr, [&] {
    if (devices == nullptr || out == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    for (int32_t i = 0; i < n && i < FDB_MAX_PARTS; i++) out[i] = nullptr;  // (nothing dangling if creation fails)
    wrap_all(fdb::local_init(devices, n), out);
  });
}

int32_t fdb_comm_rank(const fdb_comm* comm) { return comm && comm->c ? comm->c->rank : -1; }
int32_t fdb_comm_size(const fdb_comm* comm) { return comm && comm->c ? comm->c->size : 0; }
int32_t fdb_comm_transport_ranks(fdb_comm* comm) {
  if (comm == nullptr || !comm->c) return -1;
  try { return comm->c->transport_ranks(); } catch (...) { return -1; }
}
const char* fdb_comm_last_error(const fdb_comm* comm) { return comm && comm->c ? comm->c->error.c_str() : g_last_error.c_str(); }
void fdb_comm_destroy(fdb_comm* comm) { delete comm; }

int fdb_plan_allreduce(fdb_plan* plan, fdb_comm* comm, int32_t* aligned) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (comm == nullptr || !comm->c || aligned == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    single_table_only(plan);
    *aligned = plan->plan.comm_allreduce(*comm->c) ? 1 : 0;
  });
}

int fdb_plan_exchange(fdb_plan* plan, fdb_comm* comm, fdb_plan** shard) {
  if (!plan) return FDB_ERR_INVALID;
  return guard(plan, [&] {
    if (comm == nullptr || !comm->c || shard == nullptr) throw fdb::Error(FDB_ERR_INVALID, "null argument");
    single_table_only(plan);
    *shard = nullptr;
    std::unique_ptr<fdb_plan> s(new fdb_plan(plan->plan));
    plan->plan.comm_exchange(*comm->c, s->plan);
    *shard = s.release();
  });
}

int fdb_live_allocations(int64_t* device_blocks, int64_t* device_bytes, int64_t* pinned_blocks) {
  return guard(nullptr, [&] { fdb::live_allocations(device_blocks, device_bytes, pinned_blocks); });
}

}  // extern "C"
