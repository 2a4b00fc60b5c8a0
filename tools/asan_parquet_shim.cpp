// ASan harness for the host side of fdb_batch_from_parquet: kernel launchers are stubbed (parsing happens before any of them)
#include <hip/hip_runtime_api.h>
#include "fdb_kernels.h"
#include "fdb_arrow.h"
#include "fdb_plan.h"
#include "fdb_context.h"
#include "frostdb_amd.h"
#include <string>
hipError_t fdb_launch_pq_validity(const uint8_t*, const FdbPqRun*, int32_t, int64_t, uint32_t*, uint32_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t fdb_launch_pq_decode(int, const uint8_t*, const uint32_t*, const uint32_t*, const FdbPqPlainPage*, int32_t, const FdbPqRun*, int32_t, int64_t, void*, hipStream_t) { return hipErrorNotSupported; }
hipError_t fdb_launch_pq_delta(const uint8_t*, const FdbPqDeltaPage*, int32_t, const FdbPqDeltaMini*, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
hipError_t fdb_launch_exclusive_scan(uint32_t*, int64_t, uint32_t*, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
hipError_t fdb_launch_validate_indices(const uint32_t*, const uint8_t*, int64_t, uint32_t, uint32_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t fdb_launch_snappy_decode(const uint8_t*, const FdbSnappyPage*, int32_t, uint8_t*, uint32_t*, hipStream_t) { return hipErrorNotSupported; }
static thread_local std::string g_err;
extern "C" const char* fdb_last_error(void) { return g_err.c_str(); }
extern "C" int fdb_batch_from_parquet(const fdb_parquet_chunk* chunks, int32_t n, int64_t rows, int device, fdb_batch** out) {
  try { auto b = fdb::batch_from_parquet(chunks, n, rows, device); (void)b; return 0; }
  catch (const fdb::Error& e) { g_err = e.what(); return e.code; }
  catch (const std::exception& e) { g_err = e.what(); return FDB_ERR_INVALID; }
}
extern "C" int fdb_batches_from_parquet(const fdb_parquet_row_group* groups, int32_t n_groups, int device, fdb_batch** out) {
  try { auto b = fdb::batches_from_parquet(groups, n_groups, device); (void)b; return 0; }
  catch (const fdb::Error& e) { g_err = e.what(); return e.code; }
  catch (const std::exception& e) { g_err = e.what(); return FDB_ERR_INVALID; }
}
namespace fdb {
void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) { (void)hipGetLastError(); throw Error(e == hipErrorOutOfMemory ? FDB_ERR_OOM : FDB_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }
}
}
namespace fdb {
DeviceBatch::~DeviceBatch() {
  if (arena == nullptr) return;
  if (arena_ctx != nullptr) { arena_ctx->dev_free(arena); return; }
  device_pool_free(device, arena);
}
}
