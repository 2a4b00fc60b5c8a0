// fdb_sort.hip — the one library primitive of the kernel set: a stable device radix sort of (key, value) pairs (rocPRIM), in a translation
// unit of its own so that its templates are not compiled with every change to fdb_kernels.hip. It serves the ordered Finish's FALLBACK —
// a run store whose keys did not arrive in order (≙ the merge of several ordered sets, ordered_aggregate.go:449-470) — not the scan.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "fdb_kernels.h"

hipError_t fdb_sort_pairs_u64(void* temp, size_t* temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned long long* vals_in,
                              unsigned long long* vals_out, int64_t n, int bits, hipStream_t stream) {
  if (n < 0 || bits < 1 || bits > 64) return hipErrorInvalidValue;
  return rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned int)bits, stream);
}
