// Measurement aid (round 4): the rate of scattered global atomics by width, kind and memory scope (the high-cardinality scan's
// bottleneck: tools/scatter_probe.hip measured 23 G agent-scope atomics/s whatever the span, against 49–250 G loads/s).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o tools/atomic_probe && tools/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ __forceinline__ u64 mix(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

// KIND: 0 = u64 add, 1 = f64 add, 2 = u32 add, 3 = f32 add, 4 = i64 min, 5 = u64 CAS (expect 0), 6 = plain 8-byte store, 7 = u64 add returning
template <int KIND, int SCOPE>
__global__ __launch_bounds__(256) void k(u64* table, u64 n_entries, u64 window_entries, u64 ops_per_thread, u64* sink) {
  const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 base = 0, span = n_entries;
  if (window_entries != 0) { const u64 nw = n_entries / window_entries; base = (blockIdx.x % nw) * window_entries; span = window_entries; }
  u64 acc = 0;
  for (u64 it = 0; it < ops_per_thread; it += 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u64 idx = base + mix((tid * ops_per_thread + it + j) * 0x9E3779B97F4A7C15ULL + 12345) % span;
      u64* e = table + idx * 4 + 3;
      if (KIND == 0) __hip_atomic_fetch_add(e, 1ull, __ATOMIC_RELAXED, SCOPE);
      if (KIND == 1) __hip_atomic_fetch_add(reinterpret_cast<double*>(e), 1.0, __ATOMIC_RELAXED, SCOPE);
      if (KIND == 2) __hip_atomic_fetch_add(reinterpret_cast<u32*>(e), 1u, __ATOMIC_RELAXED, SCOPE);
      if (KIND == 3) __hip_atomic_fetch_add(reinterpret_cast<float*>(e), 1.0f, __ATOMIC_RELAXED, SCOPE);
      if (KIND == 4) __hip_atomic_fetch_min(reinterpret_cast<long long*>(e), (long long)idx, __ATOMIC_RELAXED, SCOPE);
      if (KIND == 5) { u64 exp = 0; __hip_atomic_compare_exchange_strong(e, &exp, idx | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE); acc += exp; }
      if (KIND == 6) __hip_atomic_store(e, idx, __ATOMIC_RELAXED, SCOPE);
      if (KIND == 7) acc += __hip_atomic_fetch_add(e, 1ull, __ATOMIC_RELAXED, SCOPE);
    }
  }
  if (acc == 0x1234567u) *sink = acc;
}
template <int KIND, int SCOPE>
void run(const char* what, const char* scope, u64* table, u64 n_entries, u64 window_entries, int grid, u64* sink) {
  const u64 ops = 64ull << 20, per_thread = (ops / ((u64)grid * 256) + 3) & ~3ull;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(a)); hipLaunchKernelGGL((k<KIND, SCOPE>), dim3(grid), dim3(256), 0, 0, table, n_entries, window_entries, per_thread, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  const double n = (double)per_thread * grid * 256;
  printf("%-22s scope %-10s span %7.1f MB window %8.3f MB: %7.3f ms = %6.1f G ops/s\n", what, scope, n_entries * 32 / 1e6, (window_entries ? window_entries : n_entries) * 32 / 1e6, best, n / best / 1e6);
}
#define ALLK(SC, name) \
  run<0, SC>("u64 add", name, table, n_entries, win, grid, sink); run<1, SC>("f64 add", name, table, n_entries, win, grid, sink); \
  run<2, SC>("u32 add", name, table, n_entries, win, grid, sink); run<3, SC>("f32 add", name, table, n_entries, win, grid, sink); \
  run<4, SC>("i64 min", name, table, n_entries, win, grid, sink); run<5, SC>("u64 CAS", name, table, n_entries, win, grid, sink); \
  run<6, SC>("8-byte store", name, table, n_entries, win, grid, sink); run<7, SC>("u64 add returning", name, table, n_entries, win, grid, sink);
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  u64* sink; CHECK(hipMalloc(&sink, 8));
  for (u64 span_mb : {2ull, 1024ull}) {
    const u64 n_entries = span_mb * 1024 * 1024 / 32;
    u64* table; CHECK(hipMalloc(&table, n_entries * 32)); CHECK(hipMemset(table, 0, n_entries * 32));
    const int grid = cus * 8;
    u64 win = 0;
    ALLK(__HIP_MEMORY_SCOPE_AGENT, "agent")
    ALLK(__HIP_MEMORY_SCOPE_WORKGROUP, "workgroup")
    ALLK(__HIP_MEMORY_SCOPE_SYSTEM, "system")
    if (span_mb >= 1024) { win = 128 * 1024 / 32; ALLK(__HIP_MEMORY_SCOPE_AGENT, "agent") ALLK(__HIP_MEMORY_SCOPE_WORKGROUP, "workgroup") }
    CHECK(hipFree(table));
  }
  return 0;
}
