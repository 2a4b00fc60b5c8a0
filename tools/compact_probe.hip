// Measurement aid: stream compaction WITHOUT LDS staging. A lane owns row (group · 64 + lane) instead of 4 consecutive rows, so
// the selection bitmap's 64-bit words ARE the ballots (scalar loads, no __ballot chain), the rows a group keeps land at consecutive
// output positions (base + mbcnt(word)) and are stored straight from the registers they were loaded into — partially filled but
// contiguous stores that L2 has to merge. Question: does that beat load → ballots → LDS scatter → LDS read → 16-byte stores
// (compact_multi_kernel, 0.69 ms per 100 M rows × 24 B at 50 %)?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/compact_probe.hip -o tools/compact_probe && tools/compact_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define G1 __attribute__((address_space(1)))
#define C4 __attribute__((address_space(4)))
typedef unsigned long long u64;

__device__ __forceinline__ u64 mix64(u64 x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void fill_masks(u64* m, long long n_words, int mode) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (long long)gridDim.x * 256) {
    const u64 a = mix64(i * 3), b = mix64(i * 3 + 1), c = mix64(i * 3 + 2);
    m[i] = mode == 0 ? (a & b & c) : mode == 1 ? a : (a | b | c);  // 12.5 % / 50 % / 87.5 %
  }
}
template <typename T> __global__ void fill_iota(T* p, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = (T)i;
}

struct Col { const void* src; void* dst; int width; };
struct Args { Col cols[4]; int wave_begin[5]; const u64* masks; const uint32_t* tile_offsets; long long n_tiles; };

constexpr int TILE_GROUPS = 32;  // 2 048 rows per tile

template <typename T, int U, bool NT>
__device__ __forceinline__ void stream(const T* __restrict__ src, T* __restrict__ dst, const u64* __restrict__ masks, const uint32_t* __restrict__ offsets, long long first, long long stride,
                                       long long n_tiles, int lane) {
  const C4 u64* sm = (const C4 u64*)masks;
  const C4 uint32_t* so = (const C4 uint32_t*)offsets;
  const u64 lane_bit = 1ull << lane;
  auto load = [&](T (&x)[U], long long tile, int g0) {
    const T* s = src + tile * (TILE_GROUPS * 64) + (long long)g0 * 64 + lane;
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = NT ? __builtin_nontemporal_load((const G1 T*)(s + u * 64)) : s[u * 64];
  };
  uint32_t base = 0;
  auto store = [&](const T (&x)[U], long long tile, int g0) {
    if (g0 == 0) base = so[tile];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 w = sm[tile * TILE_GROUPS + g0 + u];
      const uint32_t p = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(w >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)w, 0u));
      if (w & lane_bit) { if (NT) __builtin_nontemporal_store(x[u], (G1 T*)(dst + p)); else dst[p] = x[u]; }
      base += (uint32_t)__popcll(w);
    }
  };
  long long tile = first; int g0 = 0;
  if (tile >= n_tiles) return;
  T a[U], b[U];
  long long ta = tile; int ga = g0;
  load(a, ta, ga);
  auto advance = [&]() { g0 += U; if (g0 == TILE_GROUPS) { g0 = 0; tile += stride; } };
  advance();
  for (;;) {
    const bool more_b = tile < n_tiles;
    long long tb = tile; int gb = g0;
    if (more_b) { load(b, tb, gb); advance(); }
    store(a, ta, ga);
    if (!more_b) break;
    const bool more_a = tile < n_tiles;
    ta = tile; ga = g0;
    if (more_a) { load(a, ta, ga); advance(); }
    store(b, tb, gb);
    if (!more_a) break;
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void compact_direct(const Args args) {
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  int col = 0;
  while (col < 3 && g >= args.wave_begin[col + 1]) col++;
  col = __builtin_amdgcn_readfirstlane(col);
  const int first = g - args.wave_begin[col], stride = args.wave_begin[col + 1] - args.wave_begin[col];
  if (first >= stride) return;
  if (args.cols[col].width == 8) stream<u64, U, NT>((const u64*)args.cols[col].src, (u64*)args.cols[col].dst, args.masks, args.tile_offsets, first, stride, args.n_tiles, lane);
  else stream<uint32_t, U, NT>((const uint32_t*)args.cols[col].src, (uint32_t*)args.cols[col].dst, args.masks, args.tile_offsets, first, stride, args.n_tiles, lane);
}

template <int U, bool NT>
void run(Args args, int cus, long long n, long long selected, const uint32_t* d_check, long long want_last) {
  for (int waves_per_cu : {16, 24, 32, 48}) {
    const int total = cus * waves_per_cu;  // dealt to the columns by width: 8 8 4 4
    args.wave_begin[0] = 0; args.wave_begin[1] = total / 3; args.wave_begin[2] = 2 * (total / 3); args.wave_begin[3] = args.wave_begin[2] + total / 6; args.wave_begin[4] = args.wave_begin[3] + total / 6;
    const int grid = (args.wave_begin[4] + 3) / 4;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((compact_direct<U, NT>), dim3(grid), dim3(256), 0, 0, args);
    CHECK(hipEventRecord(a));
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL((compact_direct<U, NT>), dim3(grid), dim3(256), 0, 0, args);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double bytes = 24.0 * n + 24.0 * selected + n / 8.0 * 4;
    uint32_t last = 0; CHECK(hipMemcpy(&last, d_check + selected - 1, 4, hipMemcpyDeviceToHost));
    printf("U=%2d nt=%d waves/CU=%2d  %.4f ms  %.0f GB/s  %s\n", U, (int)NT, waves_per_cu, ms, bytes / ms / 1e6, (long long)last == want_last ? "ok" : "WRONG");
  }
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const long long n_tiles = 48829, n = n_tiles * TILE_GROUPS * 64, n_words = n / 64;
  u64* masks; uint32_t* offsets;
  CHECK(hipMalloc(&masks, n_words * 8)); CHECK(hipMalloc(&offsets, n_tiles * 4));
  void *src[4], *dst[4]; const int width[4] = {8, 8, 4, 4};
  for (int c = 0; c < 4; c++) {
    CHECK(hipMalloc(&src[c], (size_t)n * width[c])); CHECK(hipMalloc(&dst[c], (size_t)n * width[c]));
    if (width[c] == 8) hipLaunchKernelGGL(fill_iota<u64>, dim3(2048), dim3(256), 0, 0, (u64*)src[c], n);
    else hipLaunchKernelGGL(fill_iota<uint32_t>, dim3(2048), dim3(256), 0, 0, (uint32_t*)src[c], n);
  }
  printf("# %s CUs=%d, %lld rows x (8 + 8 + 4 + 4) bytes\n", prop.name, prop.multiProcessorCount, n);
  for (int mode = 0; mode < 3; mode++) {
    hipLaunchKernelGGL(fill_masks, dim3(2048), dim3(256), 0, 0, masks, n_words, mode);
    std::vector<u64> h(n_words); CHECK(hipMemcpy(h.data(), masks, n_words * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> off(n_tiles); long long total = 0, last_row = -1;
    for (long long t = 0; t < n_tiles; t++) { off[t] = (uint32_t)total; for (int g = 0; g < TILE_GROUPS; g++) total += __builtin_popcountll(h[t * TILE_GROUPS + g]); }
    for (long long w = n_words - 1; w >= 0; w--) if (h[w]) { last_row = w * 64 + 63 - __builtin_clzll(h[w]); break; }
    CHECK(hipMemcpy(offsets, off.data(), n_tiles * 4, hipMemcpyHostToDevice));
    Args args{}; for (int c = 0; c < 4; c++) args.cols[c] = Col{src[c], dst[c], width[c]};
    args.masks = masks; args.tile_offsets = offsets; args.n_tiles = n_tiles;
    printf("## %.1f %% selected\n", 100.0 * total / n);
    run<8, true>(args, prop.multiProcessorCount, n, total, (const uint32_t*)dst[3], last_row);
    run<8, false>(args, prop.multiProcessorCount, n, total, (const uint32_t*)dst[3], last_row);
    if (mode == 1) { run<4, true>(args, prop.multiProcessorCount, n, total, (const uint32_t*)dst[3], last_row); run<16, true>(args, prop.multiProcessorCount, n, total, (const uint32_t*)dst[3], last_row); }
  }
  return 0;
}
