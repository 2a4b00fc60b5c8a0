// Measurement aid (round 5): can the COLUMN STREAM of the high-cardinality scan (cfg 5: 136 B/row, sequential) and its
// SCATTERED operations (one 16-byte load of the home entry + one atomicAdd(double) per row, anywhere in a 1 GB table) overlap on this GPU,
// or do they queue for the same thing?  Round 4 measured stream alone 21 ps/row, load + atomic alone 44 ps/row, the product kernel 69.
//   A  stream alone                      B  scattered operations alone
//   M  both in ONE launch, a workgroup is either a streamer or an operator (each kind has its own fixed amount of work: n rows)
//   F  fused the way the product kernel is: a wave streams a tile, derives 4 slots per lane, loads the 4 entries, adds
//   P  fused + software pipelined: tile t's entries are requested before tile t + 1's columns, its atomics issued behind them
// M is the ideal any wave specialisation inside a workgroup (producer waves streaming into an LDS ring, consumer waves probing) could
// reach: both kinds of work in flight all the time, neither waiting for the other, no ring to maintain.
// If M ≈ max(A, B) the two overlap and F → P / specialised waves are worth building into fdb_hash_kernel; if M ≈ A + B they queue for
// the same thing and no restructuring of the kernel changes the sum.
// RESULT (profiles/round5_overlap_probe.txt): A 20.5, B 44.0, M 64.5–67.7, F 70.4, P 72.5 ps/row — additive.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/overlap_probe && tools/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
constexpr int NCOL = 34;  // 32 index columns + the 8-byte value column counted as two

__device__ __forceinline__ u64 mix(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

template <typename F>
float timed(F f, int reps = 5) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
  return best;
}

struct Args {
  const u32x4* cols; u64 col_stride16; u64 n_rows;   // stream
  u64* table; u64 mask;                              // 32-byte entries, mask = entries - 1
  u64 n_ops;                                         // scattered operations (role B of mode M)
  u64* sink;
  int grid_a;                                        // mode M: workgroups [0, grid_a) stream, the rest operate
};

// one tile = 256 threads × 4 rows; returns a per-lane fold of everything loaded (4 words: one per row of the lane)
__device__ __forceinline__ void stream_tile(const Args& a, u64 row0, u32x4& fold) {
  fold = u32x4{0, 0, 0, 0};
  for (int c0 = 0; c0 < NCOL; c0 += 8) {
    u32x4 v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = (c0 + c < NCOL) ? __builtin_nontemporal_load(a.cols + (u64)(c0 + c) * a.col_stride16 + row0 / 4) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 8; c++) fold += v[c] * (u32)(0x9E3779B1u + 2 * (c0 + c));
    asm volatile("" : "+v"(fold));
  }
}

__device__ __forceinline__ void ops4(const Args& a, const u64 (&idx)[4], u64& acc) {
  u64x2 v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const u64x2*>(a.table + idx[k] * 4);
#pragma unroll
  for (int k = 0; k < 4; k++) acc += v[k].x ^ v[k].y;
#pragma unroll
  for (int k = 0; k < 4; k++) atomicAdd(reinterpret_cast<double*>(a.table + idx[k] * 4 + 3), 1.0);
}

__global__ __launch_bounds__(256) void stream_kernel(const Args a) {
  const u64 n_tiles = (a.n_rows + 1023) / 1024;
  u32 extra = 0;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const u64 row0 = t * 1024 + (u64)threadIdx.x * 4;
    if (row0 >= a.n_rows) continue;
    u32x4 f; stream_tile(a, row0, f); extra += f.x ^ f.y ^ f.z ^ f.w;
  }
  if (extra == 0x12345u) *a.sink = extra;
}

__global__ __launch_bounds__(256) void ops_kernel(const Args a) {
  const u64 n_tiles = (a.n_ops + 1023) / 1024;
  u64 acc = 0;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const u64 op0 = t * 1024 + (u64)threadIdx.x * 4;
    u64 idx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) idx[k] = mix((op0 + k) * 0x9E3779B97F4A7C15ULL + 12345) & a.mask;
    ops4(a, idx, acc);
  }
  if (acc == 0x1234567u) *a.sink = acc;
}

__global__ __launch_bounds__(256) void mixed_kernel(const Args a) {
  if ((int)blockIdx.x < a.grid_a) {
    const u64 n_tiles = (a.n_rows + 1023) / 1024;
    u32 extra = 0;
    for (u64 t = blockIdx.x; t < n_tiles; t += a.grid_a) {
      const u64 row0 = t * 1024 + (u64)threadIdx.x * 4;
      if (row0 >= a.n_rows) continue;
      u32x4 f; stream_tile(a, row0, f); extra += f.x ^ f.y ^ f.z ^ f.w;
    }
    if (extra == 0x12345u) *a.sink = extra;
  } else {
    const u64 n_tiles = (a.n_ops + 1023) / 1024, gb = gridDim.x - a.grid_a;
    u64 acc = 0;
    for (u64 t = blockIdx.x - a.grid_a; t < n_tiles; t += gb) {
      const u64 op0 = t * 1024 + (u64)threadIdx.x * 4;
      u64 idx[4];
#pragma unroll
      for (int k = 0; k < 4; k++) idx[k] = mix((op0 + k) * 0x9E3779B97F4A7C15ULL + 12345) & a.mask;
      ops4(a, idx, acc);
    }
    if (acc == 0x1234567u) *a.sink = acc;
  }
}

// the product kernel's shape: stream → slots → 4 entry loads → 4 atomics, per wave and tile
template <int OPS>  // 0: stream only, 1: + loads, 2: + loads + atomics
__global__ __launch_bounds__(256) void fused_kernel(const Args a) {
  const u64 n_tiles = (a.n_rows + 1023) / 1024;
  u64 acc = 0;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const u64 row0 = t * 1024 + (u64)threadIdx.x * 4;
    if (row0 >= a.n_rows) continue;
    u32x4 f; stream_tile(a, row0, f);
    u64 idx[4] = {mix(f.x + (row0 + 0) * 0x9E3779B97F4A7C15ULL) & a.mask, mix(f.y + (row0 + 1) * 0x9E3779B97F4A7C15ULL) & a.mask, mix(f.z + (row0 + 2) * 0x9E3779B97F4A7C15ULL) & a.mask,
                  mix(f.w + (row0 + 3) * 0x9E3779B97F4A7C15ULL) & a.mask};
    if (OPS == 0) { acc += idx[0] ^ idx[1] ^ idx[2] ^ idx[3]; continue; }
    u64x2 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const u64x2*>(a.table + idx[k] * 4);
#pragma unroll
    for (int k = 0; k < 4; k++) acc += v[k].x ^ v[k].y;
    if (OPS == 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) atomicAdd(reinterpret_cast<double*>(a.table + idx[k] * 4 + 3), 1.0);
    }
  }
  if (acc == 0x1234567u) *a.sink = acc;
}

// software pipelined: the entries of tile t are requested, then tile t + 1's columns are streamed (the requests are in flight
// behind them), then tile t's atomics go out while tile t + 1's slots are computed
__global__ __launch_bounds__(256) void pipelined_kernel(const Args a) {
  const u64 n_tiles = (a.n_rows + 1023) / 1024;
  u64 acc = 0;
  u64 idx[4]; bool have = false;
  for (u64 t = blockIdx.x; ; t += gridDim.x) {
    const bool more = t < n_tiles && t * 1024 + (u64)threadIdx.x * 4 < a.n_rows;
    u64x2 v[4];
    if (have) {
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const u64x2*>(a.table + idx[k] * 4);
    }
    u32x4 f = {0, 0, 0, 0};
    const u64 row0 = t * 1024 + (u64)threadIdx.x * 4;
    if (more) stream_tile(a, row0, f);
    if (have) {
#pragma unroll
      for (int k = 0; k < 4; k++) acc += v[k].x ^ v[k].y;
#pragma unroll
      for (int k = 0; k < 4; k++) atomicAdd(reinterpret_cast<double*>(a.table + idx[k] * 4 + 3), 1.0);
    }
    if (!more) break;
    idx[0] = mix(f.x + (row0 + 0) * 0x9E3779B97F4A7C15ULL) & a.mask; idx[1] = mix(f.y + (row0 + 1) * 0x9E3779B97F4A7C15ULL) & a.mask;
    idx[2] = mix(f.z + (row0 + 2) * 0x9E3779B97F4A7C15ULL) & a.mask; idx[3] = mix(f.w + (row0 + 3) * 0x9E3779B97F4A7C15ULL) & a.mask;
    have = true;
  }
  if (acc == 0x1234567u) *a.sink = acc;
}


// ---- C: how fast is ONE counter? every wave draws `per_wave` tickets from the same 8-byte word with a returning atomicAdd (one lane
// per wave, the wave waits for the answer — what a wave that appends its new groups to a dense log would do once per tile)
__global__ __launch_bounds__(256) void counter_kernel(u64* counter, int per_wave, u64* sink) {
  u64 acc = 0;
  for (int i = 0; i < per_wave; i++) {
    u64 t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(counter, 22ull);
    t = __shfl(t, 0, 64);
    acc += t;
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0x1234567u) *sink = acc;
}

// ---- S: 144-byte key tuples (nine 16-byte stores from the inserting lane, as fdb_hash_kernel writes them), 35 % of the lanes insert:
// at slot × 144 of a table-sized key store (today), or at consecutive positions of a dense log (one counter draw per wave)
template <bool DENSE>
__global__ __launch_bounds__(256) void tuples_kernel(u32x4* keys, u64 mask, u64 n_rows, u64* counter, u64* sink) {
  const u64 n_tiles = (n_rows + 255) / 256;
  const int lane = threadIdx.x & 63;
  for (u64 t = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += (u64)gridDim.x * 4) {
    const u64 row = t * 256 + (u64)lane * 4;
    const u64 hsh = mix(row * 0x9E3779B97F4A7C15ULL + 777);
    const bool ins = (hsh & 1023) < 358;  // 35 %
    const unsigned long long b = __ballot(ins);
    if (b == 0ull) continue;
    u64 pos;
    if (DENSE) {
      u64 base = 0;
      if (lane == __builtin_ctzll(b)) base = atomicAdd(counter, (u64)__popcll(b));
      base = __shfl(base, __builtin_ctzll(b), 64);
      pos = base + (u64)__popcll(b & ((1ull << lane) - 1ull));
    } else {
      pos = (hsh >> 10) & mask;
    }
    if (ins) {
      u32x4* d = keys + pos * 9;
#pragma unroll
      for (int w = 0; w < 9; w++) d[w] = u32x4{(u32)row, (u32)w, (u32)hsh, 7u};
    }
  }
}

int main(int argc, char** argv) {
  const u64 n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  const u64 table_mb = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1024;
  const char* only = argc > 3 ? argv[3] : "ABMF";
  auto want = [&](char c) { return strchr(only, c) != nullptr; };
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, %llu rows, table %llu MB of 32-byte entries\n", prop.name, cus, n, table_mb);
  Args a{};
  a.n_rows = n; a.n_ops = n;
  a.col_stride16 = (n + 3) / 4 + 64;
  u32x4* cols; CHECK(hipMalloc(&cols, a.col_stride16 * 16 * NCOL)); CHECK(hipMemset(cols, 1, a.col_stride16 * 16 * NCOL));
  a.cols = cols;
  const u64 entries = table_mb * 1024 * 1024 / 32;
  CHECK(hipMalloc(&a.table, entries * 32)); CHECK(hipMemset(a.table, 0, entries * 32));
  a.mask = entries - 1;
  CHECK(hipMalloc(&a.sink, 8));
  if (want('A')) for (int per_cu : {4, 8}) {
    const int g = cus * per_cu;
    const float A = timed([&] { hipLaunchKernelGGL(stream_kernel, dim3(g), dim3(256), 0, 0, a); });
    const float B = timed([&] { hipLaunchKernelGGL(ops_kernel, dim3(g), dim3(256), 0, 0, a); });
    printf("A stream alone, %d workgroups/CU:                      %7.3f ms = %5.1f ps/row (%.2f TB/s)\n", per_cu, A, A * 1e9 / n, n * 136.0 / A / 1e9);
    printf("B load + atomicAdd alone, %d workgroups/CU:            %7.3f ms = %5.1f ps/row\n", per_cu, B, B * 1e9 / n);
  }
  if (want('M')) for (int per_cu : {8}) {
    for (int share_a : {2, 3, 4, 5, 6}) {  // eighths of the grid that stream
      Args m = a; const int g = cus * per_cu; m.grid_a = g * share_a / 8;
      const float M = timed([&] { hipLaunchKernelGGL(mixed_kernel, dim3(g), dim3(256), 0, 0, m); });
      printf("M both, %d workgroups/CU, %d/8 of them stream:          %7.3f ms = %5.1f ps/row\n", per_cu, share_a, M, M * 1e9 / n);
    }
  }
  if (want('F')) for (int per_cu : {4, 8}) {
    const int g = cus * per_cu;
    const float F0 = timed([&] { hipLaunchKernelGGL(fused_kernel<0>, dim3(g), dim3(256), 0, 0, a); });
    const float F1 = timed([&] { hipLaunchKernelGGL(fused_kernel<1>, dim3(g), dim3(256), 0, 0, a); });
    const float F2 = timed([&] { hipLaunchKernelGGL(fused_kernel<2>, dim3(g), dim3(256), 0, 0, a); });
    const float P = timed([&] { hipLaunchKernelGGL(pipelined_kernel, dim3(g), dim3(256), 0, 0, a); });
    printf("F fused, %d workgroups/CU: stream + slots %7.3f ms, + entry loads %7.3f ms, + atomics %7.3f ms = %5.1f ps/row\n", per_cu, F0, F1, F2, F2 * 1e9 / n);
    printf("P fused + software pipelined, %d workgroups/CU:         %7.3f ms = %5.1f ps/row\n", per_cu, P, P * 1e9 / n);
  }

  if (want('C')) {
    u64* counter; CHECK(hipMalloc(&counter, 256)); CHECK(hipMemset(counter, 0, 256));
    for (int per_cu : {4, 8}) for (int per_wave : {16, 64}) {
      const int g = cus * per_cu;
      const float ms = timed([&] { hipLaunchKernelGGL(counter_kernel, dim3(g), dim3(256), 0, 0, counter, per_wave, a.sink); });
      const double n_at = (double)g * 4 * per_wave;
      printf("C one counter, %d workgroups/CU, %d draws per wave: %8.0f draws in %7.3f ms = %6.1f ns per draw (%5.1f M draws/s)\n", per_cu, per_wave, n_at, ms, ms * 1e6 / n_at, n_at / ms / 1e3);
    }
    CHECK(hipFree(counter));
  }
  if (want('S')) {
    const u64 slots = 1ull << 26;  // a 64 M-slot key store: 9.7 GB (only the touched tuples are ever written)
    u32x4* keys; CHECK(hipMalloc(&keys, slots * 144));
    u64* counter; CHECK(hipMalloc(&counter, 256));
    const u64 rows = 96ull << 20;  // 24 M lanes × 4 rows: the insert-heavy launch of cfg 5 (≈ 8.4 M inserts)
    const int g = cus * 8;
    const float s0 = timed([&] { hipLaunchKernelGGL(tuples_kernel<false>, dim3(g), dim3(256), 0, 0, keys, slots - 1, rows, counter, a.sink); });
    const float s1 = timed([&] { CHECK(hipMemsetAsync(counter, 0, 8)); hipLaunchKernelGGL(tuples_kernel<true>, dim3(g), dim3(256), 0, 0, keys, slots - 1, rows, counter, a.sink); });
    u64 n_ins = 0; CHECK(hipMemcpy(&n_ins, counter, 8, hipMemcpyDeviceToHost));
    printf("S key tuples of %.1f M inserts (9 x 16-byte stores each): at slot x 144 of a 64 M-slot store %7.3f ms, at consecutive positions of a dense log %7.3f ms\n", n_ins / 1e6, s0, s1);
    CHECK(hipFree(keys)); CHECK(hipFree(counter));
  }
  return 0;
}
