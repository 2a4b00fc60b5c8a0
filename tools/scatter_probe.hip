// Measurement aid (round 4): what do SCATTERED memory operations cost on this GPU, and does locality change it?
//
// The high-cardinality scan (cfg 5) is bound by ≈40 G scattered single-sector operations per second (one 16-byte load of a table
// entry + one atomic per row). Two questions decide how a partition-by-fingerprint front end has to be built:
//   L  does the rate depend on the SPAN the operations are spread over (whole 4 GB table / a 1 MB window per workgroup — what a
//      workgroup that owns one partition's slice of the table would see), and on the kind of operation (16-byte load, returning /
//      non-returning atomic, plain read-modify-write, 32-byte store)?
//   S  what does a write-combining scatter cost — a tile of rows counting-sorted by partition in LDS and appended, run by run, to
//      workgroup-private chunks of every partition, so that one store INSTRUCTION writes whole 128-byte lines — against the naive
//      one-store-per-row scatter (2.6 ms per 100 M rows, profiles/round3_part_probe.txt)?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/scatter_probe.hip -o tools/scatter_probe && tools/scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 mix(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

template <typename F>
float timed(F f, int reps = 3) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
  return best;
}

// ---- L: random operations on 32-byte entries -------------------------------------------------------------------------------------
// MODE bits: 1 = 16-byte load of the entry's first half (consumed), 2 = atomicAdd(double) on word 3 (no return), 4 = plain 8-byte
// load + store of word 3 (what exclusive ownership allows), 8 = 32-byte store of the whole entry, 16 = returning atomicAdd on word 2
// window_entries = 0: every operation anywhere in [0, n_entries); else workgroup b works in window (b % n_windows)
template <int MODE>
__global__ __launch_bounds__(256) void rand_ops_kernel(u64* table, u64 n_entries, u64 window_entries, u64 ops_per_thread, u64* sink) {
  const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 base = 0, span = n_entries;
  if (window_entries != 0) { const u64 nw = n_entries / window_entries; base = (blockIdx.x % nw) * window_entries; span = window_entries; }
  u64 acc = 0;
  for (u64 it = 0; it < ops_per_thread; it += 4) {
    u64 idx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) idx[k] = base + mix((tid * ops_per_thread + it + k) * 0x9E3779B97F4A7C15ULL + 12345) % span;
    if (MODE & 1) {
      u64x2 v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const u64x2*>(table + idx[k] * 4);
#pragma unroll
      for (int k = 0; k < 4; k++) acc += v[k].x ^ v[k].y;
    }
    if (MODE & 16) {
#pragma unroll
      for (int k = 0; k < 4; k++) acc += atomicAdd(table + idx[k] * 4 + 2, 1ull);
    }
    if (MODE & 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) atomicAdd(reinterpret_cast<double*>(table + idx[k] * 4 + 3), 1.0);
    }
    if (MODE & 4) {
      u64 w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) w[k] = table[idx[k] * 4 + 3];
#pragma unroll
      for (int k = 0; k < 4; k++) table[idx[k] * 4 + 3] = w[k] + 1;
    }
    if (MODE & 8) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        u64x2* e = reinterpret_cast<u64x2*>(table + idx[k] * 4);
        e[0] = u64x2{idx[k], it}; e[1] = u64x2{tid, 1};
      }
    }
  }
  if (acc == 0x1234567u) *sink = acc;
}

template <int MODE>
void run_rand(const char* what, u64* table, u64 n_entries, u64 window_entries, u64 total_ops, int grid, u64* sink) {
  const u64 per_thread = (total_ops / ((u64)grid * 256) + 3) & ~3ull;
  const float ms = timed([&] { hipLaunchKernelGGL((rand_ops_kernel<MODE>), dim3(grid), dim3(256), 0, 0, table, n_entries, window_entries, per_thread, sink); });
  const double n = (double)per_thread * grid * 256;
  printf("L %-46s span %8.1f MB window %8.3f MB: %7.3f ms per %.0f M rows = %6.1f G rows/s\n", what, n_entries * 32 / 1e6, (window_entries ? window_entries : n_entries) * 32 / 1e6, ms, n / 1e6,
         n / ms / 1e6);
}

// ---- S: write-combining scatter ---------------------------------------------------------------------------------------------------
// A workgroup of BLK threads takes tiles of BLK × 4 rows. Per tile: partition id of every row (top PBITS of its fingerprint) →
// LDS histogram (the returning atomic gives the row's rank inside its partition) → exclusive prefix sum over the partitions →
// every record lands in an LDS staging area, sorted by partition → the staged records leave as RUNS: a partition's run is appended
// to the workgroup's private chunk of that partition (chunks of CHUNK records are handed out by one global atomic per chunk; a
// chunk is filled by ONE workgroup, so the partial lines a run leaves behind are completed by the same CU's next tiles, in the
// same XCD's L2). The copy loop moves 16 bytes per lane: consecutive lanes write consecutive 16-byte pieces, so one store
// instruction covers whole lines wherever a run is long enough.
// Layout: chunk_dir[p * max_chunks + k] = fill count of partition p's k-th chunk (written when the chunk is closed),
// recs[(p * max_chunks + k) * CHUNK + i]. A partition that runs out of chunks sends its records to the overflow list.
struct ScatterArgs {
  u64 n, n_groups;
  u64* recs;          // [P][max_chunks][CHUNK] records of RW 8-byte words
  u32* chunk_cursor;  // [P] next chunk index of the partition
  u32* chunk_fill;    // [P][max_chunks]
  u64* overflow;      // records that found no chunk
  u64* overflow_cursor;
  u32 max_chunks;
};

template <int PBITS, int BLK, int RW, int CHUNK, bool STREAM>
__global__ __launch_bounds__(BLK) void scatter_kernel(const ScatterArgs a, const u32x4* __restrict__ cols, u64 col_stride16) {
  constexpr int P = 1 << PBITS, TILE = BLK * 4;
  extern __shared__ __align__(16) unsigned char smem[];
  u64* stage = reinterpret_cast<u64*>(smem);                    // TILE records
  u32* cnt = reinterpret_cast<u32*>(stage + (size_t)TILE * RW);  // P   (per tile: histogram, then exclusive offsets)
  u32* run = cnt + P;                                           // P   (per tile: the partition's count)
  u32* cpos = run + P;                                          // P   records already in the workgroup's open chunk of p (CHUNK = none open)
  u32* cidx = cpos + P;                                         // P   index of that chunk
  __shared__ u32 wsum[BLK / 64];
  const u32 tid = threadIdx.x;
  for (int p = tid; p < P; p += BLK) { cpos[p] = CHUNK; cidx[p] = 0; }
  const u64 n_tiles = (a.n + TILE - 1) / TILE;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    for (int p = tid; p < P; p += BLK) cnt[p] = 0;
    __syncthreads();
    u64 lo[4], hi[4];
    u32 part[4], rank[4];
    const u64 row0 = t * TILE + (u64)tid * 4;
    u32 extra = 0;
    if (STREAM) {  // the column stream of cfg 5: 32 index columns + one 8-byte column, 16-byte loads, 8 in flight
      for (int c0 = 0; c0 < 34; c0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = (c0 + c < 34 && row0 < a.n) ? __builtin_nontemporal_load(cols + (u64)(c0 + c) * col_stride16 + row0 / 4) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 8; c++) extra += v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 i = row0 + k;
      part[k] = 0xFFFFFFFFu;
      if (i < a.n) {
        const u64 g = mix(i * 0x9E3779B97F4A7C15ULL + 1) % a.n_groups;
        lo[k] = mix(g + 0x1234567ULL) | 1ull;
        hi[k] = mix(g * 0xD6E8FEB86659FD93ULL + 7) + (extra == 0x12345u);
        part[k] = (u32)(lo[k] >> (64 - PBITS));
        rank[k] = atomicAdd(&cnt[part[k]], 1u);
      }
    }
    __syncthreads();
    // exclusive prefix sum of cnt[0 .. P): every thread owns P / BLK consecutive partitions
    {
      constexpr int PER = (P + BLK - 1) / BLK;
      u32 local[PER], sum = 0;
#pragma unroll
      for (int j = 0; j < PER; j++) { const int p = tid * PER + j; local[j] = p < P ? cnt[p] : 0; sum += local[j]; }
      u32 incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const u32 v = __shfl_up(incl, off, 64); if ((int)(tid & 63) >= off) incl += v; }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      u32 wbase = 0;
      for (int w = 0; w < (int)(tid >> 6); w++) wbase += wsum[w];
      u32 ex = wbase + incl - sum;
#pragma unroll
      for (int j = 0; j < PER; j++) { const int p = tid * PER + j; if (p < P) { run[p] = local[j]; cnt[p] = ex; ex += local[j]; } }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (part[k] == 0xFFFFFFFFu) continue;
      u64* r = stage + (size_t)(cnt[part[k]] + rank[k]) * RW;
      r[0] = lo[k]; r[1] = hi[k]; r[2] = 0x3FF0000000000000ull; r[3] = row0 + k;
      for (int w = 4; w < RW; w++) r[w] = lo[k] ^ w;
    }
    __syncthreads();
    // runs → chunks. A wave takes partitions round-robin; lanes copy 16-byte pieces.
    const int wave = tid >> 6, lane = tid & 63;
    for (int p = wave; p < P; p += BLK / 64) {
      u32 left = run[p];
      if (left == 0) continue;
      u32 src = cnt[p];
      u32 pos = cpos[p], ci = cidx[p];
      while (left > 0) {
        if (pos == CHUNK) {  // open a new chunk (lane 0 asks, everybody learns)
          u32 nc = 0;
          if (lane == 0) nc = atomicAdd(&a.chunk_cursor[p], 1u);
          ci = __shfl(nc, 0, 64);
          pos = 0;
        }
        const u32 room = (u32)CHUNK - pos;
        const u32 take = left < room ? left : room;
        const u64x2* s = reinterpret_cast<const u64x2*>(stage + (size_t)src * RW);
        if (ci < a.max_chunks) {
          u64x2* dst = reinterpret_cast<u64x2*>(a.recs + ((size_t)((u64)p * a.max_chunks + ci) * CHUNK + pos) * RW);
          for (u32 q = lane; q < take * (RW / 2); q += 64) dst[q] = s[q];
        } else {  // no chunk left: the overflow list (rare; one atomic per run)
          u64 ob = 0;
          if (lane == 0) ob = atomicAdd(a.overflow_cursor, (u64)take);
          ob = __shfl(ob, 0, 64);
          u64x2* dst = reinterpret_cast<u64x2*>(a.overflow + ob * RW);
          for (u32 q = lane; q < take * (RW / 2); q += 64) dst[q] = s[q];
        }
        pos += take; src += take; left -= take;
        if (pos == CHUNK && ci < a.max_chunks && lane == 0) a.chunk_fill[(u64)p * a.max_chunks + ci] = CHUNK;
      }
      if (lane == 0) { cpos[p] = pos; cidx[p] = ci; }
    }
    __syncthreads();
  }
  // close the open chunks
  for (int p = tid; p < P; p += BLK)
    if (cpos[p] != CHUNK && cidx[p] < a.max_chunks) a.chunk_fill[(u64)p * a.max_chunks + cidx[p]] = cpos[p];
}

// ---- D: direct scatter into workgroup-private chunks (no LDS staging) ---------------------------------------------------------------
// Every row's record goes straight from registers to the workgroup's open chunk of its partition: position from a returning LDS
// atomic on the workgroup's cursor of that partition; the lane that draws position CHUNK opens the next chunk (one global atomic).
// A line is completed by the same workgroup over its next tiles; the partial lines live in L2 / the Infinity Cache meanwhile
// (open lines in total: workgroups × P × 128 B).
template <int PBITS, int BLK, int RW, int CHUNK, bool STREAM>
__global__ __launch_bounds__(BLK) void direct_scatter_kernel(const ScatterArgs a, const u32x4* __restrict__ cols, u64 col_stride16) {
  constexpr int P = 1 << PBITS, TILE = BLK * 4;
  extern __shared__ __align__(16) unsigned char smem[];
  u32* cur = reinterpret_cast<u32*>(smem);  // P: records claimed in the open chunk (≥ CHUNK: closed, being replaced)
  u32* cidx = cur + P;                      // P: index of the open chunk
  const u32 tid = threadIdx.x;
  for (int p = tid; p < P; p += BLK) {
    const u32 c = atomicAdd(&a.chunk_cursor[p], 1u);
    cidx[p] = c; cur[p] = 0;
  }
  __syncthreads();
  const u64 n_tiles = (a.n + TILE - 1) / TILE;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const u64 row0 = t * TILE + (u64)tid * 4;
    u32 extra = 0;
    if (STREAM) {
      for (int c0 = 0; c0 < 34; c0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = (c0 + c < 34 && row0 < a.n) ? __builtin_nontemporal_load(cols + (u64)(c0 + c) * col_stride16 + row0 / 4) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 8; c++) extra += v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
      }
    }
    u64 lo[4], hi[4];
    u32 part[4];
    u32 pending = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 i = row0 + k;
      if (i < a.n) {
        const u64 g = mix(i * 0x9E3779B97F4A7C15ULL + 1) % a.n_groups;
        lo[k] = mix(g + 0x1234567ULL) | 1ull;
        hi[k] = mix(g * 0xD6E8FEB86659FD93ULL + 7) + (extra == 0x12345u);
        part[k] = (u32)(lo[k] >> (64 - PBITS));
        pending |= 1u << k;
      }
    }
    // claim + store; a lane whose partition's chunk is closed retries in the next round (never spins inside a round: the lane that
    // opens the next chunk may sit in the same wave)
    while (__any(pending != 0)) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!((pending >> k) & 1u)) continue;
        const u32 p = part[k];
        if (__hip_atomic_load(&cur[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (u32)CHUNK + 1) continue;  // closed: try again later
        const u32 pos = atomicAdd(&cur[p], 1u);
        u32 ci = __hip_atomic_load(&cidx[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pos > (u32)CHUNK) continue;
        u32 at = pos;
        if (pos == (u32)CHUNK) {  // this lane opens the next chunk and takes its first slot
          if (ci < a.max_chunks) a.chunk_fill[(u64)p * a.max_chunks + ci] = CHUNK;
          ci = atomicAdd(&a.chunk_cursor[p], 1u);
          __hip_atomic_store(&cidx[p], ci, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __hip_atomic_store(&cur[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          at = 0;
        }
        u64* r;
        if (ci < a.max_chunks) r = a.recs + ((size_t)((u64)p * a.max_chunks + ci) * CHUNK + at) * RW;
        else r = a.overflow + atomicAdd(a.overflow_cursor, 1ull) * RW;
        reinterpret_cast<u64x2*>(r)[0] = u64x2{lo[k], hi[k]};
        reinterpret_cast<u64x2*>(r)[1] = u64x2{0x3FF0000000000000ull, row0 + k};
        for (int w = 4; w < RW; w += 2) reinterpret_cast<u64x2*>(r)[w / 2] = u64x2{lo[k] ^ w, hi[k]};
        pending &= ~(1u << k);
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < P; p += BLK) {
    const u32 c = cur[p] > (u32)CHUNK ? (u32)CHUNK : cur[p];
    if (cidx[p] < a.max_chunks) a.chunk_fill[(u64)p * a.max_chunks + cidx[p]] = c;
  }
}

// naive: one 32-byte store per row at a position reserved per (tile, partition) — what part_probe's B3 measured
template <int PBITS, int RW>
__global__ __launch_bounds__(256) void naive_scatter_kernel(u64 n, u64 n_groups, u64* cursor, u64* out, u64 cap_per_part) {
  constexpr int P = 1 << PBITS, TILE = 4096;
  __shared__ u32 cnt[P];
  __shared__ u64 base[P];
  const u64 n_tiles = (n + TILE - 1) / TILE;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    for (int k = threadIdx.x; k < P; k += 256) cnt[k] = 0;
    __syncthreads();
    u64 lo[TILE / 256], hi[TILE / 256];
    u32 off[TILE / 256];
#pragma unroll
    for (int u = 0; u < TILE / 256; u++) {
      const u64 i = t * TILE + (u64)u * 256 + threadIdx.x;
      off[u] = 0xFFFFFFFFu;
      if (i < n) {
        const u64 g = mix(i * 0x9E3779B97F4A7C15ULL + 1) % n_groups;
        lo[u] = mix(g + 0x1234567ULL) | 1ull; hi[u] = mix(g * 0xD6E8FEB86659FD93ULL + 7);
        off[u] = atomicAdd(&cnt[lo[u] >> (64 - PBITS)], 1u);
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < P; k += 256) if (cnt[k]) base[k] = atomicAdd(&cursor[k], (u64)cnt[k]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TILE / 256; u++) {
      if (off[u] == 0xFFFFFFFFu) continue;
      const u32 p = (u32)(lo[u] >> (64 - PBITS));
      const u64 at = base[p] + off[u];
      if (at >= cap_per_part) continue;
      u64x2* r = reinterpret_cast<u64x2*>(out + ((u64)p * cap_per_part + at) * RW);
      r[0] = u64x2{lo[u], hi[u]}; r[1] = u64x2{0x3FF0000000000000ull, t * TILE + (u64)u * 256 + threadIdx.x};
    }
    __syncthreads();
  }
}

// check: every row arrived exactly once (sum of the row ids and of the counts over all chunks + overflow)
template <int RW>
__global__ void check_kernel(const u64* recs, const u32* chunk_fill, u64 n_chunks_total, int chunk, u64* out /* [count, sum of rows] */) {
  u64 c = 0, s = 0;
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n_chunks_total; k += (u64)gridDim.x * blockDim.x) {
    const u32 f = chunk_fill[k];
    for (u32 i = 0; i < f; i++) { c++; s += recs[(k * chunk + i) * RW + 3]; }
  }
  atomicAdd(&out[0], c); atomicAdd(&out[1], s);
}

template <int PBITS, int BLK, int RW, int CHUNK, bool STREAM>
void run_scatter(u64 n, u64 n_groups, int cus, const u32x4* cols, u64 col_stride16) {
  constexpr int P = 1 << PBITS, TILE = BLK * 4;
  ScatterArgs a;
  a.n = n; a.n_groups = n_groups;
  a.max_chunks = (u32)((n / P) * 5 / 4 / CHUNK + 256 + 2);  // the expected share + 25 % + one open chunk per workgroup
  const u64 n_chunks_total = (u64)P * a.max_chunks;
  CHECK(hipMalloc(&a.recs, n_chunks_total * CHUNK * RW * 8));
  CHECK(hipMalloc(&a.chunk_cursor, P * 4)); CHECK(hipMalloc(&a.chunk_fill, n_chunks_total * 4));
  CHECK(hipMalloc(&a.overflow, (n / 16 + 1024) * RW * 8)); CHECK(hipMalloc(&a.overflow_cursor, 8));
  const size_t lds = (size_t)TILE * RW * 8 + (size_t)P * 16;
  auto kern = scatter_kernel<PBITS, BLK, RW, CHUNK, STREAM>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 1;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLK, lds));
  const int grid = cus * (per_cu < 1 ? 1 : per_cu);
  const float ms_set = timed([&] { CHECK(hipMemsetAsync(a.chunk_cursor, 0, P * 4)); CHECK(hipMemsetAsync(a.chunk_fill, 0, n_chunks_total * 4)); CHECK(hipMemsetAsync(a.overflow_cursor, 0, 8)); });
  const float ms = timed([&] {
    CHECK(hipMemsetAsync(a.chunk_cursor, 0, P * 4)); CHECK(hipMemsetAsync(a.chunk_fill, 0, n_chunks_total * 4)); CHECK(hipMemsetAsync(a.overflow_cursor, 0, 8));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK), lds, 0, a, cols, col_stride16);
  });
  u64* d_out; CHECK(hipMalloc(&d_out, 16)); CHECK(hipMemset(d_out, 0, 16));
  hipLaunchKernelGGL((check_kernel<RW>), dim3(cus * 4), dim3(256), 0, 0, a.recs, a.chunk_fill, n_chunks_total, CHUNK, d_out);
  u64 out[2], ovf = 0; CHECK(hipMemcpy(out, d_out, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&ovf, a.overflow_cursor, 8, hipMemcpyDeviceToHost));
  const u64 want_sum = n * (n - 1) / 2;
  printf("S %s P %4d BLK %4d (tile %5d, %d/CU) record %2d B chunk %4d rec: %7.3f ms (− %.3f ms of memsets) per %.0f M rows; LDS %zu KB; rows in chunks %llu + overflow %llu %s\n",
         STREAM ? "stream+scatter" : "scatter only  ", P, BLK, TILE, per_cu, RW * 8, CHUNK, ms, ms_set, n / 1e6, lds / 1024, out[0], ovf,
         (out[0] + ovf == n && (ovf != 0 || out[1] == want_sum)) ? "OK" : "WRONG");
  CHECK(hipFree(a.recs)); CHECK(hipFree(a.chunk_cursor)); CHECK(hipFree(a.chunk_fill)); CHECK(hipFree(a.overflow)); CHECK(hipFree(a.overflow_cursor)); CHECK(hipFree(d_out));
}

template <int PBITS, int BLK, int RW, int CHUNK, bool STREAM>
void run_direct(u64 n, u64 n_groups, int cus, int wg_per_cu, const u32x4* cols, u64 col_stride16) {
  constexpr int P = 1 << PBITS;
  const int grid = cus * wg_per_cu;
  ScatterArgs a;
  a.n = n; a.n_groups = n_groups;
  a.max_chunks = (u32)((n / P) * 5 / 4 / CHUNK + grid + 2);
  const u64 n_chunks_total = (u64)P * a.max_chunks;
  CHECK(hipMalloc(&a.recs, n_chunks_total * CHUNK * RW * 8));
  CHECK(hipMalloc(&a.chunk_cursor, P * 4)); CHECK(hipMalloc(&a.chunk_fill, n_chunks_total * 4));
  CHECK(hipMalloc(&a.overflow, (n / 16 + 1024) * RW * 8)); CHECK(hipMalloc(&a.overflow_cursor, 8));
  const size_t lds = (size_t)P * 8;
  auto kern = direct_scatter_kernel<PBITS, BLK, RW, CHUNK, STREAM>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float ms_set = timed([&] { CHECK(hipMemsetAsync(a.chunk_cursor, 0, P * 4)); CHECK(hipMemsetAsync(a.chunk_fill, 0, n_chunks_total * 4)); CHECK(hipMemsetAsync(a.overflow_cursor, 0, 8)); });
  const float ms = timed([&] {
    CHECK(hipMemsetAsync(a.chunk_cursor, 0, P * 4)); CHECK(hipMemsetAsync(a.chunk_fill, 0, n_chunks_total * 4)); CHECK(hipMemsetAsync(a.overflow_cursor, 0, 8));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK), lds, 0, a, cols, col_stride16);
  });
  u64* d_out; CHECK(hipMalloc(&d_out, 16)); CHECK(hipMemset(d_out, 0, 16));
  hipLaunchKernelGGL((check_kernel<RW>), dim3(cus * 4), dim3(256), 0, 0, a.recs, a.chunk_fill, n_chunks_total, CHUNK, d_out);
  u64 out[2], ovf = 0; CHECK(hipMemcpy(out, d_out, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&ovf, a.overflow_cursor, 8, hipMemcpyDeviceToHost));
  const u64 want_sum = n * (n - 1) / 2;
  printf("D %s P %4d BLK %4d × %d/CU record %2d B chunk %3d rec (area %.1f GB, open lines %.0f MB): %7.3f ms (− %.3f ms of memsets) per %.0f M rows; rows in chunks %llu + overflow %llu %s\n",
         STREAM ? "stream+direct" : "direct only  ", P, BLK, wg_per_cu, RW * 8, CHUNK, n_chunks_total * CHUNK * RW * 8 / 1e9, (double)grid * P * 128 / 1e6, ms, ms_set, n / 1e6, out[0], ovf,
         (out[0] + ovf == n && (ovf != 0 || out[1] == want_sum)) ? "OK" : "WRONG");
  CHECK(hipFree(a.recs)); CHECK(hipFree(a.chunk_cursor)); CHECK(hipFree(a.chunk_fill)); CHECK(hipFree(a.overflow)); CHECK(hipFree(a.overflow_cursor)); CHECK(hipFree(d_out));
}

// ---- T: the pipeline that would be built: P1 = column stream + fingerprint + tile-sorted scatter + narrow tuple planes; 2A = per
// (partition, sub-range) LDS aggregation, the A sub-ranges of a partition read the same records on the same XCD (L2 serves the re-reads)
struct PipeArgs {
  u64 n, n_groups;
  u64* recs; u32* chunk_cursor; u32* chunk_fill; u64* overflow; u64* overflow_cursor; u32 max_chunks;
  u32x4* t_lo; u32x4* t_hi;  // narrow key tuples, two planes of 16 bytes, indexed by the record's t
};
template <int PBITS, int BLK, int CHUNK>
__global__ __launch_bounds__(BLK) void p1_kernel(const PipeArgs a, const u32x4* __restrict__ cols, u64 col_stride16) {
  constexpr int P = 1 << PBITS, TILE = BLK * 4, PER = (P + BLK - 1) / BLK;
  extern __shared__ __align__(16) unsigned char smem[];
  u64x2* stage = reinterpret_cast<u64x2*>(smem);                      // TILE × 2 pieces of 16 bytes
  u64* dst = reinterpret_cast<u64*>(stage + (size_t)TILE * 2);         // P: global record index of this tile's run of p (bit 63: overflow area)
  u32* cnt = reinterpret_cast<u32*>(dst + P);                          // P: histogram of the tile
  u32* off = cnt + P;                                                  // P: exclusive offsets
  u32* cpos = off + P;                                                 // P: records in the open chunk
  u32* cidx = cpos + P;                                                // P: its index
  unsigned short* spart = reinterpret_cast<unsigned short*>(cidx + P); // TILE: partition of a staged record
  __shared__ u32 wsum[BLK / 64];
  const u32 tid = threadIdx.x;
  for (int p = tid; p < P; p += BLK) { cpos[p] = CHUNK; cidx[p] = 0xFFFFFFFFu; }
  const u64 n_tiles = (a.n + TILE - 1) / TILE;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    for (int p = tid; p < P; p += BLK) cnt[p] = 0;
    __syncthreads();
    const u64 row0 = t * TILE + (u64)tid * 4;
    u32 extra = 0;
    u32x4 tl[4], th[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { tl[k] = u32x4{0, 0, 0, 0}; th[k] = u32x4{0, 0, 0, 0}; }
    for (int c0 = 0; c0 < 34; c0 += 8) {
      u32x4 v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = (c0 + c < 34 && row0 < a.n) ? __builtin_nontemporal_load(cols + (u64)(c0 + c) * col_stride16 + row0 / 4) : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 8; c++) {
        extra += v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
        // (stand-in for packing the column's 4 key ids into byte c of the 4 rows' tuples)
        const int cc = c0 + c;
        if (cc < 16) { tl[0][cc / 4] |= (v[c].x & 0xFFu) << (8 * (cc % 4)); tl[1][cc / 4] |= (v[c].y & 0xFFu) << (8 * (cc % 4)); tl[2][cc / 4] |= (v[c].z & 0xFFu) << (8 * (cc % 4)); tl[3][cc / 4] |= (v[c].w & 0xFFu) << (8 * (cc % 4)); }
        else if (cc < 32) { const int d = cc - 16; th[0][d / 4] |= (v[c].x & 0xFFu) << (8 * (d % 4)); th[1][d / 4] |= (v[c].y & 0xFFu) << (8 * (d % 4)); th[2][d / 4] |= (v[c].z & 0xFFu) << (8 * (d % 4)); th[3][d / 4] |= (v[c].w & 0xFFu) << (8 * (d % 4)); }
      }
    }
    u64 lo[4], hi[4];
    u32 part[4], rank[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 i = row0 + k;
      part[k] = 0xFFFFFFFFu;
      if (i < a.n) {
        const u64 g = mix(i * 0x9E3779B97F4A7C15ULL + 1) % a.n_groups;
        lo[k] = mix(g + 0x1234567ULL) | 1ull;
        hi[k] = mix(g * 0xD6E8FEB86659FD93ULL + 7) + (extra == 0x12345u);
        part[k] = (u32)(lo[k] >> (64 - PBITS));
        rank[k] = atomicAdd(&cnt[part[k]], 1u);
        // tuple planes: t = tile base + k × BLK + tid — for a fixed k consecutive lanes write consecutive 16-byte pieces
        const u64 tt = t * TILE + (u64)k * BLK + tid;
        __builtin_nontemporal_store(tl[k], a.t_lo + tt);
        __builtin_nontemporal_store(th[k], a.t_hi + tt);
      }
    }
    __syncthreads();
    {  // exclusive offsets + where every partition's run goes
      u32 local[PER], sum = 0;
#pragma unroll
      for (int j = 0; j < PER; j++) { const int p = tid * PER + j; local[j] = p < P ? cnt[p] : 0; sum += local[j]; }
      u32 incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((int)(tid & 63) >= o) incl += v; }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      u32 wbase = 0;
      for (int w = 0; w < (int)(tid >> 6); w++) wbase += wsum[w];
      u32 ex = wbase + incl - sum;
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int p = tid * PER + j;
        if (p >= P) continue;
        off[p] = ex; ex += local[j];
        const u32 c = local[j];
        if (c == 0) continue;
        u32 pos = cpos[p], ci = cidx[p];
        if (pos + c > (u32)CHUNK) {  // the run does not fit the open chunk: close it, take ceil(c / CHUNK) consecutive fresh ones
          if (ci != 0xFFFFFFFFu && ci < a.max_chunks) a.chunk_fill[(u64)p * a.max_chunks + ci] = pos;
          const u32 m = (c + CHUNK - 1) / CHUNK;
          ci = atomicAdd(&a.chunk_cursor[p], m);
          for (u32 x = 0; x + 1 < m; x++) if (ci + x < a.max_chunks) a.chunk_fill[(u64)p * a.max_chunks + ci + x] = CHUNK;
          if (ci + m <= a.max_chunks) { dst[p] = ((u64)p * a.max_chunks + ci) * CHUNK; ci += m - 1; pos = c - (m - 1) * CHUNK; }
          else { dst[p] = (1ull << 63) | atomicAdd(a.overflow_cursor, (u64)c); ci = 0xFFFFFFFFu; pos = CHUNK; }
        } else {
          dst[p] = ((u64)p * a.max_chunks + ci) * CHUNK + pos;
          pos += c;
        }
        cpos[p] = pos; cidx[p] = ci;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (part[k] == 0xFFFFFFFFu) continue;
      const u32 at = off[part[k]] + rank[k];
      stage[(size_t)at * 2] = u64x2{lo[k], hi[k]};
      stage[(size_t)at * 2 + 1] = u64x2{0x3FF0000000000000ull, t * TILE + (u64)k * BLK + tid};
      spart[at] = (unsigned short)part[k];
    }
    __syncthreads();
    const u64 left = a.n - t * TILE;
    const u32 total = left < (u64)TILE ? (u32)left : (u32)TILE;
    for (u32 q = tid; q < total * 2; q += BLK) {
      const u32 rec = q >> 1;
      const u32 p = spart[rec];
      const u64 d = dst[p];
      u64x2* base = (d >> 63) ? reinterpret_cast<u64x2*>(a.overflow) : reinterpret_cast<u64x2*>(a.recs);
      base[((d & ~(1ull << 63)) + (rec - off[p])) * 2 + (q & 1)] = stage[q];
    }
    __syncthreads();
  }
  for (int p = tid; p < P; p += BLK)
    if (cidx[p] != 0xFFFFFFFFu && cidx[p] < a.max_chunks) a.chunk_fill[(u64)p * a.max_chunks + cidx[p]] = cpos[p];
}

// 2A: workgroup b → XCD b % 8; the A sub-ranges of a partition sit on one XCD next to each other in dispatch order
template <int PBITS, int ABITS, int SLOTS, int CHUNK>
__global__ __launch_bounds__(1024) void p2a_kernel(const PipeArgs a, u64* out, u64* out_cursor, u32* leftover) {
  constexpr int A = 1 << ABITS;
  extern __shared__ __align__(16) unsigned char smem[];
  u64* t_lo = reinterpret_cast<u64*>(smem);
  u64* t_hi = t_lo + SLOTS;
  double* t_sum = reinterpret_cast<double*>(t_hi + SLOTS);
  u64* t_aux = reinterpret_cast<u64*>(t_sum + SLOTS);  // representative t (low 40 bits) | count << 40 … kept simple: first t
  __shared__ u32 n_here, wcur;
  __shared__ u64 out_base;
  const u32 xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const u32 p = (j >> ABITS) * 8 + xcd, sub = j & (A - 1);
  if (p >= (1u << PBITS)) return;
  for (int k = threadIdx.x; k < SLOTS; k += 1024) { t_lo[k] = 0; t_hi[k] = 0; t_sum[k] = 0.0; }
  if (threadIdx.x == 0) n_here = 0;
  __syncthreads();
  const u32 n_chunks = min(a.chunk_cursor[p], a.max_chunks);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (u32 k = wave; k < n_chunks; k += 16) {
    const u32 fill = a.chunk_fill[(u64)p * a.max_chunks + k];
    if ((u32)lane >= fill) continue;
    const u64x2* r = reinterpret_cast<const u64x2*>(a.recs) + (((u64)p * a.max_chunks + k) * CHUNK + lane) * 2;
    const u64x2 r0 = r[0];
    if (((r0.x >> (64 - PBITS - ABITS)) & (A - 1)) != sub) continue;
    const u64x2 r1 = r[1];
    u32 slot = (u32)(r0.x >> 16) & (SLOTS - 1);
    for (int tries = 0;; tries++) {
      if (tries >= SLOTS) { atomicAdd(leftover, 1u); break; }
      u64 cur = t_lo[slot];
      if (cur == 0) { cur = atomicCAS(&t_lo[slot], 0ull, r0.x); if (cur == 0) { t_aux[slot] = r1.y; __hip_atomic_store(&t_hi[slot], r0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicAdd(&n_here, 1u); cur = r0.x; } }
      if (cur == r0.x) {
        u64 h2;
        while ((h2 = __hip_atomic_load(&t_hi[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) {}
        if (h2 == r0.y) { atomicAdd(&t_sum[slot], __longlong_as_double((long long)r1.x)); break; }
      }
      slot = (slot + 1) & (SLOTS - 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) { out_base = atomicAdd(out_cursor, (u64)n_here); wcur = 0; }
  __syncthreads();
  for (int k0 = 0; k0 < SLOTS; k0 += 1024) {
    const int k = k0 + threadIdx.x;
    const bool occ = t_lo[k] != 0;
    const u64 bal = __ballot(occ);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&wcur, (u32)__popcll(bal));
    wbase = __shfl(wbase, 0, 64);
    if (occ) {
      const u32 at = wbase + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
      u64x2* o = reinterpret_cast<u64x2*>(out) + (out_base + at) * 2;
      o[0] = u64x2{t_lo[k], t_hi[k]};
      o[1] = u64x2{(u64)__double_as_longlong(t_sum[k]), t_aux[k]};
    }
  }
}

template <int PBITS, int BLK, int CHUNK, int ABITS>
void run_pipe(u64 n, u64 n_groups, int cus, int wg_per_cu, const u32x4* cols, u64 col_stride16) {
  constexpr int P = 1 << PBITS, TILE = BLK * 4, SLOTS = 4096;
  const int grid = cus * wg_per_cu;
  PipeArgs a;
  a.n = n; a.n_groups = n_groups;
  a.max_chunks = (u32)((n / P) * 5 / 4 / CHUNK + grid + 2);
  const u64 n_chunks_total = (u64)P * a.max_chunks;
  const u64 n_pad = (n + TILE - 1) / TILE * TILE;
  CHECK(hipMalloc(&a.recs, n_chunks_total * CHUNK * 32));
  CHECK(hipMalloc(&a.chunk_cursor, P * 4)); CHECK(hipMalloc(&a.chunk_fill, n_chunks_total * 4));
  CHECK(hipMalloc(&a.overflow, (n / 16 + 1024) * 32)); CHECK(hipMalloc(&a.overflow_cursor, 8));
  CHECK(hipMalloc(&a.t_lo, n_pad * 16)); CHECK(hipMalloc(&a.t_hi, n_pad * 16));
  const size_t lds = (size_t)TILE * 32 + (size_t)P * 24 + (size_t)TILE * 2;
  auto kern = p1_kernel<PBITS, BLK, CHUNK>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLK, lds));
  const float ms1 = timed([&] {
    CHECK(hipMemsetAsync(a.chunk_cursor, 0, P * 4)); CHECK(hipMemsetAsync(a.chunk_fill, 0, n_chunks_total * 4)); CHECK(hipMemsetAsync(a.overflow_cursor, 0, 8));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK), lds, 0, a, cols, col_stride16);
  });
  u64* d_out; CHECK(hipMalloc(&d_out, 16)); CHECK(hipMemset(d_out, 0, 16));
  hipLaunchKernelGGL((check_kernel<4>), dim3(cus * 4), dim3(256), 0, 0, a.recs, a.chunk_fill, n_chunks_total, CHUNK, d_out);
  u64 chk[2], ovf = 0; CHECK(hipMemcpy(chk, d_out, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&ovf, a.overflow_cursor, 8, hipMemcpyDeviceToHost));
  // 2A
  u64 *out, *out_cursor; u32* leftover;
  CHECK(hipMalloc(&out, (n_groups + 4096) * 32 * 2)); CHECK(hipMalloc(&out_cursor, 8)); CHECK(hipMalloc(&leftover, 4)); CHECK(hipMemset(leftover, 0, 4));
  auto k2 = p2a_kernel<PBITS, ABITS, SLOTS, CHUNK>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, SLOTS * 32));
  const int grid2 = P << ABITS;
  const float ms2 = timed([&] { CHECK(hipMemsetAsync(out_cursor, 0, 8)); hipLaunchKernelGGL(k2, dim3(grid2), dim3(1024), SLOTS * 32, 0, a, out, out_cursor, leftover); });
  u64 groups = 0; u32 lo = 0; CHECK(hipMemcpy(&groups, out_cursor, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&lo, leftover, 4, hipMemcpyDeviceToHost));
  std::vector<u64> h(groups * 4); CHECK(hipMemcpy(h.data(), out, groups * 32, hipMemcpyDeviceToHost));
  double total = 0; for (u64 g = 0; g < groups; g++) { double d; memcpy(&d, &h[g * 4 + 2], 8); total += d; }
  printf("T P %4d BLK %4d × %d/CU (occupancy %d) chunk %3d A %d: P1 %7.3f ms (rows in chunks %llu + overflow %llu %s), 2A %7.3f ms (%llu groups, sum %.0f %s, leftovers %u) → %.3f ms\n", P, BLK, wg_per_cu, per_cu,
         CHUNK, 1 << ABITS, ms1, chk[0], ovf, chk[0] + ovf == n ? "OK" : "WRONG", ms2, groups, total, (total == (double)(n - ovf) && lo == 0) ? "OK" : "CHECK", lo / 3, ms1 + ms2);
  CHECK(hipFree(a.recs)); CHECK(hipFree(a.chunk_cursor)); CHECK(hipFree(a.chunk_fill)); CHECK(hipFree(a.overflow)); CHECK(hipFree(a.overflow_cursor)); CHECK(hipFree(a.t_lo)); CHECK(hipFree(a.t_hi));
  CHECK(hipFree(d_out)); CHECK(hipFree(out)); CHECK(hipFree(out_cursor)); CHECK(hipFree(leftover));
}

// the column stream alone (what the scatter is added to)
__global__ __launch_bounds__(1024) void stream_kernel(u64 n, const u32x4* __restrict__ cols, u64 col_stride16, u64* sink) {
  const u64 n_tiles = (n + 4095) / 4096;
  u32 extra = 0;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const u64 row0 = t * 4096 + (u64)threadIdx.x * 4;
    for (int c0 = 0; c0 < 34; c0 += 8) {
      u32x4 v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = (c0 + c < 34 && row0 < n) ? __builtin_nontemporal_load(cols + (u64)(c0 + c) * col_stride16 + row0 / 4) : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 8; c++) extra += v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
    }
  }
  if (extra == 0x12345u) *sink = extra;
}

int main(int argc, char** argv) {
  const u64 n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull, n_groups = argc > 2 ? strtoull(argv[2], nullptr, 10) : 10000000ull;
  const char* only = argc > 3 ? argv[3] : "LS";
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, %llu rows, %llu groups\n", prop.name, cus, n, n_groups);
  u64* sink; CHECK(hipMalloc(&sink, 8));
  bool doL = false, doS = false;
  for (const char* c = only; *c; c++) { if (*c == 'L') doL = true; if (*c == 'S') doS = true; }
  if (doL) {
    const u64 ops = 64ull << 20;
    for (u64 span_mb : {1ull, 32ull, 1024ull, 4096ull}) {
      const u64 n_entries = span_mb * 1024 * 1024 / 32;
      u64* table; CHECK(hipMalloc(&table, n_entries * 32)); CHECK(hipMemset(table, 0, n_entries * 32));
      const int grid = cus * 8;
      run_rand<1>("16-byte load", table, n_entries, 0, ops, grid, sink);
      run_rand<2>("atomicAdd f64 (no return)", table, n_entries, 0, ops, grid, sink);
      run_rand<3>("load + atomicAdd", table, n_entries, 0, ops, grid, sink);
      run_rand<4>("plain 8-byte load + store", table, n_entries, 0, ops, grid, sink);
      run_rand<8>("32-byte store", table, n_entries, 0, ops, grid, sink);
      run_rand<16>("returning atomicAdd u64", table, n_entries, 0, ops, grid, sink);
      if (span_mb >= 1024) {
        for (u64 win_kb : {128ull, 1024ull, 8192ull}) {
          const u64 we = win_kb * 1024 / 32;
          // one workgroup per window at a time: grid = windows (capped), several passes
          const int g2 = (int)((n_entries / we) < (u64)(cus * 8) ? (n_entries / we) : (u64)(cus * 8));
          run_rand<1>("16-byte load, windowed", table, n_entries, we, ops, g2, sink);
          run_rand<2>("atomicAdd f64, windowed", table, n_entries, we, ops, g2, sink);
          run_rand<3>("load + atomicAdd, windowed", table, n_entries, we, ops, g2, sink);
          run_rand<4>("plain load + store, windowed", table, n_entries, we, ops, g2, sink);
          run_rand<5>("16-byte load + plain load + store, windowed", table, n_entries, we, ops, g2, sink);
        }
      }
      CHECK(hipFree(table));
    }
  }
  if (doS) {
    // 34 "columns" of n × 4 bytes (the 8-byte value column counts as two)
    const u64 col_stride16 = (n + 3) / 4 + 64;
    u32x4* cols; CHECK(hipMalloc(&cols, col_stride16 * 16 * 34)); CHECK(hipMemset(cols, 1, col_stride16 * 16 * 34));
    const float ms_stream = timed([&] { hipLaunchKernelGGL(stream_kernel, dim3(cus), dim3(1024), 0, 0, n, cols, col_stride16, sink); });
    printf("S column stream alone (34 × 4 B/row, 1 024-thread workgroups, 1/CU): %.3f ms = %.2f TB/s\n", ms_stream, n * 136.0 / ms_stream / 1e9);
    {
      constexpr int PB = 10;
      u64 *cursor, *out; const u64 cap = n / (1 << PB) * 5 / 4 + 4096;
      CHECK(hipMalloc(&cursor, (1 << PB) * 8)); CHECK(hipMalloc(&out, cap * (1 << PB) * 32));
      const float ms = timed([&] { CHECK(hipMemsetAsync(cursor, 0, (1 << PB) * 8)); hipLaunchKernelGGL((naive_scatter_kernel<PB, 4>), dim3(cus * 4), dim3(256), 0, 0, n, n_groups, cursor, out, cap); });
      printf("S naive scatter (one 32-byte store per row), P 1024: %.3f ms\n", ms);
      CHECK(hipFree(cursor)); CHECK(hipFree(out));
    }
    run_scatter<10, 1024, 4, 64, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<10, 512, 4, 64, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<9, 1024, 4, 64, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<8, 1024, 4, 128, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<8, 512, 4, 128, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<8, 256, 4, 128, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<6, 256, 4, 256, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<10, 512, 6, 64, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<8, 512, 6, 128, false>(n, n_groups, cus, cols, col_stride16);
    run_scatter<10, 1024, 4, 64, true>(n, n_groups, cus, cols, col_stride16);
    run_scatter<10, 512, 4, 64, true>(n, n_groups, cus, cols, col_stride16);
    run_scatter<8, 512, 4, 128, true>(n, n_groups, cus, cols, col_stride16);
    run_scatter<8, 256, 4, 128, true>(n, n_groups, cus, cols, col_stride16);
    run_scatter<6, 256, 4, 256, true>(n, n_groups, cus, cols, col_stride16);
    run_direct<12, 1024, 4, 64, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<12, 1024, 4, 32, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<12, 1024, 4, 16, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<12, 512, 4, 32, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<11, 1024, 4, 32, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<10, 1024, 4, 32, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<10, 512, 4, 32, false>(n, n_groups, cus, 2, cols, col_stride16);
    run_direct<13, 1024, 4, 32, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<12, 1024, 6, 32, false>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<12, 1024, 4, 32, true>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<12, 512, 4, 32, true>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<11, 1024, 4, 32, true>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<10, 1024, 4, 32, true>(n, n_groups, cus, 1, cols, col_stride16);
    run_direct<10, 512, 4, 32, true>(n, n_groups, cus, 2, cols, col_stride16);
    CHECK(hipFree(cols));
  }
  bool doT = false;
  for (const char* c = only; *c; c++) if (*c == 'T') doT = true;
  if (doT) {
    const u64 col_stride16 = (n + 3) / 4 + 64;
    u32x4* cols; CHECK(hipMalloc(&cols, col_stride16 * 16 * 34)); CHECK(hipMemset(cols, 1, col_stride16 * 16 * 34));
    run_pipe<10, 512, 64, 2>(n, n_groups, cus, 1, cols, col_stride16);
    run_pipe<10, 512, 64, 2>(n, n_groups, cus, 2, cols, col_stride16);
    run_pipe<9, 512, 64, 3>(n, n_groups, cus, 2, cols, col_stride16);
    run_pipe<8, 512, 128, 4>(n, n_groups, cus, 2, cols, col_stride16);
    run_pipe<10, 1024, 64, 2>(n, n_groups, cus, 1, cols, col_stride16);
    run_pipe<9, 1024, 64, 3>(n, n_groups, cus, 1, cols, col_stride16);
    run_pipe<10, 256, 64, 2>(n, n_groups, cus, 4, cols, col_stride16);
    run_pipe<9, 256, 64, 3>(n, n_groups, cus, 4, cols, col_stride16);
    run_pipe<11, 512, 32, 1>(n, n_groups, cus, 2, cols, col_stride16);
    CHECK(hipFree(cols));
  }
  return 0;
}
