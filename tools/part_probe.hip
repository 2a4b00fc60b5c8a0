// Measurement aid: is PARTITION-BY-FINGERPRINT worth building for the high-cardinality scan (cfg 5)?
//
// fdb_hash_kernel touches the global table once per ROW (a 16-byte load of the home entry + one atomic per aggregate): 100 M rows
// over 10 M groups are 200 M scattered single-sector operations, ≈4.2 ms of the 6.9 ms steady-state scan. The alternative prices
// here, on the part that differs (the 138.5 B/row column stream and the fingerprint arithmetic are the same either way):
//   A  direct: every row probes a global open-addressing table of 32-byte entries [fp lo | fp hi | count | sum] (CAS to claim,
//      atomicAdd to update) — what the product does today;
//   B  partition, then aggregate per partition in LDS:
//      B1 histogram of the partition ids (top bits of the fingerprint), B2 exclusive scan, B3 scatter of 32-byte records
//      (fp lo, fp hi, value, row) into their partitions (per-tile LDS histogram → one global reservation per tile and partition →
//      per-row store), B4 one workgroup per partition: LDS hash table keyed by the 128-bit fingerprint, then its groups appended to
//      a dense output (one global write per GROUP instead of two scattered operations per ROW).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/part_probe.hip -o tools/part_probe && tools/part_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }
// row i belongs to group g(i) (uniform over n_groups); its 128-bit fingerprint is a function of the group only
__device__ __forceinline__ void row_fp(u64 i, u64 n_groups, u64* lo, u64* hi) {
  const u64 g = mix(i * 0x9E3779B97F4A7C15ULL + 1) % n_groups;
  *lo = mix(g + 0x1234567ULL) | 1ull;
  *hi = mix(g * 0xD6E8FEB86659FD93ULL + 7);
}

// ---- A: direct probing ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void direct_kernel(u64 n, u64 n_groups, u64* table, u64 mask, u64* n_inserted) {
  u64 ins = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    u64 lo, hi;
    row_fp(i, n_groups, &lo, &hi);
    u64 slot = lo & mask;
    for (;;) {
      u64* e = table + slot * 4;
      u64 cur = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0) { cur = atomicCAS(e, 0ull, lo); if (cur == 0) { __hip_atomic_store(e + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ins++; cur = lo; } }
      if (cur == lo) {
        u64 h2;
        while ((h2 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {}
        if (h2 == hi) { atomicAdd(reinterpret_cast<double*>(e + 3), 1.0); break; }
      }
      slot = (slot + 1) & mask;
    }
  }
  if (ins) atomicAdd(n_inserted, ins);
}

// ---- B1: histogram of partition ids --------------------------------------------------------------------------------------------
template <int PBITS>
__global__ __launch_bounds__(256) void hist_kernel(u64 n, u64 n_groups, unsigned int* hist) {
  __shared__ unsigned int h[1 << PBITS];
  for (int k = threadIdx.x; k < (1 << PBITS); k += 256) h[k] = 0;
  __syncthreads();
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    u64 lo, hi;
    row_fp(i, n_groups, &lo, &hi);
    atomicAdd(&h[lo >> (64 - PBITS)], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < (1 << PBITS); k += 256) if (h[k]) atomicAdd(&hist[k], h[k]);
}

// ---- B3: scatter into partitions: per tile of TILE rows an LDS histogram, one global reservation per (tile, partition) ------------
struct Rec { u64 lo, hi; double v; u64 row; };
template <int PBITS, int TILE>
__global__ __launch_bounds__(256) void scatter_kernel(u64 n, u64 n_groups, u64* cursor, Rec* out) {
  __shared__ unsigned int cnt[1 << PBITS];
  __shared__ u64 base[1 << PBITS];
  const u64 n_tiles = (n + TILE - 1) / TILE;
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    for (int k = threadIdx.x; k < (1 << PBITS); k += 256) cnt[k] = 0;
    __syncthreads();
    u64 lo[TILE / 256], hi[TILE / 256];
    unsigned int off[TILE / 256];
#pragma unroll
    for (int u = 0; u < TILE / 256; u++) {
      const u64 i = t * TILE + (u64)u * 256 + threadIdx.x;
      off[u] = 0xFFFFFFFFu;
      if (i < n) { row_fp(i, n_groups, &lo[u], &hi[u]); off[u] = atomicAdd(&cnt[lo[u] >> (64 - PBITS)], 1u); }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < (1 << PBITS); k += 256) if (cnt[k]) base[k] = atomicAdd(&cursor[k], (u64)cnt[k]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TILE / 256; u++) {
      if (off[u] == 0xFFFFFFFFu) continue;
      const u64 i = t * TILE + (u64)u * 256 + threadIdx.x;
      out[base[lo[u] >> (64 - PBITS)] + off[u]] = Rec{lo[u], hi[u], 1.0, i};
    }
    __syncthreads();
  }
}

// ---- B4: one workgroup per partition: aggregate in an LDS table, append the groups to the output --------------------------------
template <int SLOTS>
__global__ __launch_bounds__(1024) void agg_kernel(const Rec* recs, const u64* part_begin, int n_parts, u64* out_lo, u64* out_hi, double* out_sum, u64* out_cursor,
                                                   unsigned int* overflow) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* t_lo = reinterpret_cast<u64*>(smem);
  u64* t_hi = t_lo + SLOTS;
  double* t_sum = reinterpret_cast<double*>(t_hi + SLOTS);
  __shared__ unsigned int n_groups_here;
  __shared__ u64 out_base;
  for (int p = blockIdx.x; p < n_parts; p += gridDim.x) {
    for (int k = threadIdx.x; k < SLOTS; k += 1024) { t_lo[k] = 0; t_hi[k] = 0; t_sum[k] = 0.0; }
    if (threadIdx.x == 0) n_groups_here = 0;
    __syncthreads();
    const u64 b = part_begin[p], e = part_begin[p + 1];
    for (u64 i = b + threadIdx.x; i < e; i += 1024) {
      const Rec r = recs[i];
      unsigned int slot = (unsigned int)(r.lo >> 20) & (SLOTS - 1);
      for (int tries = 0;; tries++) {
        if (tries >= SLOTS) { atomicAdd(overflow, 1u); break; }
        u64 cur = t_lo[slot];
        if (cur == 0) { cur = atomicCAS(&t_lo[slot], 0ull, r.lo); if (cur == 0) { t_hi[slot] = r.hi; atomicAdd(&n_groups_here, 1u); cur = r.lo; } }
        if (cur == r.lo) {
          u64 h2;
          while ((h2 = __hip_atomic_load(&t_hi[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) {}
          if (h2 == r.hi) { atomicAdd(&t_sum[slot], r.v); break; }
        }
        slot = (slot + 1) & (SLOTS - 1);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) out_base = atomicAdd(out_cursor, (u64)n_groups_here);
    __syncthreads();
    // compact the occupied slots (order inside a partition does not matter): wave ballots
    __shared__ unsigned int wcur;
    if (threadIdx.x == 0) wcur = 0;
    __syncthreads();
    for (int k0 = 0; k0 < SLOTS; k0 += 1024) {
      const int k = k0 + threadIdx.x;
      const bool occ = t_lo[k] != 0;
      const u64 bal = __ballot(occ);
      unsigned int wbase = 0;
      if ((threadIdx.x & 63) == 0 && bal) wbase = atomicAdd(&wcur, (unsigned int)__popcll(bal));
      wbase = __shfl(wbase, 0, 64);
      if (occ) {
        const unsigned int at = wbase + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0u));
        out_lo[out_base + at] = t_lo[k]; out_hi[out_base + at] = t_hi[k]; out_sum[out_base + at] = t_sum[k];
      }
    }
    __syncthreads();
  }
}

template <typename F>
float timed(F f, int reps = 3) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const u64 n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull, n_groups = argc > 2 ? strtoull(argv[2], nullptr, 10) : 10000000ull;
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %llu rows, %llu groups\n", prop.name, n, n_groups);
  // ---- A
  u64 cap = 1; while (cap < 2 * n_groups) cap <<= 1; cap <<= 1;  // load ≤ 0.3 like the product's table after its growth
  u64 *table, *d_ins;
  CHECK(hipMalloc(&table, cap * 32)); CHECK(hipMalloc(&d_ins, 8));
  CHECK(hipMemset(table, 0, cap * 32)); CHECK(hipMemset(d_ins, 0, 8));
  const float a_first = timed([&] { hipLaunchKernelGGL(direct_kernel, dim3(cus * 8), dim3(256), 0, 0, n, n_groups, table, cap - 1, d_ins); }, 1);
  const float a_steady = timed([&] { hipLaunchKernelGGL(direct_kernel, dim3(cus * 8), dim3(256), 0, 0, n, n_groups, table, cap - 1, d_ins); });
  u64 ins = 0; CHECK(hipMemcpy(&ins, d_ins, 8, hipMemcpyDeviceToHost));
  printf("A direct probing into a %.2f GB table: first pass (creates %llu groups) %.3f ms, steady state %.3f ms = %.1f ps/row\n", cap * 32 / 1e9, ins, a_first, a_steady, a_steady * 1e9 / n);
  CHECK(hipFree(table));
  // ---- B
  constexpr int PBITS = 12, P = 1 << PBITS, SLOTS = 4096;  // 4 096 partitions of ≈24 k rows / ≈2.4 k groups; LDS table 4 096 × 24 B = 96 KiB
  unsigned int *hist, *d_over; u64 *cursor, *part_begin, *out_lo, *out_hi, *out_cursor; double* out_sum; Rec* recs;
  CHECK(hipMalloc(&hist, P * 4)); CHECK(hipMalloc(&cursor, P * 8)); CHECK(hipMalloc(&part_begin, (P + 1) * 8)); CHECK(hipMalloc(&recs, n * sizeof(Rec)));
  CHECK(hipMalloc(&out_lo, n_groups * 8 + 4096)); CHECK(hipMalloc(&out_hi, n_groups * 8 + 4096)); CHECK(hipMalloc(&out_sum, n_groups * 8 + 4096)); CHECK(hipMalloc(&out_cursor, 8)); CHECK(hipMalloc(&d_over, 4));
  CHECK(hipMemset(d_over, 0, 4));
  const float b1 = timed([&] { CHECK(hipMemsetAsync(hist, 0, P * 4)); hipLaunchKernelGGL((hist_kernel<PBITS>), dim3(cus * 8), dim3(256), 0, 0, n, n_groups, hist); });
  std::vector<unsigned int> h(P); CHECK(hipMemcpy(h.data(), hist, P * 4, hipMemcpyDeviceToHost));
  std::vector<u64> pb(P + 1, 0); for (int k = 0; k < P; k++) pb[k + 1] = pb[k] + h[k];
  CHECK(hipMemcpy(part_begin, pb.data(), (P + 1) * 8, hipMemcpyHostToDevice));
  if (pb[P] != n) { printf("histogram lost rows\n"); return 1; }
  float b3[2];
  b3[0] = timed([&] { CHECK(hipMemcpyAsync(cursor, part_begin, P * 8, hipMemcpyDeviceToDevice)); hipLaunchKernelGGL((scatter_kernel<PBITS, 2048>), dim3(cus * 4), dim3(256), 0, 0, n, n_groups, cursor, recs); });
  b3[1] = timed([&] { CHECK(hipMemcpyAsync(cursor, part_begin, P * 8, hipMemcpyDeviceToDevice)); hipLaunchKernelGGL((scatter_kernel<PBITS, 4096>), dim3(cus * 4), dim3(256), 0, 0, n, n_groups, cursor, recs); });
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(agg_kernel<SLOTS>), hipFuncAttributeMaxDynamicSharedMemorySize, SLOTS * 24));
  const float b4 = timed([&] { CHECK(hipMemsetAsync(out_cursor, 0, 8)); hipLaunchKernelGGL((agg_kernel<SLOTS>), dim3(cus), dim3(1024), SLOTS * 24, 0, recs, part_begin, P, out_lo, out_hi, out_sum, out_cursor, d_over); });
  u64 groups = 0; unsigned int over = 0; CHECK(hipMemcpy(&groups, out_cursor, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&over, d_over, 4, hipMemcpyDeviceToHost));
  std::vector<double> sums(groups); CHECK(hipMemcpy(sums.data(), out_sum, groups * 8, hipMemcpyDeviceToHost));
  double total = 0; for (double s : sums) total += s;
  printf("B1 histogram (%d partitions) %.3f ms | B3 scatter of 32-byte records: %.3f ms (2 048-row tiles), %.3f ms (4 096-row tiles) | B4 LDS aggregate per partition %.3f ms\n", P, b1, b3[0], b3[1], b4);
  const float b_best = b1 + (b3[0] < b3[1] ? b3[0] : b3[1]) + b4;
  printf("B total %.3f ms = %.1f ps/row (%llu groups out, sum of sums %.0f = rows: %s, LDS table overflows: %u)\n", b_best, b_best * 1e9 / n, groups, total, total == (double)n ? "yes" : "NO", over);
  printf("=> partitioning %s: direct steady state %.3f ms vs partition + aggregate %.3f ms (the fingerprints are computed %s in B: once more in the histogram pass)\n",
         b_best < a_steady ? "WINS" : "LOSES", a_steady, b_best, "twice");
  return 0;
}
