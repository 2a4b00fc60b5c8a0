// fdb_merge.hip — gfx950 kernels that move GROUPS (key tuple + count + accumulators) between hash tables: the merge of one table into
// another (≙ Synchronizer + HashAggregate(final=true) for two chains of one GPU, synchronize.go:31-53, aggregate.go:340-348, :965-969),
// the import of packed rows (the receiving side of the cross-GPU exchange, the run store's fallback, the dense → hash migration) and the
// hash-partitioned export (the sending side).
//
// What these kernels move is 150-300 bytes per group, once — HBM-bound copy work with a probe in the middle. Round 5's versions were
// thread-per-tuple loops: a lane walked "its" 144-byte tuple with 34 four-byte loads 144 bytes apart from its neighbours' (every load
// its own sector), re-walked it to store an insert, and a 10 M-group merge took 13.5 ms (0.13 TB/s). Here a WAVE owns 64 tuples:
//   A  the tuples are fetched cooperatively — lane ↦ 16-byte quad of the tuples laid end to end, so a tuple is read by 9 adjacent
//      lanes as whole sectors — into an LDS tile [word][tuple];
//   B  lane ↦ tuple: every column is read from the tile (conflict-free), translated (id LUT; none when the two plans' dictionaries
//      agree), folded into the 128-bit fingerprint and written to a second tile in the DESTINATION layout;
//   C  probe / claim (hash_find_or_insert: all 64 probes of the wave in flight together);
//   D  the tuples of the lanes that inserted leave as 16-byte stores, again quad by quad across the lanes;
//   E  count and accumulators: plain stores into a slot this lane just created when the source holds every group once (a table),
//      atomics otherwise.
// A table source is scanned in place: occupied slots are queued in LDS until 64 are pending, so every step runs with full lanes
// whatever the load factor (cfg 5: 15 %). The export runs the same A / B, then ranks the tuples of each partition inside the wave;
// per-wave counts from a first pass are prefix-summed on the device, so the scatter pass needs no atomics and no host round trip.
#include <hip/hip_runtime.h>

#define FDB_DEVICE_HELPERS 1
#include <algorithm>
#include <cstring>

#include "fdb_kernels.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define FDB_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const FDB_GLOBAL T* as_global(const T* p) { return (const FDB_GLOBAL T*)p; }

// Lanes of one wave hand data to each other through LDS: the hardware executes a wave's LDS operations in order, the fence keeps the
// compiler from moving them across the hand-over.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform column descriptors through the constant address space: scalar loads
typedef const __attribute__((address_space(4))) FdbHashCol* ConstCols;

struct WaveTiles {
  uint32_t* in;                  // [in_words][64]
  uint32_t* out;                 // [out_words][64]
  unsigned long long* queue;     // [128] pending source slots (table source)
  unsigned long long* ins_slot;  // [64]
  uint32_t* ins_lane;            // [64]
};
// (`alias`: the incoming and the outgoing tuple have the same layout — the usual case, two plans of one query — and are translated in place)
__host__ __device__ inline size_t wave_lds_bytes(int in_words, int out_words, bool alias) {
  return (size_t)4 * 64 * (alias ? (in_words > out_words ? in_words : out_words) : in_words + out_words) + 128 * 8 + 64 * 8 + 64 * 4;
}
__device__ __forceinline__ WaveTiles wave_tiles(unsigned char* smem, int wv, int in_words, int out_words, bool alias, size_t extra) {
  unsigned char* base = smem + (size_t)wv * (wave_lds_bytes(in_words, out_words, alias) + extra);
  WaveTiles t;
  t.queue = reinterpret_cast<unsigned long long*>(base);
  t.ins_slot = t.queue + 128;
  t.ins_lane = reinterpret_cast<uint32_t*>(t.ins_slot + 64);
  t.in = t.ins_lane + 64;
  t.out = alias ? t.in : t.in + (size_t)64 * in_words;
  return t;
}

// A: `n_act` tuples (tuple t at base + off[t] words, `words` of each wanted: a multiple of 4 when `quads`) → tile[word][t]
template <typename TuplePtr>
__device__ __forceinline__ void gather_tuples(uint32_t* tile, int n_act, int words, bool quads, int lane, TuplePtr tuple_ptr) {
  if (quads) {
    const int Q = words >> 2, total = n_act * Q;
    for (int q0 = 0; q0 < total; q0 += 256) {  // four loads in flight per lane
      u32x4 v[4];
      int t[4], j[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = q0 + u * 64 + lane;
        t[u] = q < total ? q / Q : -1;
        j[u] = q - t[u] * Q;
        if (t[u] >= 0) v[u] = *reinterpret_cast<const FDB_GLOBAL u32x4*>(as_global(tuple_ptr(t[u]) + j[u] * 4));
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (t[u] < 0) continue;
        uint32_t* d = tile + (size_t)(j[u] * 4) * 64 + t[u];
        d[0] = v[u].x; d[64] = v[u].y; d[128] = v[u].z; d[192] = v[u].w;
      }
    }
  } else {
    const int total = n_act * words;
    for (int q = lane; q < total; q += 64) {
      const int t = q / words, w = q - t * words;
      tile[(size_t)w * 64 + t] = as_global(tuple_ptr(t))[w];
    }
  }
  wave_sync();
}

// B: lane ↦ tuple. Translates the tuple of tile `in` into the destination layout in tile `out` and returns its fingerprint.
__device__ __forceinline__ void translate_tuple(const uint32_t* in, int in_words, uint32_t* out, int out_words, const FdbHashCol* cols, int n_cols, bool dst_cols, int lane,
                                                bool active, unsigned long long& h1, unsigned long long& h2) {
  ConstCols q = (ConstCols)cols;
  if (out != in) for (int w = 0; w < out_words; w++) out[(size_t)w * 64 + lane] = 0u;
  else for (int w = in_words; w < out_words; w++) out[(size_t)w * 64 + lane] = 0u;  // (in place: what the incoming tuple does not reach is padding — zeros, not what the tile held)
  const unsigned long long in_mask = (unsigned long long)in[lane] | ((unsigned long long)in[64 + lane] << 32);
  unsigned long long vm = 0;
  h1 = 0; h2 = 0;
  for (int c = 0; c < n_cols; c++) {
    // merge (dst_cols): cols[c] is DESTINATION column c — src_word in the incoming tuple (-1: absent), word in ours;
    // export (!dst_cols): cols[c] is SOURCE column c — word in the source tuple, src_word in the destination layout.
    const int kind = q[c].kind, iw = dst_cols ? q[c].src_word : q[c].word, ow = dst_cols ? q[c].word : q[c].src_word, gi = q[c].gi;
    if (iw < 0) continue;
    const unsigned long long k1 = q[c].k1, k2 = q[c].k2;
    if (kind == 0) {
      const uint32_t* lut = q[c].lut;
      uint32_t id = active ? in[(size_t)iw * 64 + lane] : 0u;  // (a lane without a tuple holds whatever the tile held: it must not index the LUT)
      if (id != 0u && lut != nullptr) id = as_global(lut)[id];
      if (id != 0u) { fp_add32(h1, h2, k1, k2, id); vm |= 1ull << gi; }
      out[(size_t)ow * 64 + lane] = id;
    } else {
      const uint32_t lo = in[(size_t)iw * 64 + lane], hi = in[(size_t)(iw + 1) * 64 + lane];
      if ((in_mask >> q[c].lut_len) & 1ull) {  // lut_len: the column's bit in the INCOMING valid mask
        const unsigned long long v = (unsigned long long)lo | ((unsigned long long)hi << 32);
        if (v != 0ull) fp_add(h1, h2, k1, k2, v);
        vm |= 1ull << gi;
        out[(size_t)ow * 64 + lane] = lo; out[(size_t)(ow + 1) * 64 + lane] = hi;
      } else if (out == in) { out[(size_t)ow * 64 + lane] = 0u; out[(size_t)(ow + 1) * 64 + lane] = 0u; }
    }
  }
  out[lane] = (uint32_t)vm; out[64 + lane] = (uint32_t)(vm >> 32);
  fp_final(h1, h2);
}

// A wave's walk over its contiguous range of a table's slots: occupied slots wait in the LDS queue until 64 are pending, then
// `process(64)` handles queue[0 .. 64) with full lanes; the rest at the end. Four batches of occupancy words in flight per step.
template <typename Process>
__device__ __forceinline__ void for_occupied_slots(const unsigned long long* table, uint64_t capacity, int ew, const WaveTiles& T, int lane, int64_t gw, int64_t n_gw,
                                                   Process process) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int64_t n_batches = (int64_t)((capacity + 63) / 64), per = (n_batches + n_gw - 1) / n_gw;
  const int64_t b0 = gw * per, b1 = b0 + per < n_batches ? b0 + per : n_batches;
  int pending = 0;
  for (int64_t b = b0; b < b1; b += 4) {
    unsigned long long fp[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t s = (uint64_t)(b + u) * 64 + lane;
      fp[u] = (b + u < b1 && s < capacity) ? __builtin_nontemporal_load(as_global(table + s * (uint64_t)ew)) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool occ = fp[u] != 0ull;
      const unsigned long long ob = __ballot(occ);
      if (ob == 0ull) continue;
      if (occ) T.queue[pending + __popcll(ob & lt)] = (uint64_t)(b + u) * 64 + lane;
      pending += __popcll(ob);
      wave_sync();
      if (pending >= 64) {
        process(64);
        const unsigned long long rest = lane < pending - 64 ? T.queue[64 + lane] : 0ull;
        wave_sync();
        if (lane < pending - 64) T.queue[lane] = rest;
        pending -= 64;
        wave_sync();
      }
    }
  }
  if (pending > 0) process(pending);
}

// ---- merge ------------------------------------------------------------------------------------------------------------------------
template <bool TABLE_SRC>
__global__ __launch_bounds__(256) void hash_merge_wave_kernel(const FdbHashMergeArgs m, const int in_words, const bool quads, const bool alias) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ unsigned int s_new;
  if (threadIdx.x == 0) s_new = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
  const int kw = m.key_words, ew = m.entry_words;
  const WaveTiles T = wave_tiles(smem, wv, in_words, kw, alias, 0);
  const unsigned long long lt = (1ull << lane) - 1ull;

  // one step over ≤ 64 incoming groups, lane t holds group t: queue[t] is its slot (table source), row0 + t its row otherwise
  auto process = [&](int n_act, int64_t row0) {
    if (TABLE_SRC) gather_tuples(T.in, n_act, in_words, quads, lane, [&](int t) { return m.src_keys + T.queue[t] * (uint64_t)m.src_key_words; });
    else gather_tuples(T.in, n_act, in_words, quads, lane, [&](int t) { return m.in_keys + (row0 + t) * (int64_t)m.in_key_words; });
    unsigned long long h1, h2;
    const bool active = lane < n_act;
    translate_tuple(T.in, in_words, T.out, kw, m.cols, m.n_cols, true, lane, active, h1, h2);
    bool inserted = false;
    uint64_t slot = 0;
    if (active) slot = hash_find_or_insert(m.table, m.mask, ew, h1, h2, inserted);
    // D: the new groups' tuples
    const unsigned long long ib = __ballot(active && inserted);
    const int n_ins = __popcll(ib);
    if (active && inserted) { const int r = __popcll(ib & lt); T.ins_lane[r] = (uint32_t)lane; T.ins_slot[r] = slot; }
    wave_sync();
    const int QO = kw >> 2, total = n_ins * QO;
    for (int q = lane; q < total; q += 64) {
      const int r = q / QO, j = q - r * QO;
      const uint32_t t = T.ins_lane[r];
      const uint32_t* s = T.out + (size_t)(j * 4) * 64 + t;
      *reinterpret_cast<u32x4*>(m.keys + T.ins_slot[r] * (uint64_t)kw + j * 4) = u32x4{s[0], s[64], s[128], s[192]};
    }
    if (lane == 0 && n_ins != 0) atomicAdd(&s_new, (unsigned int)n_ins);
    // E: count and accumulators
    if (active) {
      const unsigned long long* e = TABLE_SRC ? m.src_table + T.queue[lane] * (uint64_t)m.src_entry_words + 2 : m.entries + (row0 + lane) * (int64_t)m.in_entry_words;
      unsigned long long* d = m.table + slot * (uint64_t)ew;
      if (inserted && m.unique_source) {  // nobody else merges into a slot this lane just created: plain stores over the identities
        d[2] = as_global(e)[0];
        for (int j = 0; j < m.n_aggs; j++) if (m.funcs[j] != 0) d[3 + j] = as_global(e)[1 + j];
      } else {
        atomicAdd(d + 2, as_global(e)[0]);
        for (int j = 0; j < m.n_aggs; j++) {
          const int f = m.funcs[j];
          const unsigned long long v = as_global(e)[1 + j];
          if (f == 1) atomicAdd(d + 3 + j, v);
          else if (f == 2) atomicAdd(reinterpret_cast<double*>(d + 3 + j), __longlong_as_double((long long)v));
          else if (f == 3) atomicMin(reinterpret_cast<long long*>(d + 3 + j), (long long)v);
          else if (f == 4) atomicMax(reinterpret_cast<long long*>(d + 3 + j), (long long)v);
        }
      }
    }
    wave_sync();
  };

  const int64_t gw = (int64_t)blockIdx.x * n_wv + wv, n_gw = (int64_t)gridDim.x * n_wv;
  if (TABLE_SRC) {
    for_occupied_slots(m.src_table, m.src_capacity, m.src_entry_words, T, lane, gw, n_gw, [&](int n_act) { process(n_act, 0); });
  } else {
    const int64_t n_batches = (m.n + 63) / 64;
    for (int64_t b = gw; b < n_batches; b += n_gw) {
      const int64_t row0 = b * 64;
      process((int)(m.n - row0 < 64 ? m.n - row0 : 64), row0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_new != 0) atomicAdd(m.n_groups, (unsigned long long)s_new);
}

// ---- hash-partitioned export ----------------------------------------------------------------------------------------------------------
// PASS 0 counts the groups of every (wave, partition); PASS 1 writes the rows: wave g puts its k-th group of partition p at
// wave_base[g][p] + k. Both passes walk the table the same way (a contiguous range of slots per wave, same grid).
#define FDB_PART_EXTRA (FDB_MAX_PARTS * 8)  // a wave's running row positions, one per partition
template <int PASS>
__global__ __launch_bounds__(256) void hash_partition_wave_kernel(const FdbHashPartArgs p, const int in_words, const bool quads, const bool alias) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
  const int row_words = p.row_words32, dkw = p.dst_key_words;
  const WaveTiles T = wave_tiles(smem, wv, in_words, row_words, alias, FDB_PART_EXTRA);
  // [n_parts] next row of this wave in every partition: counts from 0 (PASS 0), positions from the wave's bases (PASS 1)
  unsigned long long* my_pos = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(T.queue) + wave_lds_bytes(in_words, row_words, alias));
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int64_t gw = (int64_t)blockIdx.x * n_wv + wv, n_gw = (int64_t)gridDim.x * n_wv;
  if (lane < p.n_parts) my_pos[lane] = PASS == 1 ? p.wave_bases[gw * p.n_parts + lane] : 0ull;
  wave_sync();

  auto process = [&](int n_act) {
    const bool active = lane < n_act;
    unsigned long long h1 = 0, h2 = 0;
    if (PASS == 0 && p.same_ids) {
      // the destination's key ids are ours: the fingerprint in the entry is the destination's — the counting pass needs no tuple
      if (active) h2 = __builtin_nontemporal_load(as_global(p.table + T.queue[lane] * (uint64_t)p.entry_words + 1));
    } else {
      gather_tuples(T.in, n_act, in_words, quads, lane, [&](int t) { return p.keys + T.queue[t] * (uint64_t)p.key_words; });
      // (PASS 0 needs the fingerprint only; the ids translate_tuple writes into the tile on the way are not read)
      translate_tuple(T.in, in_words, T.out, PASS == 1 ? row_words : 2, p.cols, p.n_cols, false, lane, active, h1, h2);
      if (p.same_ids && active) h2 = as_global(p.table + T.queue[lane] * (uint64_t)p.entry_words)[1];  // (what the counting pass saw)
    }
    const uint32_t part = active ? (uint32_t)((h2 >> 32) % (unsigned long long)p.n_parts) : 0xFFFFFFFFu;
    unsigned long long dst_row = 0;
    for (int q = 0; q < p.n_parts; q++) {  // rank inside the wave, partition by partition (wave-uniform loop)
      const unsigned long long b = __ballot(part == (uint32_t)q);
      if (b == 0ull) continue;
      const unsigned long long before = my_pos[q];
      if (part == (uint32_t)q) dst_row = before + (unsigned long long)__popcll(b & lt);
      wave_sync();
      if (lane == 0) my_pos[q] = before + (unsigned long long)__popcll(b);
      wave_sync();
    }
    if (PASS == 1) {
      if (active) {
        const unsigned long long* e = p.table + T.queue[lane] * (uint64_t)p.entry_words + 2;
        for (int v = 0; v < p.n_vals; v++) {
          const unsigned long long x = as_global(e)[v];
          T.out[(size_t)(dkw + 2 * v) * 64 + lane] = (uint32_t)x; T.out[(size_t)(dkw + 2 * v + 1) * 64 + lane] = (uint32_t)(x >> 32);
        }
        for (int w = dkw + 2 * p.n_vals; w < row_words; w++) T.out[(size_t)w * 64 + lane] = 0u;
        T.ins_slot[lane] = dst_row;
      }
      wave_sync();
      const int RQ = row_words >> 2, total = n_act * RQ;
      for (int q = lane; q < total; q += 64) {
        const int t = q / RQ, j = q - t * RQ;
        const uint32_t* s = T.out + (size_t)(j * 4) * 64 + t;
        *reinterpret_cast<u32x4*>(p.out + T.ins_slot[t] * (uint64_t)row_words + j * 4) = u32x4{s[0], s[64], s[128], s[192]};
      }
    }
    wave_sync();
  };
  for_occupied_slots(p.table, p.capacity, p.entry_words, T, lane, gw, n_gw, process);
  if (PASS == 0 && lane < p.n_parts) p.wave_counts[gw * p.n_parts + lane] = (uint32_t)my_pos[lane];
}

// counts[n_waves][n_parts] → bases[n_waves][n_parts] (partition regions back to back, waves in order inside a region) and the totals.
// One workgroup of 16 waves; a wave owns partitions wave, wave + 16, …: it sums a partition's counts 64 waves at a time (phase 1), the
// totals meet in LDS, and it then walks the counts again with a wave-wide scan (phase 2). (First version: one thread per partition
// walking its column alone — 0.48 ms for 2 304 waves × 2 partitions, a chain of dependent loads.)
__global__ __launch_bounds__(1024) void partition_bases_kernel(const uint32_t* __restrict__ counts, unsigned long long* __restrict__ bases, unsigned long long* __restrict__ totals,
                                                               int64_t n_waves, int n_parts) {
  __shared__ unsigned long long s_tot[FDB_MAX_PARTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = wave; q < n_parts; q += 16) {
    unsigned long long tot = 0;
    for (int64_t g0 = 0; g0 < n_waves; g0 += 256) {
      uint32_t c[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int64_t g = g0 + u * 64 + lane; c[u] = g < n_waves ? counts[g * n_parts + q] : 0u; }
      tot += (unsigned long long)c[0] + c[1] + c[2] + c[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += (unsigned long long)__shfl_xor((long long)tot, o, 64);
    if (lane == 0) { s_tot[q] = tot; totals[q] = tot; }
  }
  __syncthreads();
  for (int q = wave; q < n_parts; q += 16) {
    unsigned long long run = 0;
    for (int i = 0; i < q; i++) run += s_tot[i];
    for (int64_t g0 = 0; g0 < n_waves; g0 += 64) {
      const int64_t g = g0 + lane;
      const unsigned long long c = g < n_waves ? counts[g * n_parts + q] : 0ull;
      unsigned long long incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = (unsigned long long)__shfl_up((long long)incl, o, 64); if (lane >= o) incl += t; }
      if (g < n_waves) bases[g * n_parts + q] = run + incl - c;
      run += (unsigned long long)__shfl((long long)incl, 63, 64);
    }
  }
}

struct Geometry { int waves; int blocks; size_t lds; };
// Four waves per workgroup when their tiles fit 40 KiB (fewer otherwise), as many workgroups per CU as the 160 KiB of LDS hold (≤ 4):
// the probes are latency-bound, so occupancy is what hides them. The grid is persistent (a wave walks a contiguous range).
Geometry geometry(int device, size_t per_wave, int64_t batches) {
  Geometry g;
  g.waves = (int)std::max<size_t>(1, std::min<size_t>(4, ((size_t)40 << 10) / per_wave));
  g.lds = per_wave * (size_t)g.waves;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, ((size_t)156 << 10) / (g.lds + 64)));
  const int64_t cus = fdb_scan_default_grid(device) / 2;
  g.blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cus * per_cu, (batches + g.waves - 1) / g.waves));
  return g;
}
bool quads_ok(const void* base, int stride_words, int in_words) { return ((uintptr_t)base & 15u) == 0 && stride_words % 4 == 0 && in_words % 4 == 0; }

}  // namespace

hipError_t fdb_launch_hash_merge(const FdbHashMergeArgs& args, int device, hipStream_t stream) {
  const bool table_src = args.src_table != nullptr;
  if (table_src ? args.src_capacity == 0 : args.n <= 0) return hipSuccess;
  const int in_words = std::max(2, args.in_words > 0 ? args.in_words : (table_src ? args.src_key_words : args.in_key_words));  // (the incoming valid mask is read unconditionally)
  const bool quads = table_src ? quads_ok(args.src_keys, args.src_key_words, in_words) : quads_ok(args.in_keys, args.in_key_words, in_words);
  const bool alias = args.same_layout != 0 && in_words <= args.key_words;
  const int64_t batches = table_src ? (int64_t)((args.src_capacity + 63) / 64) / 16 + 1 : (args.n + 63) / 64;
  const Geometry g = geometry(device, wave_lds_bytes(in_words, args.key_words, alias), batches);
  if (g.lds > ((size_t)150 << 10) || args.key_words % 4 != 0) return hipErrorInvalidValue;
  if (g.lds > ((size_t)48 << 10)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_merge_wave_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_merge_wave_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10);
    (void)hipGetLastError();
  }
  if (table_src) hipLaunchKernelGGL(hash_merge_wave_kernel<true>, dim3(g.blocks), dim3(64 * g.waves), g.lds, stream, args, in_words, quads, alias);
  else hipLaunchKernelGGL(hash_merge_wave_kernel<false>, dim3(g.blocks), dim3(64 * g.waves), g.lds, stream, args, in_words, quads, alias);
  return hipGetLastError();
}

namespace {
struct PartPlan { int in_words; bool quads, alias; Geometry g; };
PartPlan part_plan(int device, const FdbHashPartArgs& a) {
  PartPlan pp;
  pp.in_words = std::max(2, a.in_words > 0 ? a.in_words : a.key_words);
  pp.quads = quads_ok(a.keys, a.key_words, pp.in_words);
  pp.alias = a.same_layout != 0 && pp.in_words <= a.row_words32;
  pp.g = geometry(device, wave_lds_bytes(pp.in_words, a.row_words32, pp.alias) + FDB_PART_EXTRA, (int64_t)((a.capacity + 63) / 64) / 16 + 1);
  return pp;
}
}  // namespace

size_t fdb_hash_partition_scratch_bytes(int device, const FdbHashPartArgs& args) {
  const PartPlan pp = part_plan(device, args);
  return (size_t)pp.g.blocks * pp.g.waves * args.n_parts * (4 + 8) + 256;
}

hipError_t fdb_launch_hash_partition(const FdbHashPartArgs& a, int device, void* scratch, hipStream_t stream) {
  if (a.capacity == 0) return hipSuccess;
  FdbHashPartArgs args = a;
  const PartPlan pp = part_plan(device, args);
  if (args.row_words32 % 4 != 0 || ((uintptr_t)args.out & 15u) != 0 || pp.g.lds > ((size_t)150 << 10)) return hipErrorInvalidValue;
  const int64_t n_waves = (int64_t)pp.g.blocks * pp.g.waves;
  args.wave_bases = reinterpret_cast<unsigned long long*>(scratch);
  args.wave_counts = reinterpret_cast<uint32_t*>(args.wave_bases + n_waves * args.n_parts);
  const dim3 grid(pp.g.blocks), block(64 * pp.g.waves);
  if (pp.g.lds > ((size_t)48 << 10)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_partition_wave_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_partition_wave_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10);
    (void)hipGetLastError();
  }
  hipLaunchKernelGGL(hash_partition_wave_kernel<0>, grid, block, pp.g.lds, stream, args, pp.in_words, pp.quads, pp.alias);
  hipLaunchKernelGGL(partition_bases_kernel, dim3(1), dim3(1024), 0, stream, args.wave_counts, args.wave_bases, args.counts, n_waves, args.n_parts);
  hipLaunchKernelGGL(hash_partition_wave_kernel<1>, grid, block, pp.g.lds, stream, args, pp.in_words, pp.quads, pp.alias);
  return hipGetLastError();
}
