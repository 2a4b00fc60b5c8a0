// fdb_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the fused  PredicateFilter → HashAggregate  scan.
//
// Replaces, per batch, the reference's three hot loops:
//   filter():            roaring bitmap → index ranges → slice + concatenate    (filter.go:276-323)
//   HashAggregate:       per-row hash + Go map probe + builder append           (aggregate.go:398-486)
//   Finish reducers:     per-group array reduce                                 (aggregate.go:734-971)
// with ONE pass over the referenced columns: coalesced 16-byte loads → predicate bit masks in registers →
// group slot from per-dictionary-entry LUTs (LDS copies) → online SUM/COUNT/MIN/MAX into a per-workgroup
// LDS table (ds_add_f64 / ds_add_u64 / ds_min_i64 / ds_max_i64) → one flush per workgroup into the global
// table with global_atomic_*. The path is HBM-bound: no MFMA, no GEMM reshaping.
//
// Geometry: 1024-thread workgroups (16 waves share one LDS table, so flush traffic is per-CU-half, not
// per-wave), persistent grid = 2 workgroups per CU, tiles of 1024 × R rows taken grid-stride so that
// consecutive workgroups (which land on different XCDs, block b → XCD b % 8) stream disjoint 32-64 KiB
// spans. Each lane owns R consecutive rows: a uint32 index column is one or two dwordx4 loads per lane,
// an 8-byte value column R/2 of them, a validity bitmap one byte.
#include <hip/hip_runtime.h>

#define FDB_DEVICE_HELPERS 1
#include <cstdlib>
#include <cstring>

#include "fdb_kernels.h"

namespace {

// fdb_op values used on the device (include/frostdb_amd.h)
constexpr int OP_EQ = 1, OP_NOT_EQ = 2, OP_LT = 3, OP_LT_EQ = 4, OP_GT = 5, OP_GT_EQ = 6;
constexpr int AGG_SUM = 1, AGG_MIN = 2, AGG_MAX = 3, AGG_COUNT = 4;

__device__ __forceinline__ long long f64_to_ordered(double d) {
  long long b = __double_as_longlong(d);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}
// The key a float64 contributes to a MIN / MAX accumulator. A NaN contributes the accumulator's identity, i.e. nothing: the
// reference's `if v < minV` loop (aggregate.go:846-857, :924-934) never replaces its running value by a NaN and never replaces a NaN,
// so a NaN only survives there when it is the group's FIRST value — an answer that depends on row order; here a group's MIN / MAX is
// that of its non-NaN values, and NaN (what the identity decodes to) only if it has no others. −0.0 counts as +0.0 (the reference
// compares them equal and keeps whichever came first).
__device__ __forceinline__ long long f64_minmax_key(double d, bool is_min) {
  d += 0.0;
  const long long k = f64_to_ordered(d);
  return d != d ? (is_min ? FDB_I64_MAX : FDB_I64_MIN) : k;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// Column, bitmap and LUT pointers reach the kernels inside argument blocks, so the compiler sees generic
// pointers and would emit flat_load (LDS-aperture check, counts against lgkmcnt as well as vmcnt). They always
// point to HBM: say so, and get global_load_dwordx4 … nt.
#define FDB_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const FDB_GLOBAL T* as_global(const T* p) {
  return (const FDB_GLOBAL T*)p;
}

template <int R>
__device__ __forceinline__ void load_u32(const uint32_t* __restrict__ p, uint32_t (&v)[R]) {
  if (R == 1) { v[0] = __builtin_nontemporal_load(as_global(p)); return; }
#pragma unroll
  for (int j = 0; j < R / 4; j++) {
    const u32x4 q = __builtin_nontemporal_load(as_global(reinterpret_cast<const u32x4*>(p)) + j);
    v[4 * j + 0] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
  }
}

template <int R>
__device__ __forceinline__ void load_u64(const unsigned long long* __restrict__ p, unsigned long long (&v)[R]) {
  if (R == 1) { v[0] = __builtin_nontemporal_load(as_global(p)); return; }
#pragma unroll
  for (int j = 0; j < R / 2; j++) {
    const u64x2 q = __builtin_nontemporal_load(as_global(reinterpret_cast<const u64x2*>(p)) + j);
    v[2 * j + 0] = q.x; v[2 * j + 1] = q.y;
  }
}

// R validity bits of rows [row0, row0+R); row0 is a multiple of R (R ∈ {4, 8}).
template <int R>
__device__ __forceinline__ uint32_t load_valid(const uint8_t* __restrict__ bm, int64_t row0) {
  const uint32_t b = as_global(bm)[row0 >> 3];
  if (R == 1) return (b >> (row0 & 7)) & 1u;
  if (R == 8) return b;
  return (b >> (row0 & 4)) & 0xFu;
}

template <int R, typename T>
__device__ __forceinline__ uint32_t cmp_mask(const T (&v)[R], T lit, int op) {
  uint32_t m = 0;
  switch (op) {
    case OP_EQ:
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)(v[r] == lit) << r;
      break;
    case OP_NOT_EQ:
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)(v[r] != lit) << r;
      break;
    case OP_LT:
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)(v[r] < lit) << r;
      break;
    case OP_LT_EQ:
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)(v[r] <= lit) << r;
      break;
    case OP_GT:
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)(v[r] > lit) << r;
      break;
    case OP_GT_EQ:
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)(v[r] >= lit) << r;
      break;
    default: break;
  }
  return m;
}

// One predicate leaf over R rows → R-bit mask (bit r = row0 + r satisfies the leaf). NULL rows never match
// a value compare (binaryscalarexpr.go:143-150, :175-177, :215-217).
template <int R>
__device__ __forceinline__ uint32_t eval_leaf(const FdbLeaf& L, int64_t row0, const unsigned char* smem) {
  constexpr uint32_t FULL = (1u << R) - 1u;
  if (L.kind == FDB_LEAF_CONST) return L.op ? FULL : 0u;
  uint32_t valid = FULL;
  if (L.validity != nullptr) valid = load_valid<R>(L.validity, row0);
  if (L.kind == FDB_LEAF_VALIDITY) return L.op ? valid : (~valid & FULL);
  uint32_t m = 0;
  if (L.kind == FDB_LEAF_DICT_LUT || L.kind == FDB_LEAF_DICT_BITS) {
    // Truth table per dictionary entry, evaluated on the host; its last entry answers for NULL rows.
    uint32_t idx[R];
    load_u32<R>(reinterpret_cast<const uint32_t*>(L.values) + row0, idx);
    const uint32_t null_at = L.lut_len - 1u;
#pragma unroll
    for (int r = 0; r < R; r++) idx[r] = ((valid >> r) & 1u) ? idx[r] : null_at;
    if (L.kind == FDB_LEAF_DICT_BITS) {
      const unsigned long long bits = (unsigned long long)L.lit;
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)((bits >> idx[r]) & 1ull) << r;
    } else if (L.lut_lds != FDB_NO_LDS) {
      const unsigned char* lut = smem + L.lut_lds;
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)lut[idx[r]] << r;
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)as_global(L.lut)[idx[r]] << r;
    }
    return m;
  } else {
    unsigned long long raw[R];
    load_u64<R>(reinterpret_cast<const unsigned long long*>(L.values) + row0, raw);
    if (L.kind == FDB_LEAF_CMP_I64) {
      long long v[R];
#pragma unroll
      for (int r = 0; r < R; r++) v[r] = (long long)raw[r];
      m = cmp_mask<R, long long>(v, (long long)L.lit, L.op);
    } else if (L.kind == FDB_LEAF_CMP_U64) {
      m = cmp_mask<R, unsigned long long>(raw, (unsigned long long)L.lit, L.op);
    } else if (L.kind == FDB_LEAF_CMP_F64) {
      double v[R];
#pragma unroll
      for (int r = 0; r < R; r++) v[r] = __longlong_as_double((long long)raw[r]);
      m = cmp_mask<R, double>(v, __longlong_as_double(L.lit), L.op);
    } else {  // FDB_LEAF_CMP_I64_F64
      double v[R];
#pragma unroll
      for (int r = 0; r < R; r++) v[r] = (double)(long long)raw[r];
      m = cmp_mask<R, double>(v, __longlong_as_double(L.lit), L.op);
    }
  }
  return m & valid;
}

// Postfix boolean program over leaf masks. The evaluation stack lives in ONE 64-bit register (8 bits per
// level, depth ≤ 8), so there is no dynamically indexed register array and every branch is wave-uniform.
template <int R>
__device__ __forceinline__ uint32_t eval_filter(const FdbScanArgs& a, int64_t row0, const unsigned char* smem) {
  constexpr uint32_t FULL = (1u << R) - 1u;
  if (a.n_code == 0) return FULL;
  unsigned long long st = 0;
  for (int pc = 0; pc < a.n_code; pc++) {
    const uint32_t c = a.code[pc];
    if (c < 0x80u) {
      st = (st << 8) | (unsigned long long)eval_leaf<R>(a.leaves[c], row0, smem);
    } else {
      const unsigned long long top = st & 0xFFull;
      st >>= 8;
      if (c == FDB_CODE_AND) st = (st & ~0xFFull) | ((st & 0xFFull) & top);
      else st = st | top;
    }
  }
  return (uint32_t)(st & FULL);
}

template <int R>
__device__ __forceinline__ void group_slots(const FdbScanArgs& a, int64_t row0, const unsigned char* smem, uint32_t (&gid)[R]) {
#pragma unroll
  for (int r = 0; r < R; r++) gid[r] = 0;
  for (int g = 0; g < a.n_gcols; g++) {
    const FdbGroupCol& G = a.gcols[g];
    uint32_t valid = (1u << R) - 1u;
    if (G.validity != nullptr) valid = load_valid<R>(G.validity, row0);
    uint32_t idx[R];
    load_u32<R>(G.idx + row0, idx);
    if (G.lut_lds != FDB_NO_LDS) {
      const uint32_t* lut = reinterpret_cast<const uint32_t*>(smem + G.lut_lds);
#pragma unroll
      for (int r = 0; r < R; r++) gid[r] += (((valid >> r) & 1u) ? lut[idx[r]] : 0u) * G.stride;
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) gid[r] += (((valid >> r) & 1u) ? as_global(G.lut)[idx[r]] : 0u) * G.stride;
    }
  }
}

__device__ __forceinline__ unsigned long long agg_identity(int func, int type) {
  if (func == AGG_MIN) return (unsigned long long)FDB_I64_MAX;
  if (func == AGG_MAX) return (unsigned long long)FDB_I64_MIN;
  return 0ull;  // SUM/COUNT: integer 0 and +0.0 share the all-zero pattern
}

// Fused scan. LDS = true: partial aggregates staged in the workgroup's LDS table and flushed once.
template <int R, bool LDS>
__global__ __launch_bounds__(FDB_BLOCK, 8) void scan_dense_kernel(const FdbScanArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr uint32_t FULL = (1u << R) - 1u;
  const int tid = threadIdx.x;
  const uint32_t n_slots = a.n_slots;

  // ---- stage the per-dictionary LUTs in LDS ----------------------------------------------------------
  for (int l = 0; l < a.n_leaves; l++) {
    const FdbLeaf& L = a.leaves[l];
    if (L.kind == FDB_LEAF_DICT_LUT && L.lut_lds != FDB_NO_LDS)
      for (uint32_t i = tid; i < L.lut_len; i += FDB_BLOCK) smem[L.lut_lds + i] = as_global(L.lut)[i];
  }
  for (int g = 0; g < a.n_gcols; g++) {
    const FdbGroupCol& G = a.gcols[g];
    if (G.lut_lds != FDB_NO_LDS) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(smem + G.lut_lds);
      for (uint32_t i = tid; i < G.lut_len; i += FDB_BLOCK) dst[i] = as_global(G.lut)[i];
    }
  }
  uint32_t* l_cnt = reinterpret_cast<uint32_t*>(smem + a.lds_lut_bytes);
  unsigned long long* l_acc =
      reinterpret_cast<unsigned long long*>(smem + a.lds_lut_bytes + (((size_t)n_slots * 4 + 15) & ~(size_t)15));
  if (LDS) {
    for (uint32_t i = tid; i < n_slots; i += FDB_BLOCK) l_cnt[i] = 0;
    for (int j = 0; j < a.n_aggs; j++) {
      const unsigned long long ident = agg_identity(a.aggs[j].func, a.aggs[j].type);
      for (uint32_t i = tid; i < n_slots; i += FDB_BLOCK) l_acc[(size_t)j * n_slots + i] = ident;
    }
  }
  __syncthreads();

  const int64_t tile_rows = (int64_t)FDB_BLOCK * R;
  const int64_t n_tiles = (a.n_rows + tile_rows - 1) / tile_rows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * tile_rows + (int64_t)tid * R;
    uint32_t sel = 0;
    if (row0 < a.n_rows) {
      const int64_t left = a.n_rows - row0;
      const uint32_t in_range = left >= R ? FULL : ((1u << (int)left) - 1u);
      sel = ((a.ablate & 8) ? FULL : eval_filter<R>(a, row0, smem)) & in_range;
    }
    if (__ballot(sel != 0) == 0ull) continue;  // whole wave filtered out: skip the group/value columns
    if (row0 >= a.n_rows) continue;            // (lanes past the end hold sel == 0; keep their loads in bounds)

    uint32_t gid[R];
    if (a.ablate & 4) {
#pragma unroll
      for (int r = 0; r < R; r++) gid[r] = 0;
    } else {
      group_slots<R>(a, row0, smem, gid);
    }

    // occupancy / COUNT
    if (a.ablate & 1) {
    } else if (LDS) {
      if (a.need_count) {
#pragma unroll
        for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicAdd(&l_cnt[gid[r]], 1u);
      } else {
#pragma unroll
        for (int r = 0; r < R; r++) if ((sel >> r) & 1u) l_cnt[gid[r]] = 1u;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicAdd(&a.cnt[gid[r]], 1ull);
    }

    for (int j = 0; j < a.n_aggs; j++) {
      const FdbAgg& A = a.aggs[j];
      if (A.func == AGG_COUNT) continue;  // served by the row count above (CountAggregation counts NULLs too)
      uint32_t valid = FULL;
      if (A.validity != nullptr) valid = load_valid<R>(A.validity, row0);
      unsigned long long raw[R];
      load_u64<R>(reinterpret_cast<const unsigned long long*>(A.values) + row0, raw);
      // A NULL contributes the builder's zeroed slot: 0 to SUM *and* to MIN/MAX (aggregate.go:784-935 read raw
      // values; pqarrow/builder/optbuilders.go:337-340 zero-fills).
#pragma unroll
      for (int r = 0; r < R; r++) if (!((valid >> r) & 1u)) raw[r] = A.null_value;
      unsigned long long* acc = LDS ? (l_acc + (size_t)j * n_slots) : A.acc;
      if (a.ablate & 2) {
        unsigned long long x = 0;
#pragma unroll
        for (int r = 0; r < R; r++) x ^= raw[r];
        if (x == 0x123456789abcdefull) acc[0] = x;
      } else if (A.func == AGG_SUM) {
        if (A.type == FDB_T_F64) {
#pragma unroll
          for (int r = 0; r < R; r++)
            if ((sel >> r) & 1u) atomicAdd(reinterpret_cast<double*>(acc) + gid[r], __longlong_as_double((long long)raw[r]));
        } else {
#pragma unroll
          for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicAdd(acc + gid[r], raw[r]);
        }
      } else {
        long long key[R];
        if (A.type == FDB_T_F64) {
#pragma unroll
          for (int r = 0; r < R; r++) key[r] = f64_minmax_key(__longlong_as_double((long long)raw[r]), A.func == AGG_MIN);
        } else {
#pragma unroll
          for (int r = 0; r < R; r++) key[r] = (long long)raw[r];
        }
        if (A.func == AGG_MIN) {
#pragma unroll
          for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicMin(reinterpret_cast<long long*>(acc) + gid[r], key[r]);
        } else {
#pragma unroll
          for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicMax(reinterpret_cast<long long*>(acc) + gid[r], key[r]);
        }
      }
    }
  }

  if (LDS && a.partials != nullptr) {
    // Flush without atomics: this workgroup's table goes out as plain coalesced stores; a small second kernel
    // folds the tables in workgroup order (512 workgroups hammering 1 025 addresses with atomics cost ≈10 µs).
    __syncthreads();
    unsigned long long* out = a.partials + (size_t)blockIdx.x * (size_t)(1 + a.n_aggs) * n_slots;
    for (uint32_t i = tid; i < n_slots; i += FDB_BLOCK) out[i] = (unsigned long long)l_cnt[i];
    for (int j = 0; j < a.n_aggs; j++) {
      if (a.aggs[j].func == AGG_COUNT) continue;
      for (uint32_t i = tid; i < n_slots; i += FDB_BLOCK) out[(size_t)(1 + j) * n_slots + i] = l_acc[(size_t)j * n_slots + i];
    }
  } else if (LDS) {
    __syncthreads();
    for (uint32_t i = tid; i < n_slots; i += FDB_BLOCK) {
      const uint32_t c = l_cnt[i];
      if (c == 0) continue;
      atomicAdd(&a.cnt[i], (unsigned long long)c);
      for (int j = 0; j < a.n_aggs; j++) {
        const FdbAgg& A = a.aggs[j];
        if (A.func == AGG_COUNT) continue;
        const unsigned long long v = l_acc[(size_t)j * n_slots + i];
        if (A.func == AGG_SUM) {
          if (A.type == FDB_T_F64) atomicAdd(reinterpret_cast<double*>(A.acc) + i, __longlong_as_double((long long)v));
          else atomicAdd(A.acc + i, v);
        } else if (A.func == AGG_MIN) {
          atomicMin(reinterpret_cast<long long*>(A.acc) + i, (long long)v);
        } else {
          atomicMax(reinterpret_cast<long long*>(A.acc) + i, (long long)v);
        }
      }
    }
  }
}


// =========================================================================================================
// Slot kernel: same semantics as scan_dense_kernel, but every referenced column of a tile is LOADED FIRST
// (≤ FDB_MAX_C4 four-byte + FDB_MAX_C8 eight-byte column slots, fully unrolled, wave-uniform predicates) and
// only then consumed. The sequential kernel keeps one column in flight per wave (16 B/lane): with 32 waves
// per CU that is 32 KiB per CU, below the ≈50 KiB HBM latency×bandwidth product, and it measured 6.0 TB/s
// whatever work was ablated. Here a cfg-2 wave has 66 B/lane in flight.
// =========================================================================================================
#define SLOT_BLOCK 512
#define SLOT_MAX_LEAVES 6
#define SLOT_MAX_GCOLS 2
#define SLOT_MAX_AGGS 6

template <int NC4, int NC8>
struct SlotRegs {
  u32x4 r4[NC4 > 0 ? NC4 : 1];
  u64x2 r8[NC8 > 0 ? NC8 : 1][2];
  uint32_t v4[NC4 > 0 ? NC4 : 1];
  uint32_t v8[NC8 > 0 ? NC8 : 1];
};

// The plan of ONE record, decoded once (when a workgroup enters the record) into wave-uniform values that
// stay in scalar registers across the tile loop. Everything is indexed with compile-time constants, so the
// per-tile code contains no descriptor loads at all: the sequential kernel and the first slot kernel spent
// ≈40 dependent scalar loads and ≈300 scalar ALU instructions per 256-row tile re-reading their argument block.
template <int NC4, int NC8, int L4, int L8>
struct PlanRegs {
  int64_t n_rows, tile_begin;
  const void* v4[NC4 > 0 ? NC4 : 1]; const uint8_t* b4[NC4 > 0 ? NC4 : 1];
  const void* v8[NC8 > 0 ? NC8 : 1]; const uint8_t* b8[NC8 > 0 ? NC8 : 1];
  const void* lv4[L4 > 0 ? L4 : 1]; const uint8_t* lb4[L4 > 0 ? L4 : 1];
  const void* lv8[L8 > 0 ? L8 : 1]; const uint8_t* lb8[L8 > 0 ? L8 : 1];
  int n_leaves, n_gcols, need_count;
  struct { int kind, slot, wide, op; uint32_t lut_lds, lut_len, ops_after; long long lit; const uint8_t* lut; } leaf[SLOT_MAX_LEAVES];
  struct { int slot; uint32_t lut_lds, stride; const uint32_t* lut; } gcol[SLOT_MAX_GCOLS];
  struct { int func, type, slot; unsigned long long* acc; unsigned long long null_value; } agg[SLOT_MAX_AGGS];
  unsigned long long* cnt;
};

template <int NC4, int NC8, int L4, int L8>
__device__ __forceinline__ void decode_plan(const FdbScanArgs& a, PlanRegs<NC4, NC8, L4, L8>& P) {
  P.n_rows = a.n_rows; P.tile_begin = a.tile_begin;
#pragma unroll
  for (int s = 0; s < NC4; s++) { P.v4[s] = s < a.n_c4 ? a.c4[s].values : nullptr; P.b4[s] = s < a.n_c4 ? a.c4[s].validity : nullptr; }
#pragma unroll
  for (int s = 0; s < NC8; s++) { P.v8[s] = s < a.n_c8 ? a.c8[s].values : nullptr; P.b8[s] = s < a.n_c8 ? a.c8[s].validity : nullptr; }
#pragma unroll
  for (int s = 0; s < L4; s++) { P.lv4[s] = s < a.n_l4 ? a.l4[s].values : nullptr; P.lb4[s] = s < a.n_l4 ? a.l4[s].validity : nullptr; }
#pragma unroll
  for (int s = 0; s < L8; s++) { P.lv8[s] = s < a.n_l8 ? a.l8[s].values : nullptr; P.lb8[s] = s < a.n_l8 ? a.l8[s].validity : nullptr; }
  P.n_leaves = a.n_leaves; P.n_gcols = a.n_gcols; P.need_count = a.need_count; P.cnt = a.cnt;
#pragma unroll
  for (int l = 0; l < SLOT_MAX_LEAVES; l++) {
    const FdbLeaf& L = a.leaves[l];
    P.leaf[l].kind = L.kind; P.leaf[l].slot = L.slot; P.leaf[l].wide = L.wide; P.leaf[l].op = L.op;
    P.leaf[l].lut_lds = L.lut_lds; P.leaf[l].lut_len = L.lut_len; P.leaf[l].ops_after = a.ops_after[l]; P.leaf[l].lit = L.lit;
    P.leaf[l].lut = L.lut;
  }
#pragma unroll
  for (int g = 0; g < SLOT_MAX_GCOLS; g++) {
    P.gcol[g].slot = a.gcols[g].slot; P.gcol[g].lut_lds = a.gcols[g].lut_lds; P.gcol[g].stride = a.gcols[g].stride; P.gcol[g].lut = a.gcols[g].lut;
  }
#pragma unroll
  for (int j = 0; j < SLOT_MAX_AGGS; j++) {
    P.agg[j].func = a.aggs[j].func; P.agg[j].type = a.aggs[j].type; P.agg[j].slot = a.aggs[j].slot; P.agg[j].acc = a.aggs[j].acc; P.agg[j].null_value = a.aggs[j].null_value;
  }
}

template <int N4, int N8>
__device__ __forceinline__ void slot_load(const void* const (&v4)[N4 > 0 ? N4 : 1], const uint8_t* const (&b4)[N4 > 0 ? N4 : 1],
                                          const void* const (&v8)[N8 > 0 ? N8 : 1], const uint8_t* const (&b8)[N8 > 0 ? N8 : 1], int64_t row0,
                                          SlotRegs<N4, N8>& S) {
#pragma unroll
  for (int s = 0; s < N4; s++) {
    S.r4[s] = u32x4{0, 0, 0, 0};
    S.v4[s] = 0xFu;
    if (v4[s] != nullptr) S.r4[s] = __builtin_nontemporal_load(as_global(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint32_t*>(v4[s]) + row0)));
    if (b4[s] != nullptr) S.v4[s] = load_valid<4>(b4[s], row0);
  }
#pragma unroll
  for (int s = 0; s < N8; s++) {
    S.r8[s][0] = u64x2{0, 0};
    S.r8[s][1] = u64x2{0, 0};
    S.v8[s] = 0xFu;
    if (v8[s] != nullptr) {
      const FDB_GLOBAL u64x2* p = as_global(reinterpret_cast<const u64x2*>(reinterpret_cast<const unsigned long long*>(v8[s]) + row0));
      S.r8[s][0] = __builtin_nontemporal_load(p);
      S.r8[s][1] = __builtin_nontemporal_load(p + 1);
    }
    if (b8[s] != nullptr) S.v8[s] = load_valid<4>(b8[s], row0);
  }
}

template <int NC4, int NC8>
__device__ __forceinline__ void pick4(const SlotRegs<NC4, NC8>& S, int slot, uint32_t (&idx)[4], uint32_t& valid) {
  u32x4 q = S.r4[0];
  valid = S.v4[0];
#pragma unroll
  for (int s = 1; s < NC4; s++)
    if (slot == s) { q = S.r4[s]; valid = S.v4[s]; }
  idx[0] = q.x; idx[1] = q.y; idx[2] = q.z; idx[3] = q.w;
}

template <int NC4, int NC8>
__device__ __forceinline__ void pick8(const SlotRegs<NC4, NC8>& S, int slot, unsigned long long (&raw)[4], uint32_t& valid) {
  u64x2 q0 = S.r8[0][0], q1 = S.r8[0][1];
  valid = S.v8[0];
#pragma unroll
  for (int s = 1; s < NC8; s++)
    if (slot == s) { q0 = S.r8[s][0]; q1 = S.r8[s][1]; valid = S.v8[s]; }
  raw[0] = q0.x; raw[1] = q0.y; raw[2] = q1.x; raw[3] = q1.y;
}

// One leaf over the 4 rows of a lane; every argument except S is wave-uniform and register-resident.
template <int NC4, int NC8>
__device__ __forceinline__ uint32_t slot_eval_leaf(int kind, int slot, int wide, int op, uint32_t lut_lds, uint32_t lut_len, long long lit,
                                                   const uint8_t* lut_g, const SlotRegs<NC4, NC8>& S, const unsigned char* smem) {
  constexpr int R = 4;
  constexpr uint32_t FULL = 0xFu;
  if (kind == FDB_LEAF_CONST) return op ? FULL : 0u;
  uint32_t valid, m = 0;
  if (!wide) {
    uint32_t idx[R];
    pick4(S, slot, idx, valid);
    if (kind == FDB_LEAF_VALIDITY) return op ? valid : (~valid & FULL);
    const uint32_t null_at = lut_len - 1u;
#pragma unroll
    for (int r = 0; r < R; r++) idx[r] = ((valid >> r) & 1u) ? idx[r] : null_at;
    if (kind == FDB_LEAF_DICT_BITS) {
      const unsigned long long bits = (unsigned long long)lit;
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)((bits >> idx[r]) & 1ull) << r;
    } else if (lut_lds != FDB_NO_LDS) {
      const unsigned char* lut = smem + lut_lds;
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)lut[idx[r]] << r;
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) m |= (uint32_t)as_global(lut_g)[idx[r]] << r;
    }
    return m;
  }
  unsigned long long raw[R];
  pick8(S, slot, raw, valid);
  if (kind == FDB_LEAF_VALIDITY) return op ? valid : (~valid & FULL);
  if (kind == FDB_LEAF_CMP_I64) {
    long long v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = (long long)raw[r];
    m = cmp_mask<R, long long>(v, lit, op);
  } else if (kind == FDB_LEAF_CMP_U64) {
    m = cmp_mask<R, unsigned long long>(raw, (unsigned long long)lit, op);
  } else if (kind == FDB_LEAF_CMP_F64) {
    double v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = __longlong_as_double((long long)raw[r]);
    m = cmp_mask<R, double>(v, __longlong_as_double(lit), op);
  } else {
    double v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = (double)(long long)raw[r];
    m = cmp_mask<R, double>(v, __longlong_as_double(lit), op);
  }
  return m & valid;
}

// =========================================================================================================
// Slot kernel. Per tile: (1) issue the loads of every column the FILTER reads (≤ NC4 four-byte + NC8 eight-byte
// column slots, unrolled, wave-uniform predicates) — or of every column at all in the single-phase instance;
// (2) evaluate the predicate; (3) two-phase instance only: load the group-by / aggregate columns (late
// materialisation: nothing is fetched for a lane whose 4 rows were all filtered out); (4) accumulate into the
// workgroup's LDS table. The sequential kernel above keeps ONE column in flight per wave; this one keeps all of
// a phase's columns in flight, and carries its decoded plan in scalar registers.
// =========================================================================================================
template <bool LDS, int NC4, int NC8, int L4, int L8, int BLK>
__global__ __launch_bounds__(BLK) void scan_slots_kernel(const FdbScanArgs* __restrict__ parts, const int n_parts,
                                                         const int64_t total_tiles, const FdbScanArgs c) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int R = 4;
  constexpr uint32_t FULL = 0xFu;
  constexpr bool TWO_PHASE = (L4 + L8) > 0;
  const int tid = threadIdx.x;
  const uint32_t n_slots = c.n_slots;

  uint32_t* l_cnt = reinterpret_cast<uint32_t*>(smem + c.lds_lut_bytes);
  unsigned long long* l_acc =
      reinterpret_cast<unsigned long long*>(smem + c.lds_lut_bytes + (((size_t)n_slots * 4 + 15) & ~(size_t)15));
  if (LDS) {
    for (uint32_t i = tid; i < n_slots; i += BLK) l_cnt[i] = 0;
    for (int j = 0; j < c.n_aggs; j++) {
      const unsigned long long ident = agg_identity(c.aggs[j].func, c.aggs[j].type);
      for (uint32_t i = tid; i < n_slots; i += BLK) l_acc[(size_t)j * n_slots + i] = ident;
    }
  }

  // Global tiles are numbered across the records of this launch; a workgroup walks them grid-stride, so it
  // visits the records in order; on entering a record it decodes that record's plan into registers and, if the
  // record's LUT set differs from the one staged in LDS, re-stages it.
  constexpr int64_t tile_rows = (int64_t)BLK * R;
  PlanRegs<NC4, NC8, L4, L8> P;
  int part = -1, lut_class = -1;
  int64_t tile_end = 0;
  for (int64_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    if (part < 0 || tile >= tile_end) {
      int np = part < 0 ? 0 : part;
      while (np + 1 < n_parts && tile >= parts[np].tile_end) np++;
      part = np;
      const FdbScanArgs& pa = parts[part];
      tile_end = pa.tile_end;
      decode_plan<NC4, NC8, L4, L8>(pa, P);
      if (pa.lut_class != lut_class) {
        lut_class = pa.lut_class;
        __syncthreads();  // every wave is done with the previous record's LUTs (and with the table init)
        for (int l = 0; l < pa.n_leaves; l++) {
          const FdbLeaf& L = pa.leaves[l];
          if (L.kind == FDB_LEAF_DICT_LUT && L.lut_lds != FDB_NO_LDS)
            for (uint32_t i = tid; i < L.lut_len; i += BLK) smem[L.lut_lds + i] = as_global(L.lut)[i];
        }
        for (int g = 0; g < pa.n_gcols; g++) {
          const FdbGroupCol& G = pa.gcols[g];
          if (G.lut_lds != FDB_NO_LDS) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(smem + G.lut_lds);
            for (uint32_t i = tid; i < G.lut_len; i += BLK) dst[i] = as_global(G.lut)[i];
          }
        }
        __syncthreads();
      }
    }
    const int64_t row0 = (tile - P.tile_begin) * tile_rows + (int64_t)tid * R;
    if (row0 >= P.n_rows) continue;

    // ---- (1) loads of the filter's columns (single phase: of every column) ---------------------------------------
    SlotRegs<NC4, NC8> S;
    slot_load<NC4, NC8>(P.v4, P.b4, P.v8, P.b8, row0, S);
    const int64_t left = P.n_rows - row0;
    uint32_t sel = left >= R ? FULL : ((1u << (int)left) - 1u);

    // ---- (2) predicate: leaves in postfix order, each followed by the AND/OR ops that consume it --------------------
    if (P.n_leaves != 0) {
      unsigned long long st = 0;
#pragma unroll
      for (int l = 0; l < SLOT_MAX_LEAVES; l++) {
        if (l < P.n_leaves) {
          st = (st << 8) | (unsigned long long)slot_eval_leaf<NC4, NC8>(P.leaf[l].kind, P.leaf[l].slot, P.leaf[l].wide, P.leaf[l].op, P.leaf[l].lut_lds,
                                                                        P.leaf[l].lut_len, P.leaf[l].lit, P.leaf[l].lut, S, smem);
          uint32_t ops = P.leaf[l].ops_after >> 4;  // [3:0] count, then 2 bits per op: 1 AND, 2 OR
          for (uint32_t k = P.leaf[l].ops_after & 0xFu; k != 0; k--, ops >>= 2) {
            const unsigned long long top = st & 0xFFull;
            st >>= 8;
            if ((ops & 3u) == 1u) st = (st & ~0xFFull) | ((st & 0xFFull) & top);
            else st = st | top;
          }
        }
      }
      sel &= (uint32_t)st;
    }
    if (sel == 0) continue;

    // ---- (3) late materialisation ---------------------------------------------------------------------------------
    SlotRegs<L4, L8> SL;
    if (TWO_PHASE) slot_load<L4, L8>(P.lv4, P.lb4, P.lv8, P.lb8, row0, SL);

    // ---- (4) group slot + aggregates -------------------------------------------------------------------------------
    uint32_t gid[R] = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < SLOT_MAX_GCOLS; g++) {
      if (g < P.n_gcols) {
        uint32_t idx[R], valid;
        if (TWO_PHASE) pick4(SL, P.gcol[g].slot, idx, valid); else pick4(S, P.gcol[g].slot, idx, valid);
        if (P.gcol[g].lut_lds != FDB_NO_LDS) {
          const uint32_t* lut = reinterpret_cast<const uint32_t*>(smem + P.gcol[g].lut_lds);
#pragma unroll
          for (int r = 0; r < R; r++) gid[r] += (((valid >> r) & 1u) ? lut[idx[r]] : 0u) * P.gcol[g].stride;
        } else {
#pragma unroll
          for (int r = 0; r < R; r++) gid[r] += (((valid >> r) & 1u) ? as_global(P.gcol[g].lut)[idx[r]] : 0u) * P.gcol[g].stride;
        }
      }
    }
    if (LDS) {
      if (P.need_count) {
#pragma unroll
        for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicAdd(&l_cnt[gid[r]], 1u);
      } else {
#pragma unroll
        for (int r = 0; r < R; r++) if ((sel >> r) & 1u) l_cnt[gid[r]] = 1u;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicAdd(&P.cnt[gid[r]], 1ull);
    }
#pragma unroll
    for (int j = 0; j < SLOT_MAX_AGGS; j++) {
      if (j < c.n_aggs && P.agg[j].func != AGG_COUNT) {
        unsigned long long* acc = LDS ? (l_acc + (size_t)j * n_slots) : P.agg[j].acc;
        unsigned long long raw[R];
        uint32_t valid;
        if (TWO_PHASE) pick8(SL, P.agg[j].slot, raw, valid); else pick8(S, P.agg[j].slot, raw, valid);
        // a NULL contributes the builder's zeroed slot (aggregate.go:784-935, optbuilders.go:337-340)
#pragma unroll
        for (int r = 0; r < R; r++) if (!((valid >> r) & 1u)) raw[r] = P.agg[j].null_value;
        if (P.agg[j].func == AGG_SUM) {
          if (P.agg[j].type == FDB_T_F64) {
#pragma unroll
            for (int r = 0; r < R; r++)
              if ((sel >> r) & 1u) atomicAdd(reinterpret_cast<double*>(acc) + gid[r], __longlong_as_double((long long)raw[r]));
          } else {
#pragma unroll
            for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicAdd(acc + gid[r], raw[r]);
          }
        } else {
          long long key[R];
          if (P.agg[j].type == FDB_T_F64) {
#pragma unroll
            for (int r = 0; r < R; r++) key[r] = f64_minmax_key(__longlong_as_double((long long)raw[r]), P.agg[j].func == AGG_MIN);
          } else {
#pragma unroll
            for (int r = 0; r < R; r++) key[r] = (long long)raw[r];
          }
          if (P.agg[j].func == AGG_MIN) {
#pragma unroll
            for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicMin(reinterpret_cast<long long*>(acc) + gid[r], key[r]);
          } else {
#pragma unroll
            for (int r = 0; r < R; r++) if ((sel >> r) & 1u) atomicMax(reinterpret_cast<long long*>(acc) + gid[r], key[r]);
          }
        }
      }
    }
  }

  if (LDS && c.partials != nullptr) {
    // Flush without atomics: this workgroup's table goes out as plain coalesced stores; a small second kernel
    // folds the tables (1 024 workgroups hammering 1 025 addresses with atomics at the same moment is slower).
    __syncthreads();
    unsigned long long* out = c.partials + (size_t)blockIdx.x * (size_t)(1 + c.n_aggs) * n_slots;
    for (uint32_t i = tid; i < n_slots; i += BLK) out[i] = (unsigned long long)l_cnt[i];
    for (int j = 0; j < c.n_aggs; j++) {
      if (c.aggs[j].func == AGG_COUNT) continue;
      for (uint32_t i = tid; i < n_slots; i += BLK) out[(size_t)(1 + j) * n_slots + i] = l_acc[(size_t)j * n_slots + i];
    }
  } else if (LDS) {
    __syncthreads();
    for (uint32_t i = tid; i < n_slots; i += BLK) {
      const uint32_t n_sel = l_cnt[i];
      if (n_sel == 0) continue;
      atomicAdd(&c.cnt[i], (unsigned long long)n_sel);
      for (int j = 0; j < c.n_aggs; j++) {
        const FdbAgg& A = c.aggs[j];
        if (A.func == AGG_COUNT) continue;
        const unsigned long long v = l_acc[(size_t)j * n_slots + i];
        if (A.func == AGG_SUM) {
          if (A.type == FDB_T_F64) atomicAdd(reinterpret_cast<double*>(A.acc) + i, __longlong_as_double((long long)v));
          else atomicAdd(A.acc + i, v);
        } else if (A.func == AGG_MIN) {
          atomicMin(reinterpret_cast<long long*>(A.acc) + i, (long long)v);
        } else {
          atomicMax(reinterpret_cast<long long*>(A.acc) + i, (long long)v);
        }
      }
    }
  }
}

// =========================================================================================================
// High-cardinality path (cfg 5: 32 label columns, 10 M groups): a global open-addressing table keyed by a
// 128-bit fingerprint of the key tuple. Per selected row: fold every group column's key id into two independent
// 64-bit hashes, probe linearly with a 64-bit CAS on the low half, confirm on the high half, then update the
// group's count and accumulators — which sit in the SAME 32/64-byte entry as the fingerprint, so a row costs one
// random sector of table traffic. (The reference keys its Go map by a single 64-bit hash and never compares keys,
// aggregate.go:130,:411; 128 bits make a false merge practically impossible: ≈ n²/2¹²⁹.) The inserting lane also
// writes the group's key tuple (dictionary key ids / int64 values) once, for Finish.
// No LDS staging of aggregates here: with ~10 rows per group spread uniformly, a per-workgroup cache never hits.
// =========================================================================================================
// (fingerprint and probe helpers: fdb_kernels.h, shared with the run-time generated hash kernels)
// Column descriptors are read through the constant address space: the index is wave-uniform, so every field becomes a
// scalar load (SGPR) instead of a 64-lane vector load of one address — the per-thread descriptor walk of the merge
// kernel was 99 % of its time (136 ms → ≈2 ms for 5 M entries × 32 columns).
typedef const __attribute__((address_space(4))) FdbHashCol* ConstHashCols;
__device__ __forceinline__ FdbHashCol load_col(const FdbHashCol* cols, int c) {
  ConstHashCols q = (ConstHashCols)cols;
  FdbHashCol C;
  C.values = q[c].values; C.validity = q[c].validity; C.lut = q[c].lut; C.lut_len = q[c].lut_len; C.lut_lds = q[c].lut_lds;
  C.kind = q[c].kind; C.word = q[c].word; C.gi = q[c].gi; C.src_word = q[c].src_word; C.k1 = q[c].k1; C.k2 = q[c].k2;
  return C;
}
// One row per lane. While folding the key columns into the fingerprint each lane also parks its key tuple in LDS
// ([word][lane], conflict-free), so that a lane that turns out to be the FIRST to see its group can write the tuple to
// the key store without re-reading 32 columns (≈10 % of the rows of cfg 5 create a group).
#define HASH_UNROLL 8
__global__ __launch_bounds__(FDB_HASH_BLOCK) void scan_hash_kernel(const FdbHashArgs h) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ unsigned int s_new;
  constexpr int R = 1;
  const FdbScanArgs& a = h.base;
  const int tid = threadIdx.x;
  if (tid == 0) s_new = 0;
  for (int l = 0; l < a.n_leaves; l++) {
    const FdbLeaf& L = a.leaves[l];
    if (L.kind == FDB_LEAF_DICT_LUT && L.lut_lds != FDB_NO_LDS)
      for (uint32_t i = tid; i < L.lut_len; i += FDB_HASH_BLOCK) smem[L.lut_lds + i] = as_global(L.lut)[i];
  }
  for (int c = 0; c < h.n_hcols; c++) {
    const FdbHashCol C = load_col(h.hcols, c);
    if (C.kind == 0 && C.lut_lds != FDB_NO_LDS && C.lut != nullptr) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(smem + C.lut_lds);
      for (uint32_t i = tid; i < C.lut_len; i += FDB_HASH_BLOCK) dst[i] = as_global(C.lut)[i];
    }
  }
  uint32_t* kstage = reinterpret_cast<uint32_t*>(smem + a.lds_lut_bytes);  // [key_words][FDB_HASH_BLOCK]
  __syncthreads();

  const int ew = h.entry_words, kw = h.key_words;
  const int64_t n_tiles = (h.row_end - h.row_begin + FDB_HASH_BLOCK - 1) / FDB_HASH_BLOCK;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = h.row_begin + tile * FDB_HASH_BLOCK + tid;
    if (row >= h.row_end) continue;
    if (a.n_code != 0 && (eval_filter<R>(a, row, smem) & 1u) == 0u) continue;

    unsigned long long h1 = 0, h2 = 0, vmask = 0;
    for (int w = 2; w < kw; w++) kstage[w * FDB_HASH_BLOCK + tid] = 0u;  // columns this record does not carry are NULL
    // Key columns in groups of HASH_UNROLL: the descriptors of a group are fetched together (one scalar-load latency),
    // then all its index / validity loads are issued together (one HBM latency), then the group is folded in. One
    // column at a time this loop was a chain of 32 × (descriptor → load → LUT) latencies per row.
    for (int c0 = 0; c0 < h.n_hcols; c0 += HASH_UNROLL) {
      const void* vals[HASH_UNROLL]; const uint8_t* vbm[HASH_UNROLL]; const uint32_t* lutg[HASH_UNROLL];
      uint32_t lds[HASH_UNROLL]; int kind[HASH_UNROLL], word[HASH_UNROLL], gi[HASH_UNROLL];
      unsigned long long k1[HASH_UNROLL], k2[HASH_UNROLL];
#pragma unroll
      for (int u = 0; u < HASH_UNROLL; u++) {
        const FdbHashCol C = load_col(h.hcols, c0 + u < h.n_hcols ? c0 + u : h.n_hcols - 1);
        vals[u] = C.values; vbm[u] = C.validity; lutg[u] = C.lut; lds[u] = C.lut_lds; kind[u] = c0 + u < h.n_hcols ? C.kind : -1;
        word[u] = C.word; gi[u] = C.gi; k1[u] = C.k1; k2[u] = C.k2;
      }
      uint32_t idx[HASH_UNROLL], vb[HASH_UNROLL];
      unsigned long long wide[HASH_UNROLL];
#pragma unroll
      for (int u = 0; u < HASH_UNROLL; u++) {
        idx[u] = 0; vb[u] = 1; wide[u] = 0;
        if (kind[u] < 0) continue;
        if (vbm[u] != nullptr) vb[u] = as_global(vbm[u])[row >> 3];
        if (kind[u] == 0) idx[u] = __builtin_nontemporal_load(as_global(reinterpret_cast<const uint32_t*>(vals[u]) + row));
        else wide[u] = __builtin_nontemporal_load(as_global(reinterpret_cast<const unsigned long long*>(vals[u]) + row));
      }
#pragma unroll
      for (int u = 0; u < HASH_UNROLL; u++) {
        if (kind[u] < 0) continue;
        const bool valid = vbm[u] == nullptr || ((vb[u] >> (row & 7)) & 1u);
        if (kind[u] == 0) {
          uint32_t id = 0;
          if (valid) id = lutg[u] == nullptr ? idx[u] + 1u : lds[u] != FDB_NO_LDS ? reinterpret_cast<const uint32_t*>(smem + lds[u])[idx[u]] : as_global(lutg[u])[idx[u]];  // (no table: key id = index + 1)
          if (!(a.ablate & 4)) kstage[word[u] * FDB_HASH_BLOCK + tid] = id;
          fp_add32(h1, h2, k1[u], k2[u], id);  // id 0 (NULL) contributes nothing
          if (id != 0) vmask |= 1ull << gi[u];
        } else {
          const unsigned long long x = valid ? wide[u] : 0ull;
          kstage[word[u] * FDB_HASH_BLOCK + tid] = (uint32_t)x;
          kstage[(word[u] + 1) * FDB_HASH_BLOCK + tid] = (uint32_t)(x >> 32);
          // (NULL and the value 0 hash alike in the reference — dynparquet/hashed.go:254-272, aggregate.go:398-409 — so neither
          // contributes to the fingerprint; the valid bit keeps which of the two the inserting row was)
          if (valid) { if (x != 0ull) fp_add(h1, h2, k1[u], k2[u], x); vmask |= 1ull << gi[u]; }
        }
      }
    }
    fp_final(h1, h2);
    if (a.ablate & 1) { if ((h1 ^ h2) == 0x1234567ull) h.table[0] = vmask; continue; }  // tuning aid: stream + fingerprint only
    bool inserted;
    const uint64_t slot = hash_find_or_insert(h.table, h.mask, ew, h1, h2, inserted);
    if (inserted) {
      uint32_t* dst = h.keys + slot * (uint64_t)kw;
      dst[0] = (uint32_t)vmask; dst[1] = (uint32_t)(vmask >> 32);
      for (int w = 2; w < kw; w++) dst[w] = kstage[w * FDB_HASH_BLOCK + tid];
      atomicAdd(&s_new, 1u);
    }
    unsigned long long* e = h.table + slot * (uint64_t)ew;
    if (a.ablate & 2) continue;  // tuning aid: probe only
    atomicAdd(e + 2, 1ull);
    for (int j = 0; j < a.n_aggs; j++) {
      const FdbAgg& A = a.aggs[j];
      if (A.func == AGG_COUNT) continue;
      const bool valid = A.validity == nullptr || load_valid<R>(A.validity, row) != 0u;
      unsigned long long raw[R];
      load_u64<R>(reinterpret_cast<const unsigned long long*>(A.values) + row, raw);
      const unsigned long long x = valid ? raw[0] : A.null_value;  // NULL ⇒ the builder's zeroed slot (or the composite reducers' value)
      unsigned long long* acc = e + 3 + j;
      if (A.func == AGG_SUM) {
        if (A.type == FDB_T_F64) atomicAdd(reinterpret_cast<double*>(acc), __longlong_as_double((long long)x));
        else atomicAdd(acc, x);
      } else {
        const long long key = A.type == FDB_T_F64 ? f64_minmax_key(__longlong_as_double((long long)x), A.func == AGG_MIN) : (long long)x;
        if (A.func == AGG_MIN) atomicMin(reinterpret_cast<long long*>(acc), key);
        else atomicMax(reinterpret_cast<long long*>(acc), key);
      }
    }
  }
  __syncthreads();
  if (tid == 0 && s_new != 0) atomicAdd(h.n_groups, (unsigned long long)s_new);
}

struct HashIdents { unsigned long long v[FDB_MAX_AGGS]; };
__global__ void hash_init_kernel(unsigned long long* table, uint64_t total_words, int ew, int n_aggs, HashIdents id) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_words; i += (uint64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % (uint64_t)ew);
    table[i] = (w >= 3 && w < 3 + n_aggs) ? id.v[w - 3] : 0ull;
  }
}

__global__ void hash_rehash_kernel(const unsigned long long* old_table, const uint32_t* old_keys, uint64_t old_capacity, int okw, int oused,
                                   unsigned long long* new_table, uint32_t* new_keys, uint64_t new_mask, int ew, int kw) {
  // A lane claims the new slot of "its" old entry (CAS on the fingerprint, linear probing) and copies the entry's words; the KEY
  // TUPLES — tens of words each — are then copied by the whole wave, one tuple at a time with lane w on word w: coalesced reads and
  // writes instead of every lane walking its own 136 bytes (3.4 M entries of cfg 5: 2.9 ms → a fraction of that).
  const uint64_t n_round = (old_capacity + 63) & ~(uint64_t)63;
  const int lane = threadIdx.x & 63;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long h1 = 0;
    uint64_t slot = 0;
    if (i < old_capacity) {
      const unsigned long long* e = old_table + i * (uint64_t)ew;
      h1 = e[0];
      if (h1 != 0) {
        slot = h1 & new_mask;
        for (;;) {  // fingerprints are unique in the old table: claim the first empty slot
          if (atomicCAS(new_table + slot * (uint64_t)ew, 0ull, h1) == 0ull) break;
          slot = (slot + 1) & new_mask;
        }
        unsigned long long* d = new_table + slot * (uint64_t)ew;
        for (int w = 1; w < ew; w++) d[w] = e[w];
      }
    }
    unsigned long long todo = __ballot(h1 != 0);
    const uint64_t wave_first = i - (uint64_t)lane;
    while (todo != 0ull) {
      const int src_lane = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const uint64_t src = wave_first + (uint64_t)src_lane;
      const uint64_t dst = (uint64_t)__shfl((unsigned long long)slot, src_lane, 64);
      for (int w = lane; w < kw; w += 64) new_keys[dst * (uint64_t)kw + w] = w < oused ? old_keys[src * (uint64_t)okw + w] : 0u;  // columns added since: NULL (the old tuple's tail padding is not a column)
    }
  }
}

// Occupied entries per 64-slot chunk (one wave per chunk): the exclusive scan of these counts (hash_scan_chunks) gives every
// chunk its first output row, so compaction places entries in SLOT ORDER — the same order for every call on an unchanged table
// (keys and every aggregate column of a merge line up), with no atomics.
__global__ void hash_chunk_counts_kernel(const unsigned long long* table, uint64_t capacity, int ew, uint32_t* counts) {
  const uint64_t n_round = (capacity + 63) & ~(uint64_t)63;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * blockDim.x) {
    const bool occ = i < capacity && table[i * (uint64_t)ew] != 0ull;
    const unsigned long long m = __ballot(occ);
    if ((threadIdx.x & 63) == 0) counts[i >> 6] = (uint32_t)__popcll(m);
  }
}

__global__ void hash_compact_kernel(const unsigned long long* table, const uint32_t* keys, uint64_t capacity, int ew, int kw,
                                    unsigned long long* out_entries, uint32_t* out_keys, const uint32_t* bases) {
  const uint64_t n_round = (capacity + 63) & ~(uint64_t)63;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * blockDim.x) {
    const bool occ = i < capacity && table[i * (uint64_t)ew] != 0ull;
    const unsigned long long m = __ballot(occ);
    if (m == 0ull) continue;
    const int lane = threadIdx.x & 63;
    if (occ) {
      const uint64_t o = (uint64_t)bases[i >> 6] + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
      for (int w = 2; w < ew; w++) out_entries[o * (uint64_t)(ew - 2) + (w - 2)] = table[i * (uint64_t)ew + w];
      for (int w = 0; w < kw; w++) out_keys[o * (uint64_t)kw + w] = keys[i * (uint64_t)kw + w];
    }
  }
}
// Finish for big tables, in two fully parallel passes (every wave-task is independent, so HBM latency is hidden by occupancy):
//   1. hash_gather_rows_kernel: one wave per 64-slot chunk (grid-stride). The occupied entries' key tuples are fetched
//      cooperatively — lane ↦ word of the chunk's tuples laid end to end, 16 loads in flight per lane — and stored as ROWS of a
//      dense row-major scratch array in slot order (one contiguous run per chunk); count and accumulators go to their columns.
//   2. group_rows_to_columns_kernel (the column pass of EVERY big Finish: it reads dense key rows, whoever wrote them — the table's
//      gather above or the run store's runs_expand_kernel): one wave per group of 64 OUTPUT rows: the group's rows (one contiguous read) are parked in
//      the wave's LDS tile [64][key_words | 1] (odd pitch: conflict-free when lanes = rows), then each column leaves as one
//      aligned 64 / 128 / 256 / 512-byte store (uint8 / uint16 / uint32 indices, 8-byte keys) plus one ballot = one 8-byte word of
//      its validity bitmap.
// History (cfg 5, 10 M groups × 32 columns, table of 33.5 M slots): one lane per slot writing 32 columns itself: 4.5 ms; one wave
// per chunk emitting its ≈19 rows per column (44 M partial-line writes): 6.2 ms; a wave walking a contiguous range of chunks with an
// LDS ring to emit aligned 64-row groups: 4.4 ms — 20 KB of LDS per wave left 6 waves per CU to hide a chain of dependent loads.
// (The arrays are separate __restrict__ kernel parameters and the pointer tables are read through the constant address space on
// purpose: with everything behind one by-value struct the compiler has to assume that the stores of one step may alias the loads
// of the next — `keys` vs `dense_keys`, the pointer tables vs the byte stores of pass 2 — and waits for every store to complete
// before the next dependent load: ≈2.5 µs per column and group, 6.6 ms per Finish for pass 2 alone.)
typedef unsigned long long* U64Ptr;
typedef unsigned char* BytePtr;
__global__ __launch_bounds__(256) void hash_gather_rows_kernel(const unsigned long long* __restrict__ table, const uint32_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ bases, uint32_t* __restrict__ dense_keys,
                                                               const FdbHashColumnsArgs a) {
  const __attribute__((address_space(4))) U64Ptr* out_vals = (const __attribute__((address_space(4))) U64Ptr*)a.out_vals;
  __shared__ uint8_t slot_of_all[4][64];
  const int ew = a.entry_words, kw = a.key_words, nv = a.n_vals;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* slot_of = slot_of_all[wave];
  const uint32_t magic = (uint32_t)((0x100000000ull + (unsigned)kw - 1ull) / (unsigned)kw);  // t / kw for t < 64 · kw
  const bool quads = (kw & 3) == 0 && ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(dense_keys)) & 15u) == 0;
  const uint32_t qmagic = (uint32_t)((0x100000000ull + ((unsigned)kw >> 2) - 1ull) / (((unsigned)kw >> 2) > 0 ? ((unsigned)kw >> 2) : 1u));
  const uint64_t n_chunks = (a.capacity + 63) >> 6;
  const uint64_t n_waves = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = (uint64_t)blockIdx.x * 4 + wave; chunk < n_chunks; chunk += n_waves) {
    const uint64_t i = chunk * 64 + lane;
    const unsigned long long* e = table + i * (uint64_t)ew;
    const bool occ = i < a.capacity && e[0] != 0ull;
    const unsigned long long m = __ballot(occ);
    if (m == 0ull) continue;
    const uint32_t n_occ = (uint32_t)__popcll(m);
    const uint64_t base = bases[chunk];
    if (occ) {
      const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      slot_of[r] = (uint8_t)lane;
      for (int v = 0; v < nv; v++) out_vals[v][base + r] = e[2 + v];
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t* src = keys + chunk * 64 * (uint64_t)kw;
    uint32_t* dst = dense_keys + base * (uint64_t)kw;
    if (quads) {
      // 16-byte pieces (tuples and dense rows are multiples of 16 bytes and 16-byte aligned): a chunk's ≈ 19 tuples of 144 bytes are
      // 171 pieces — 3 loads per lane instead of 11 four-byte ones
      const uint32_t Q = (uint32_t)kw >> 2, n_quads = n_occ * Q;
      for (uint32_t t0 = 0; t0 < n_quads; t0 += 64 * 4) {
        u32x4 val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane, tc = t < n_quads ? t : n_quads - 1u;
          const uint32_t rr = __umulhi(tc, qmagic), j = tc - rr * Q;
          val[u] = *reinterpret_cast<const u32x4*>(src + (uint32_t)slot_of[rr] * (uint32_t)kw + j * 4u);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane;
          if (t < n_quads) *reinterpret_cast<u32x4*>(dst + (size_t)t * 4) = val[u];
        }
      }
    } else {
      const uint32_t n_words = n_occ * (uint32_t)kw;
      for (uint32_t t0 = 0; t0 < n_words; t0 += 64 * 16) {
        uint32_t val[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane;
          if (t < n_words) {
            const uint32_t rr = __umulhi(t, magic), w = t - rr * (uint32_t)kw;
            val[u] = src[(uint32_t)slot_of[rr] * (uint32_t)kw + w];
          }
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane;
          if (t < n_words) dst[t] = val[u];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void group_rows_to_columns_kernel(const uint32_t* __restrict__ dense_keys, const FdbHashColumnsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const __attribute__((address_space(4))) BytePtr* out_key = (const __attribute__((address_space(4))) BytePtr*)a.out_key;
  const __attribute__((address_space(4))) U64Ptr* out_bits = (const __attribute__((address_space(4))) U64Ptr*)a.out_bits;
  __shared__ unsigned int s_nulls[64];  // NULLs per column seen by this workgroup (flushed once at the end)
  if (threadIdx.x < 64) s_nulls[threadIdx.x] = 0u;
  __syncthreads();
  const int kw = a.key_words, pitch = kw | 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * 64 * pitch;
  const uint32_t magic = (uint32_t)((0x100000000ull + (unsigned)kw - 1ull) / (unsigned)kw);
  const bool quads = (kw & 3) == 0 && (reinterpret_cast<uintptr_t>(dense_keys) & 15u) == 0;
  const uint32_t qmagic = (uint32_t)((0x100000000ull + ((unsigned)kw >> 2) - 1ull) / (((unsigned)kw >> 2) > 0 ? ((unsigned)kw >> 2) : 1u));
  const uint64_t n_groups = ((a.row_end < a.n_rows ? a.row_end : a.n_rows) + 63) >> 6;
  for (uint64_t G = (a.row_begin >> 6) + (uint64_t)blockIdx.x * 4 + wave; G < n_groups; G += (uint64_t)gridDim.x * 4) {
    const uint32_t rows = (uint32_t)(a.n_rows - G * 64 < 64 ? a.n_rows - G * 64 : 64);
    const uint32_t* src = dense_keys + G * 64 * (uint64_t)kw;
    const uint32_t n_words = rows * (uint32_t)kw;
    if (quads) {
      // 16-byte pieces (rows are multiples of 16 bytes, 16-byte aligned): 64 rows of 144 bytes are 576 pieces — 9 loads per lane, not 36
      const uint32_t Q = (uint32_t)kw >> 2, n_quads = rows * Q;
      for (uint32_t t0 = 0; t0 < n_quads; t0 += 64 * 4) {
        u32x4 val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane;
          val[u] = *reinterpret_cast<const u32x4*>(src + (size_t)(t < n_quads ? t : n_quads - 1u) * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane;
          if (t < n_quads) {
            const uint32_t rr = __umulhi(t, qmagic);
            uint32_t* d = tile + rr * pitch + (t - rr * Q) * 4u;
            d[0] = val[u].x; d[1] = val[u].y; d[2] = val[u].z; d[3] = val[u].w;
          }
        }
      }
    } else {
      for (uint32_t t0 = 0; t0 < n_words; t0 += 64 * 16) {
        uint32_t val[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane;
          if (t < n_words) val[u] = src[t];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const uint32_t t = t0 + (uint32_t)u * 64 + lane;
          if (t < n_words) { const uint32_t rr = __umulhi(t, magic); tile[rr * pitch + (t - rr * (uint32_t)kw)] = val[u]; }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const bool active = (uint32_t)lane < rows;
    const uint64_t o = G * 64 + lane;
    uint32_t* k = tile + lane * pitch;
    const unsigned long long vm = active ? (unsigned long long)k[0] | ((unsigned long long)k[1] << 32) : 0ull;
    // Key ids that leave through a table (FdbHashCol::lut: the rank of the id among the ids present) are translated IN THE TILE first,
    // eight columns' lookups in flight together: the column loop below then holds no vector LOAD at all. With one in it the compiler has
    // to wait at the head of every iteration for whatever the previous one left outstanding (a register may still be the target of a
    // load) — and on gfx950 stores count on the same counter: every column waited for the previous column's store to be acknowledged,
    // ≈ 2 µs apiece, 67 µs per 64 rows × 33 columns (0.26 ms per 2^20 rows; round 6).
    if (a.any_lut) {
      for (int c0 = 0; c0 < a.n_cols; c0 += 8) {
        uint32_t xv[8];
        int wv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int c = c0 + u < a.n_cols ? c0 + u : a.n_cols - 1;
          const FdbHashCol C = load_col(a.cols, c);
          const bool has = C.kind == 0 && C.lut != nullptr;  // (wave-uniform)
          wv[u] = has && c0 + u < a.n_cols ? C.word : -1;
          // (an unconditional load — entry 0 of the table, or of the key rows for a column without one — so that the eight of them are
          // issued back to back instead of one behind each branch)
          const uint32_t id = has && active ? k[C.word] : 0u;
          const uint32_t* L = has ? C.lut : dense_keys;
          const uint32_t v = L[id];
          xv[u] = id != 0u ? v + 1u : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (wv[u] >= 0 && active) k[wv[u]] = xv[u];
      }
    }
    for (int c = 0; c < a.n_cols; c++) {
      const FdbHashCol C = load_col(a.cols, c);
      unsigned char* out = out_key[c];
      const int width = C.src_word;  // (transport width of the column: 1, 2, 4 or 8 bytes; −2 / −4: that many BITS per index)
      // sliced columns: [slice][column][row in slice] — a slice of all narrow columns is one contiguous run for the copy engine
      const uint64_t in_slice = C.lut_len ? (o & ((1ull << a.slice_shift) - 1ull)) : o;
      const uint64_t at = (C.lut_len ? (o >> a.slice_shift) * a.slice_stride : 0ull) + (width > 0 ? in_slice * (uint64_t)width : (in_slice * (uint64_t)(-width)) >> 3);
      bool ok = false;
      if (width < 0) {  // (wave-uniform) sub-byte indices: 16 / 8 lanes fold theirs into one 32-bit word, the first of them stores it
        const uint32_t id = active ? k[C.word] : 0u;
        ok = id != 0u;
        uint32_t x = id ? id - 1u : 0u;  // (translated in the tile above when the column has a table)
        if (width == -2) {
          x |= (uint32_t)__shfl_down((int)x, 1, 64) << 2; x |= (uint32_t)__shfl_down((int)x, 2, 64) << 4;
          x |= (uint32_t)__shfl_down((int)x, 4, 64) << 8; x |= (uint32_t)__shfl_down((int)x, 8, 64) << 16;
          if ((lane & 15) == 0 && (uint32_t)lane < rows) *reinterpret_cast<uint32_t*>(out + at) = x;
        } else {
          x |= (uint32_t)__shfl_down((int)x, 1, 64) << 4; x |= (uint32_t)__shfl_down((int)x, 2, 64) << 8; x |= (uint32_t)__shfl_down((int)x, 4, 64) << 16;
          if ((lane & 7) == 0 && (uint32_t)lane < rows) *reinterpret_cast<uint32_t*>(out + at) = x;
        }
      } else if (active) {
        if (C.kind == 0) {
          const uint32_t id = k[C.word];
          const uint32_t idx = id ? id - 1u : 0u;
          ok = id != 0u;
          if (width == 1) out[at] = (uint8_t)idx;
          else if (width == 2) *reinterpret_cast<uint16_t*>(out + at) = (uint16_t)idx;
          else *reinterpret_cast<uint32_t*>(out + at) = idx;
        } else {
          ok = (vm >> C.gi) & 1ull;
          *reinterpret_cast<unsigned long long*>(out + at) = ok ? ((unsigned long long)k[C.word] | ((unsigned long long)k[C.word + 1] << 32)) : 0ull;
        }
      }
      const unsigned long long w = __ballot(ok);
      if (lane == 0) {
        out_bits[c][G] = w;
        const uint32_t nulls = rows - (uint32_t)__popcll(w);
        if (nulls != 0u) atomicAdd(&s_nulls[c], nulls);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  if ((int)threadIdx.x < a.n_cols && s_nulls[threadIdx.x] != 0u) atomicAdd(a.out_nulls + threadIdx.x, (unsigned long long)s_nulls[threadIdx.x]);
}

// See FdbPresentArgs. The dense key rows are streamed as they lie — a lane takes a 16-byte quad (4 consecutive words of one row: 4 group
// columns), 4 quads in flight — and marks its words' ids in the candidates' bitmaps in GLOBAL memory (a few hundred KB that stay in
// the L2). A wave first asks a small direct-mapped set in LDS for (candidate, id) pairs it has marked before: a result's rows repeat the
// same few ids per column, so after its first rows a wave answers nearly everything from LDS. A candidate with more than 256 ids present
// cannot be narrowed any more and is dropped by everybody (live counts in a.counts; rank_ids_kernel overwrites them with the exact ones).
// (Versions 1-4 staged 64 rows in an LDS tile and walked the columns per row: 1.0-2.0 ms per 10 M rows x 32 columns, latency-bound.)
typedef uint32_t pid_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void present_ids_kernel(const FdbPresentArgs a) {
  __shared__ uint32_t s_wset[4][512];
  __shared__ int s_cand_of_word[256];      // key-row word -> candidate (or -1)
  __shared__ unsigned int s_dead[FDB_MAX_HASH_GCOLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* wset = s_wset[wave];
  for (int i = lane; i < 512; i += 64) wset[i] = 0u;
  for (int i = threadIdx.x; i < 256; i += 256) s_cand_of_word[i] = -1;
  if (threadIdx.x < FDB_MAX_HASH_GCOLS) s_dead[threadIdx.x] = 0u;
  __syncthreads();
  if ((int)threadIdx.x < a.n_cand && a.word[threadIdx.x] < 256) s_cand_of_word[a.word[threadIdx.x]] = (int)threadIdx.x;
  __syncthreads();
  const uint32_t qpr = (uint32_t)a.key_words >> 2;  // quads per row
  const uint32_t qmagic = (uint32_t)((0x100000000ull + qpr - 1ull) / qpr);  // x / qpr = umulhi(x, qmagic) for the small x below
  const uint64_t total = a.n_rows * (uint64_t)qpr;
  const pid_u32x4* src = reinterpret_cast<const pid_u32x4*>(a.dense_keys);
  // (a wave's 4 loads cover 256 consecutive quads: load u takes quad wave_base + u * 64 + lane)
  for (uint64_t wave_base = ((uint64_t)blockIdx.x * 4 + wave) * 256; wave_base < total; wave_base += (uint64_t)gridDim.x * 4 * 256) {
    pid_u32x4 v[4];
    uint64_t qi[4];
    const uint32_t jb = (uint32_t)(wave_base % qpr);  // (wave-uniform: the quad-in-row of the wave's first quad)
#pragma unroll
    for (int u = 0; u < 4; u++) { qi[u] = wave_base + (uint64_t)u * 64 + lane; v[u] = src[qi[u] < total ? qi[u] : total - 1]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (qi[u] >= total) continue;
      const uint32_t x = jb + (uint32_t)u * 64u + (uint32_t)lane;
      const uint32_t j = x - __umulhi(x, qmagic) * qpr;
      const uint32_t ids[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const int k = s_cand_of_word[j * 4 + w];
        const uint32_t id = ids[w];
        if (k < 0 || id == 0u || s_dead[k] != 0u) continue;
        const uint32_t key = (uint32_t)k * 65537u + id + 1u;
        const uint32_t slot = (key * 2654435761u) >> 23;
        if (wset[slot] == key) continue;
        uint32_t* bw = a.bitmaps + a.bm_off[k] + (id >> 5);
        const uint32_t bit = 1u << (id & 31u);
        if (!(__hip_atomic_load(bw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
          if (!(atomicOr(bw, bit) & bit)) {  // this lane set it: one more id present
            if (atomicAdd(a.counts + k, 1ull) + 1ull > 256ull) s_dead[k] = 1u;
          }
        } else if (__hip_atomic_load(a.counts + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 256ull) {
          s_dead[k] = 1u;
        }
        wset[slot] = key;  // (lanes that collide on a slot overwrite each other with valid keys: a lost entry costs one more look at the bitmap)
      }
    }
  }
}
// One workgroup per candidate: thread t owns a contiguous share of the bitmap's words; popcounts → exclusive scan over the threads →
// every set bit gets its rank.
__global__ __launch_bounds__(256) void rank_ids_kernel(const FdbPresentArgs a) {
  __shared__ uint32_t s_sum[256];
  const int k = blockIdx.x;
  const uint32_t n_words = (a.dict_len[k] + 32u) / 32u;
  const uint32_t* bm = a.bitmaps + a.bm_off[k];
  uint32_t* remap = a.remap + a.remap_off[k];
  uint32_t* present = a.present + a.remap_off[k];
  const uint32_t per = (n_words + 255u) / 256u, w0 = threadIdx.x * per, w1 = w0 + per < n_words ? w0 + per : n_words;
  uint32_t mine = 0;
  for (uint32_t w = w0; w < w1; w++) mine += (uint32_t)__popc(bm[w]);
  s_sum[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int t = 0; t < 256; t++) { const uint32_t v = s_sum[t]; s_sum[t] = run; run += v; }
    a.counts[k] = run;
    remap[0] = 0u;
  }
  __syncthreads();
  uint32_t rank = s_sum[threadIdx.x];
  for (uint32_t w = w0; w < w1; w++) {
    uint32_t bits = bm[w];
    for (uint32_t b = 0; b < 32u; b++) {
      const uint32_t id = w * 32u + b;
      if (id == 0u || id > a.dict_len[k]) continue;
      if ((bits >> b) & 1u) { remap[id] = rank; present[rank] = id - 1u; rank++; } else remap[id] = 0u;
    }
  }
}
// Multi-workgroup exclusive scan, step 1 and 3 (step 2 = scan_counts_kernel over the per-1024 sums).
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const uint32_t* __restrict__ counts, int64_t n, uint32_t* __restrict__ sums) {
  __shared__ unsigned int wave_sum[4];
  const int64_t i0 = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) if (i0 + k < n) s += counts[i0 + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}
__global__ __launch_bounds__(256) void scan_apply_kernel(uint32_t* __restrict__ counts, int64_t n, const uint32_t* __restrict__ block_offsets) {
  __shared__ unsigned int wave_sum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { v[k] = i0 + k < n ? counts[i0 + k] : 0u; s += v[k]; }
  uint32_t incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t run = block_offsets[blockIdx.x] + incl - s;
  for (int w = 0; w < wave; w++) run += wave_sum[w];
#pragma unroll
  for (int k = 0; k < 4; k++) { if (i0 + k < n) counts[i0 + k] = run; run += v[k]; }
}

// (hash_merge / hash_partition: fdb_merge.hip)

__global__ void fill_u64_kernel(unsigned long long* dst, unsigned long long value, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = value;
}

struct FillIdents { unsigned long long v[1 + FDB_MAX_AGGS]; };
__global__ void fill_state_kernel(unsigned long long* base, int64_t n, int n_arrays, FillIdents idents) {
  const int64_t total = n * n_arrays;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) base[i] = idents.v[i / n];
}

// The first n_slots entries of every array of a small table → pinned host memory (same layout): the host copy of a table that a
// collective has just changed, without a copy command (Plan::allreduce; the fold kernel below writes its own).
__global__ void state_to_host_kernel(const unsigned long long* __restrict__ state, unsigned long long* __restrict__ host_out, uint32_t n_slots, uint64_t stride) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < n_slots) host_out[(size_t)blockIdx.y * stride + slot] = state[(size_t)blockIdx.y * stride + slot];
}

// Cross-GPU merge of a small dense table in RANK ORDER (Plan::comm_allreduce): every rank packs the first n_slots entries of its
// arrays ([array][slot], no allocation padding), ONE all-gather brings all ranks' packed tables, and every rank folds them locally —
// rank 0's value first, then rank 1's, … — into its table and its pinned host copy. The order of a float64 sum is then a function of
// the communicator alone: bit-identical whatever order the ranks arrived in and on every rank (an all-reduce's order is the
// transport's business).
__global__ void state_pack_kernel(const unsigned long long* __restrict__ state, unsigned long long* __restrict__ packed, uint32_t n_slots, uint64_t stride) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < n_slots) packed[(size_t)blockIdx.y * n_slots + slot] = state[(size_t)blockIdx.y * stride + slot];
}
__global__ void state_fold_ranks_kernel(const unsigned long long* __restrict__ gathered, int n_ranks, uint32_t n_slots, int n_arrays, FillIdents ops,
                                        unsigned long long* __restrict__ state, uint64_t stride, unsigned long long* __restrict__ host_out) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_slots) return;
  const int arr = blockIdx.y;
  const int op = (int)ops.v[arr];
  const size_t per_rank = (size_t)n_arrays * n_slots;
  unsigned long long acc;
  if (op == 0) acc = state[(size_t)arr * stride + slot];  // (an array no aggregation folds into keeps this rank's contents)
  else {
    acc = gathered[(size_t)arr * n_slots + slot];
    for (int p = 1; p < n_ranks; p++) {
      const unsigned long long v = gathered[(size_t)p * per_rank + (size_t)arr * n_slots + slot];
      if (op == 1) acc += v;
      else if (op == 2) acc = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)acc) + __longlong_as_double((long long)v));
      else if (op == 3) acc = (unsigned long long)min((long long)acc, (long long)v);
      else acc = (unsigned long long)max((long long)acc, (long long)v);
    }
    state[(size_t)arr * stride + slot] = acc;
  }
  if (host_out != nullptr) host_out[(size_t)arr * stride + slot] = acc;
}

// Folds the per-workgroup partial tables into the global table. grid = (slot tiles of 64, arrays): one workgroup of 16 waves
// covers 64 consecutive slots of one array; each wave folds a contiguous run of tables with ≈16 independent coalesced loads in
// flight, the waves combine through LDS in wave order and wave 0 updates the table with a PLAIN read-modify-write — nothing else
// touches these slots during the launch, so no atomics, and the order of a float sum is fixed by the launch geometry. The
// updated value also goes to `host_out` (pinned host memory, same layout as the table; nullptr: none): Finish then needs no copy
// command of its own — the 16 KB device→host copy of a 1 025-slot table was 15 µs on the stream, this kernel is ≈5.
// (An earlier version split the tables over grid.z = 16 workgroups of 4 waves with one atomic per slot and split.)
template <int F>
__device__ __forceinline__ unsigned long long reduce_fold(unsigned long long a, unsigned long long v) {
  if (F == 1) return a + v;
  if (F == 2) return (unsigned long long)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)v));
  if (F == 3) return (unsigned long long)min((long long)a, (long long)v);
  return (unsigned long long)max((long long)a, (long long)v);
}
template <int F>
__device__ __forceinline__ void reduce_partials_body(const unsigned long long* __restrict__ partials, const int n_blocks, const int n_arrays, const uint32_t n_slots,
                                                     unsigned long long* dst, unsigned long long* host_dst, const uint32_t slot, const int arr,
                                                     unsigned long long (*part)[64]) {
  constexpr int NW = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_wave = (n_blocks + NW - 1) / NW;
  const int b0 = wave * per_wave, b1 = min(n_blocks, b0 + per_wave);
  const unsigned long long ident = F == 3 ? (unsigned long long)FDB_I64_MAX : F == 4 ? (unsigned long long)FDB_I64_MIN : 0ull;
  unsigned long long acc = ident, t = ident;
  if (slot < n_slots) {
    if (wave == 0) t = *dst;  // (in flight together with the first tables)
    const unsigned long long* p = partials + (size_t)arr * n_slots + slot;
    const size_t stride = (size_t)n_arrays * n_slots;
#pragma unroll 32
    for (int b = b0; b < b1; b++) acc = reduce_fold<F>(acc, p[(size_t)b * stride]);
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && slot < n_slots) {
#pragma unroll
    for (int w = 0; w < NW; w++) t = reduce_fold<F>(t, part[w][lane]);
    *dst = t;
    if (host_dst != nullptr) *host_dst = t;
  }
}

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const unsigned long long* __restrict__ partials, int n_blocks, int n_arrays,
                                                               uint32_t n_slots, unsigned long long* state, uint64_t state_stride,
                                                               FillIdents funcs, unsigned long long* __restrict__ host_out) {
  __shared__ unsigned long long part[16][64];
  const int arr = blockIdx.y;
  const int f = (int)funcs.v[arr];
  const uint32_t slot = blockIdx.x * 64 + (threadIdx.x & 63);
  unsigned long long* dst = state + (size_t)arr * state_stride + slot;
  unsigned long long* host_dst = host_out != nullptr ? host_out + (size_t)arr * state_stride + slot : nullptr;
  if (f == 0) {  // (an array no aggregation folds into: the host copy still gets its contents)
    if (host_dst != nullptr && threadIdx.x < 64 && slot < n_slots) *host_dst = *dst;
    return;
  }
  if (f == 1) reduce_partials_body<1>(partials, n_blocks, n_arrays, n_slots, dst, host_dst, slot, arr, part);
  else if (f == 2) reduce_partials_body<2>(partials, n_blocks, n_arrays, n_slots, dst, host_dst, slot, arr, part);
  else if (f == 3) reduce_partials_body<3>(partials, n_blocks, n_arrays, n_slots, dst, host_dst, slot, arr, part);
  else reduce_partials_body<4>(partials, n_blocks, n_arrays, n_slots, dst, host_dst, slot, arr, part);
}

__global__ void merge_u64_kernel(unsigned long long* dst, const unsigned long long* src, const uint32_t* map, int64_t n,
                                 int32_t func, int32_t is_f64, const unsigned long long* src_cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t d = map ? map[i] : (uint32_t)i;
    if (d == 0xFFFFFFFFu) continue;
    if (src_cnt != nullptr && src_cnt[i] == 0ull) continue;  // (a slot no row reached holds identities: nothing to fold — and −0.0 + 0.0 would lose a sign)
    const unsigned long long v = src[i];
    if (func == AGG_SUM || func == AGG_COUNT) {
      if (is_f64) atomicAdd(reinterpret_cast<double*>(dst) + d, __longlong_as_double((long long)v));
      else atomicAdd(dst + d, v);
    } else if (func == AGG_MIN) {
      atomicMin(reinterpret_cast<long long*>(dst) + d, (long long)v);
    } else {
      atomicMax(reinterpret_cast<long long*>(dst) + d, (long long)v);
    }
  }
}

// ---- Parquet pages → columns (see fdb_kernels.h) -----------------------------------------------------------------------------
// 64 bits starting at an arbitrary byte of a buffer whose base is 8-byte aligned and padded by ≥ 16 readable bytes
__device__ __forceinline__ unsigned long long pq_load64(const uint8_t* __restrict__ base, uint64_t byte_off) {
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(base) + (byte_off >> 3);
  const uint32_t sh = (uint32_t)(byte_off & 7u) * 8u;
  const unsigned long long lo = w[0];
  if (sh == 0u) return lo;
  return (lo >> sh) | (w[1] << (64u - sh));
}
// index of the last run whose start is ≤ x (runs sorted by start, runs[0].start ≤ x)
__device__ __forceinline__ int pq_find_run(const FdbPqRun* __restrict__ runs, int n, int64_t x) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (runs[mid].start <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__device__ __forceinline__ uint32_t pq_run_value(const uint8_t* __restrict__ chunk, const FdbPqRun& R, int64_t x) {
  if (R.kind == 0u) return (uint32_t)R.payload;
  const uint64_t bit = R.payload + (uint64_t)(x - R.start) * R.bit_width;
  const unsigned long long wnd = pq_load64(chunk, bit >> 3) >> (bit & 7u);
  return (uint32_t)(wnd & ((R.bit_width >= 32u) ? 0xFFFFFFFFull : ((1ull << R.bit_width) - 1ull)));
}

__global__ void pq_validity_kernel(const uint8_t* __restrict__ chunk, const FdbPqRun* __restrict__ runs, int n_runs, int64_t n_rows,
                                   uint32_t* __restrict__ validity, uint32_t* __restrict__ counts) {
  const int64_t n_words = (n_rows + 31) / 32;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r0 = w * 32;
    int ri = pq_find_run(runs, n_runs, r0);
    uint32_t bits = 0;
    for (int b = 0; b < 32 && r0 + b < n_rows; b++) {
      const int64_t r = r0 + b;
      while (ri + 1 < n_runs && runs[ri + 1].start <= r) ri++;
      const FdbPqRun R = runs[ri];
      if (R.kind == 0u) {  // a repeated level: fill up to the end of the run (or of the word) at once
        const int64_t end = ri + 1 < n_runs ? runs[ri + 1].start : n_rows;
        const int take = (int)((end < r0 + 32 ? end : r0 + 32) - r);
        if (R.payload & 1ull) bits |= (take >= 32 ? 0xFFFFFFFFu : ((1u << take) - 1u)) << b;
        b += take - 1;
      } else {
        bits |= (pq_run_value(chunk, R, r) & 1u) << b;
      }
    }
    validity[w] = bits;
    counts[w] = (uint32_t)__popc(bits);
  }
}

template <int KIND>
__global__ void pq_decode_kernel(const uint8_t* __restrict__ chunk, const uint32_t* __restrict__ validity, const uint32_t* __restrict__ prefix,
                                 const FdbPqPlainPage* __restrict__ pages, int n_pages, const FdbPqRun* __restrict__ idx_runs, int n_idx_runs,
                                 int64_t n_rows, void* __restrict__ out) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    bool valid = true;
    int64_t rank = r;
    if (validity != nullptr) {
      const uint32_t word = validity[r >> 5], b = (uint32_t)(r & 31);
      valid = (word >> b) & 1u;
      rank = (int64_t)prefix[r >> 5] + (int64_t)__popc(word & ((1u << b) - 1u));
    }
    if (KIND == 0) {
      unsigned long long v = 0;
      if (valid) {
        int lo = 0, hi = n_pages - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pages[mid].rank_start <= rank) lo = mid; else hi = mid - 1; }
        v = pq_load64(chunk, (uint64_t)pages[lo].byte_off + (uint64_t)(rank - pages[lo].rank_start) * 8u);
      }
      reinterpret_cast<unsigned long long*>(out)[r] = v;
    } else {
      uint32_t v = 0;
      if (valid) { const int ri = pq_find_run(idx_runs, n_idx_runs, rank); v = pq_run_value(chunk, idx_runs[ri], rank); }
      if (KIND == 2) reinterpret_cast<unsigned long long*>(out)[r] = valid ? 1ull + (unsigned long long)(v & 1u) : 0ull;  // BOOLEAN → 1 (false) / 2 (true)
      else reinterpret_cast<uint32_t*>(out)[r] = v;
    }
  }
}

// One workgroup per page (grid-stride), 1 024 values per step: 4 consecutive values per thread.
__global__ __launch_bounds__(256) void pq_delta_kernel(const uint8_t* __restrict__ chunk, const FdbPqDeltaPage* __restrict__ pages, int n_pages,
                                                       const FdbPqDeltaMini* __restrict__ minis, unsigned long long* __restrict__ dense) {
  __shared__ unsigned long long wave_sum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int pg = blockIdx.x; pg < n_pages; pg += gridDim.x) {
    const FdbPqDeltaPage P = pages[pg];
    unsigned long long carry = 0;  // sum of everything before this step (wave-uniform)
    for (int64_t j0 = 0; j0 < P.n_values; j0 += 1024) {
      unsigned long long x[4], s = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t j = j0 + (int64_t)tid * 4 + u;
        x[u] = 0;
        if (j < P.n_values) {
          if (j == 0) x[u] = P.first_value;
          else {
            const int64_t d = j - 1;
            const FdbPqDeltaMini M = minis[P.mini_begin + (int)(d / P.vpm)];
            unsigned long long v = 0;
            if (M.width != 0u) {
              const uint64_t bit = M.bit_off + (uint64_t)(d % P.vpm) * M.width;
              const uint32_t sh = (uint32_t)(bit & 7u);
              v = pq_load64(chunk, bit >> 3) >> sh;
              if (sh + M.width > 64u) v |= (pq_load64(chunk, (bit >> 3) + 8) & 0xFFull) << (64u - sh);
              if (M.width < 64u) v &= (1ull << M.width) - 1ull;
            }
            x[u] = M.min_delta + v;
          }
        }
        s += x[u];
      }
      unsigned long long incl = s;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      __syncthreads();  // (wave_sum of the previous step has been read by everyone)
      if (lane == 63) wave_sum[wave] = incl;
      __syncthreads();
      unsigned long long run = carry + incl - s;
      for (int w = 0; w < wave; w++) run += wave_sum[w];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t j = j0 + (int64_t)tid * 4 + u;
        run += x[u];
        if (j < P.n_values) dense[P.rank_start + j] = run;
      }
      carry += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    }
    __syncthreads();
  }
}

// ---- import validation: every dictionary index of a VALID row must be below the dictionary's length ----------------------------
// (Arrow forbids anything else; the scan kernels index LUTs with these values, so a malformed record must become an error code
// at import — ≙ the reference's recovered panic, recovery/recovery.go:13-30 — not an out-of-bounds read on the device.)
__global__ void validate_indices_kernel(const uint32_t* __restrict__ idx, const uint8_t* __restrict__ validity, int64_t n, uint32_t limit,
                                        uint32_t* flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (idx[i] >= limit && (validity == nullptr || ((validity[i >> 3] >> (i & 7)) & 1))) bad = true;
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// ---- local communicator: reduce one slice of an array across the ranks' buffers (peer loads) --------------------------------
struct PeerSrcs { const unsigned long long* p[FDB_MAX_PARTS]; };
__global__ void peer_reduce_kernel(unsigned long long* dst, const PeerSrcs srcs, int n_srcs, int64_t lo, int64_t hi, int op) {
  for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long acc = srcs.p[0][i];
    for (int r = 1; r < n_srcs; r++) {
      const unsigned long long v = srcs.p[r][i];
      if (op == 1) acc += v;
      else if (op == 2) acc = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)acc) + __longlong_as_double((long long)v));
      else if (op == 3) acc = (unsigned long long)min((long long)acc, (long long)v);
      else acc = (unsigned long long)max((long long)acc, (long long)v);
    }
    dst[i] = acc;
  }
}

// ---- selection vector (≙ bitmap.ToArray() of filter.go:286, built on the device) ---------------------------
// In-place exclusive scan of per-tile counts (one workgroup; n_tiles is rows / 8192) + their total.
__global__ __launch_bounds__(FDB_BLOCK) void scan_counts_kernel(uint32_t* tile_counts, int64_t n_tiles, unsigned long long* total) {
  __shared__ unsigned long long wave_sum[FDB_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (n_tiles + FDB_BLOCK - 1) / FDB_BLOCK;
  const int64_t lo = (int64_t)tid * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
  unsigned long long s = 0;
  for (int64_t i = lo; i < hi; i++) s += tile_counts[i];
  unsigned long long incl = s;  // inclusive scan of the per-thread sums over the wave, then over the 16 wave totals
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  unsigned long long before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < FDB_BLOCK / 64; w++) { if (w < wave) before += wave_sum[w]; all += wave_sum[w]; }
  if (tid == 0) *total = all;
  unsigned long long run = before + incl - s;
  for (int64_t i = lo; i < hi; i++) { const uint32_t v = tile_counts[i]; tile_counts[i] = (uint32_t)run; run += v; }
}

// Step 1 of filter(): one mask byte per lane (8 consecutive rows), selected rows per 2 048-row tile. Grid-stride over chunks of
// 256 lanes × 8 rows; a wave's 512 rows lie inside one tile, so its count is one atomic.
__global__ __launch_bounds__(FDB_COMPACT_BLOCK) void filter_flags_kernel(const FdbScanArgs a, uint8_t* masks, uint32_t* tile_counts) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int R = 8;
  const int tid = threadIdx.x;
  for (int l = 0; l < a.n_leaves; l++) {
    const FdbLeaf& L = a.leaves[l];
    if (L.kind == FDB_LEAF_DICT_LUT && L.lut_lds != FDB_NO_LDS)
      for (uint32_t i = tid; i < L.lut_len; i += FDB_COMPACT_BLOCK) smem[L.lut_lds + i] = as_global(L.lut)[i];
  }
  __syncthreads();
  const int64_t chunk_rows = (int64_t)FDB_COMPACT_BLOCK * R;
  const int64_t n_chunks = (a.n_rows + chunk_rows - 1) / chunk_rows;
  for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    const int64_t row0 = ch * chunk_rows + (int64_t)tid * R;
    uint32_t sel = 0;
    if (row0 < a.n_rows) {
      const int64_t left = a.n_rows - row0;
      sel = eval_filter<R>(a, row0, smem) & (left >= R ? 0xFFu : ((1u << (int)left) - 1u));
      masks[row0 >> 3] = (uint8_t)sel;
    }
    uint32_t cnt = __popc(sel);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0 && cnt != 0u) atomicAdd(&tile_counts[(ch * chunk_rows + (int64_t)(tid & ~63) * R) / FDB_COMPACT_TILE], cnt);
  }
}

// Step 3 of filter(): ONE column compacted per launch (see fdb_kernels.h). One wave per 2 048-row tile, no workgroup barriers:
// the tile's first output row comes from the prefix sums of step 2, everything else is wave-local (shuffles + the wave's own LDS
// region). W = 4 / 8: bytes per value; W = 0: the selection vector itself (row numbers). A pair of sub-tiles (512 rows) at a
// time: mask nibbles → wave prefix sums → load (2 × 16 or 32 bytes per lane) → scatter into the staging buffer → coalesced
// store. The kernel is deliberately small: ≈40 VGPRs, 8 waves per SIMD — memory-level parallelism comes from the 32 waves a CU
// holds. (A version that handled all columns of a tile in one launch needed 130–250 VGPRs whichever way it was written and ran
// at 1–3 waves per SIMD: 0.27–0.34 ms per 25 M rows against 0.1 ms for this one.)
template <int W>
__global__ __launch_bounds__(FDB_COMPACT_BLOCK, 5) void compact_col_kernel(const void* __restrict__ src, const uint8_t* __restrict__ src_valid, void* __restrict__ dst,
                                                                           uint8_t* __restrict__ dst_valid, const uint8_t* __restrict__ masks,
                                                                           const uint32_t* __restrict__ tile_offsets, int64_t n_rows, unsigned long long* null_count) {
  extern __shared__ __align__(16) unsigned char smem[];
  // NS sub-tiles (of 64 lanes × 4 rows) per step: 64 bytes of value loads per lane in flight either way
  constexpr int R = 4, NW = FDB_COMPACT_BLOCK / 64, NS = W == 8 ? 2 : 4, STEP_ROWS = NS * FDB_COMPACT_SUBTILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* stage = smem + (size_t)wave * FDB_COMPACT_WAVE_LDS;  // this wave's region: ≤ 4 KiB of values, then ≤ 1 KiB of validity bytes
  uint8_t* stage_valid = stage + 4096;
  const int64_t n_tiles = (n_rows + FDB_COMPACT_TILE - 1) / FDB_COMPACT_TILE;
  uint32_t my_nulls = 0;
  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NW) {
    // the tile's 2 048 mask bits: 64 words, one per lane (bits past n_rows were never written: masked below)
    const int64_t mw = tile * (FDB_COMPACT_TILE / 32) + lane;
    const uint32_t word = mw * 32 < n_rows ? as_global(reinterpret_cast<const uint32_t*>(masks))[mw] : 0u;
    if (__ballot(word != 0u) == 0ull) continue;
    unsigned long long out = tile_offsets[tile];  // wave-uniform: next output row
#pragma unroll 1
    for (int q = 0; q < FDB_COMPACT_TILE / STEP_ROWS; q++) {
      // rows of this lane in the step: tile·2048 + q·STEP_ROWS + u·256 + lane·4 … + 3 for u < NS
      const int64_t row0 = tile * FDB_COMPACT_TILE + (int64_t)q * STEP_ROWS + (int64_t)lane * R;
      uint32_t sel[NS], any = 0;
#pragma unroll
      for (int u = 0; u < NS; u++) {
        const uint32_t w = __shfl(word, (q * NS + u) * 8 + (lane >> 3), 64);
        const int64_t left = n_rows - (row0 + (int64_t)u * FDB_COMPACT_SUBTILE);
        sel[u] = (w >> ((lane & 7) * 4)) & (left >= R ? 0xFu : left > 0 ? ((1u << (int)left) - 1u) : 0u);
        any |= sel[u];
      }
      if (__ballot(any != 0u) == 0ull) continue;
      // every load of the step is issued before anything is consumed: validity bytes and values
      uint32_t valid[NS];
      uint32_t x4[W == 4 ? NS : 1][R];
      unsigned long long x8[W == 8 ? NS : 1][R];
#pragma unroll
      for (int u = 0; u < NS; u++) {
        valid[u] = 0xFu;
        if (W != 0 && src_valid != nullptr && sel[u]) valid[u] = load_valid<R>(src_valid, row0 + (int64_t)u * FDB_COMPACT_SUBTILE);
        if (W == 4 && sel[u]) load_u32<R>(reinterpret_cast<const uint32_t*>(src) + row0 + (int64_t)u * FDB_COMPACT_SUBTILE, x4[u]);
        if (W == 8 && sel[u]) load_u64<R>(reinterpret_cast<const unsigned long long*>(src) + row0 + (int64_t)u * FDB_COMPACT_SUBTILE, x8[u]);
      }
      // Output slots without a scan: the rows before this lane's are the set bits of the four per-row ballots in lower lanes
      // (v_mbcnt: no LDS traffic, no 6-step shuffle chain), a sub-tile's total is their population count.
      uint32_t pos[NS], total = 0;
#pragma unroll
      for (int u = 0; u < NS; u++) {
        uint32_t p = total;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const unsigned long long b = __ballot((sel[u] >> r) & 1u);
          p += __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
          total += (uint32_t)__popcll(b);
        }
        pos[u] = p;
      }
#pragma unroll
      for (int u = 0; u < NS; u++) {
        if (W != 0) my_nulls += __popc(sel[u] & ~valid[u]);
        uint32_t p = pos[u];
#pragma unroll
        for (int r = 0; r < R; r++) if ((sel[u] >> r) & 1u) {
          if (W == 4) reinterpret_cast<uint32_t*>(stage)[p] = ((valid[u] >> r) & 1u) ? x4[u][r] : 0u;
          else if (W == 8) reinterpret_cast<unsigned long long*>(stage)[p] = x8[u][r];
          else reinterpret_cast<uint32_t*>(stage)[p] = (uint32_t)(row0 + (int64_t)u * FDB_COMPACT_SUBTILE + r);
          if (W != 0 && dst_valid != nullptr) stage_valid[p] = (uint8_t)((valid[u] >> r) & 1u);
          p++;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (W == 8) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(dst) + out;
#pragma unroll 2
        for (uint32_t i = lane; i < total; i += 64) d[i] = reinterpret_cast<const unsigned long long*>(stage)[i];
      } else {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst) + out;
#pragma unroll 2
        for (uint32_t i = lane; i < total; i += 64) d[i] = reinterpret_cast<const uint32_t*>(stage)[i];
      }
      if (W != 0 && dst_valid != nullptr) {
        // validity bits of the rows just written: 64 output rows per ballot, OR-ed into the (zeroed) output bitmap — the run
        // starts at an arbitrary bit, so it touches two words
        unsigned long long* bits = reinterpret_cast<unsigned long long*>(dst_valid);
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < total; i0 += 64) {
          const uint32_t i = i0 + lane;
          const unsigned long long w = __ballot(i < total && stage_valid[i] != 0);
          if (lane == 0 && w != 0ull) {
            const unsigned long long at = out + i0;
            const uint32_t sh = (uint32_t)(at & 63ull);
            atomicOr(bits + (at >> 6), w << sh);
            if (sh != 0u && (w >> (64u - sh)) != 0ull) atomicOr(bits + (at >> 6) + 1, w >> (64u - sh));
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      out += total;
    }
  }
  if (W != 0 && null_count != nullptr && src_valid != nullptr) {
    // one atomic per WORKGROUP, spread over 64 counters (the caller adds them up): thousands of waves adding to one address
    // serialise in the L2 — 8 192 atomics on one word cost 80 µs, more than the compaction itself
    __shared__ unsigned int s_nulls;
    if (tid == 0) s_nulls = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) my_nulls += __shfl_xor(my_nulls, o, 64);
    if (lane == 0 && my_nulls != 0u) atomicAdd(&s_nulls, my_nulls);
    __syncthreads();
    if (tid == 0 && s_nulls != 0u) atomicAdd(null_count + (blockIdx.x & 63), (unsigned long long)s_nulls);
  }
}

// ---- filter() over every record of a scan at once (fdb_plan_filter_batches) -------------------------------------------------------
// The selection bitmap and the per-tile counts come from a kernel GENERATED for the predicate (fdb_flags_kernel, fdb_jit.cpp: one
// launch over all records, like the aggregate scan); this is the prefix-sum step — ONE launch: workgroup b owns tiles
// [1024 b, 1024 b + 1024): its base is the sum of the earlier workgroups' block sums (accumulated by the flags kernel with one
// atomic per wave), its tiles' offsets a workgroup-wide exclusive scan. Offsets are global (over all records) modulo 2^32; the
// prefix at every record's first tile is written in full (rec_base), so a tile's place INSIDE its record's output is
// offsets[tile] − (uint32)rec_base[record], and record r's row count rec_base[r + 1] − rec_base[r].
__global__ __launch_bounds__(1024) void sel_scan_kernel(const uint32_t* __restrict__ tile_counts, const unsigned long long* __restrict__ block_sums, int64_t total_tiles,
                                                        uint32_t* __restrict__ offsets, const FdbCompactRec* __restrict__ recs, int n_recs,
                                                        unsigned long long* __restrict__ rec_base) {
  __shared__ unsigned long long wave_sum[16];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long part = 0;
  if (block_sums != nullptr) { for (int64_t i = tid; i < (int64_t)blockIdx.x; i += 1024) part += block_sums[i]; }
  else {  // (exactly blockIdx.x loads per thread, 8 in flight at a time: one after the other they were most of this kernel's 15 µs at 48 workgroups)
#pragma unroll 8
    for (int b = 0; b < (int)blockIdx.x; b++) part += as_global(tile_counts)[(int64_t)b * 1024 + tid];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += (unsigned long long)__shfl_xor((long long)part, o, 64);
  if (lane == 0) wave_sum[wave] = part;
  __syncthreads();
  if (tid == 0) { unsigned long long b = 0; for (int w = 0; w < 16; w++) b += wave_sum[w]; s_base = b; }
  __syncthreads();
  const unsigned long long base = s_base;
  const int64_t t = (int64_t)blockIdx.x * 1024 + tid;
  const uint32_t c = t < total_tiles ? tile_counts[t] : 0u;
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
  __syncthreads();  // (wave_sum is re-used)
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  unsigned long long before = base;
  for (int w = 0; w < wave; w++) before += wave_sum[w];
  const unsigned long long excl = before + incl - c;
  if (t < total_tiles) {
    offsets[t] = (uint32_t)excl;
    // is this tile the first of a record? (recs[] is sorted by tile_begin; records without rows are not in it)
    int lo = 0, hi = n_recs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (recs[mid].tile_begin <= t) lo = mid; else hi = mid - 1; }
    if (recs[lo].tile_begin == t) rec_base[lo] = excl;
    if (t == total_tiles - 1) rec_base[n_recs] = excl + c;
  }
}

__global__ __launch_bounds__(256) void sel_block_sums_kernel(const uint32_t* __restrict__ counts, int64_t n, unsigned long long* __restrict__ sums) {
  __shared__ unsigned int wave_sum[4];
  const int64_t i0 = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) if (i0 + k < n) s += counts[i0 + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = (unsigned long long)wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// Zeroes a list of byte ranges (the validity bitmaps of the output records, which the compaction ORs into) in one launch.
__global__ __launch_bounds__(256) void zero_regions_kernel(const FdbZeroRegion* __restrict__ regions, int n_regions) {
  const FdbZeroRegion R = regions[blockIdx.y];
  (void)n_regions;
  uint4* p = reinterpret_cast<uint4*>(R.ptr);
  const int64_t n16 = R.bytes / 16;  // (regions are 16-byte multiples at 16-byte aligned addresses)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- every column of every record in ONE launch, software-pipelined -----------------------------------------------------------------
// What the first multi-column kernels taught (MI355X, 100 M rows, 4 columns, 50 % selected): with loads AND stores compiled out the
// kernel still took 0.37 of its 0.83 ms — a wave's step is one long dependent chain (mask word → loads → ballots → LDS scatter →
// wave barrier → LDS read → stores) and while it walks that chain it has nothing in flight; 4 … 8 waves per SIMD do not cover it.
// So a wave now always has the NEXT step's loads in flight while it stages and stores the current one (two register sets, A / B,
// alternating — no copies between them, which would wait for the loads), and it stays on ONE column for the whole launch (the
// descriptor lives in scalar registers, the code is specialised for the column's width): waves are dealt to columns in proportion
// to their bytes (col_wave_begin[]), and inside a column a wave takes every n-th tile. Mask words and tile offsets of the tile
// after the next are prefetched too. Unselected rows are staged into dump slots instead of being branched around.
template <int W>
struct CompactUnit {
  static constexpr int G = W == 8 ? 8 : 16;  // groups of 64 rows per step: 512 × 8 or 1 024 × 4 bytes = 4 KiB of values
  uint32_t word;             // the step's tile: its selection bitmap, one 32-bit word per lane (word k = rows [32 k, 32 k + 32) of the tile; rows past the record's end cleared)
  unsigned long long vword;  // lane j < G: the RAW validity word of the step's group j (consumed a step later: nothing here waits for a load)
  uint32_t x4[W == 4 ? G : 1];
  unsigned long long x8[W == 8 ? G : 1];
  void* dst; uint8_t* dst_valid;  // (wave-uniform, like everything below)
  uint32_t out0; int q, rec;
};

// rows [64 k, 64 k + 64) of a tile as ONE wave-uniform 64-bit word out of the per-lane copy of the tile's bitmap
__device__ __forceinline__ unsigned long long compact_group_word(const uint32_t word, const int k) {
  return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)word, 2 * k) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)word, 2 * k + 1) << 32);
}
__device__ __forceinline__ unsigned long long readlane_u64(const unsigned long long v, const int l) {
  return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
}

#define FDB_CONST __attribute__((address_space(4)))  // wave-uniform tables are read with scalar loads (their own counter: a prefetched tile offset does not wait for vector loads)
template <int W, bool NULLABLE>
__device__ __forceinline__ void compact_stream(const FdbCompactRec* __restrict__ recs, const int n_recs, const FdbCompactCol* __restrict__ cols, const int n_cols, const int col,
                                               const int64_t first, const int64_t stride, const uint32_t* __restrict__ masks, const uint32_t* __restrict__ tile_offsets,
                                               const unsigned long long* __restrict__ rec_base, const int64_t total_tiles, unsigned long long* __restrict__ null_counts,
                                               unsigned char* stage, const int lane, const int spread) {
  constexpr int G = CompactUnit<W>::G, STEP_ROWS = G * 64, Q = FDB_COMPACT_TILE / STEP_ROWS;
  // staging region of the wave: STEP_ROWS values, 64 dump slots (where a lane parks rows that are not selected), then one validity
  // byte per value slot
  constexpr uint32_t DUMP = STEP_ROWS;
  uint8_t* stage_valid = stage + (size_t)(STEP_ROWS + 64) * W;
  // issue side: the record of `tile` (scalars)
  int rec = -1;
  int64_t rec_begin = 0, rec_end = 0, rec_rows = 0;
  uint32_t rec_off = 0, last_vword = 0;
  const void* src = nullptr; const uint8_t* src_valid = nullptr; void* dst = nullptr; uint8_t* dst_valid = nullptr;
  int64_t tile = first;
  int q = 0;
  if (tile >= total_tiles) return;
  const FDB_CONST uint32_t* s_offsets = (const FDB_CONST uint32_t*)tile_offsets;
  const FDB_CONST FdbCompactRec* s_recs = (const FDB_CONST FdbCompactRec*)recs;
  uint32_t word = as_global(masks)[tile * 64 + lane], toff = s_offsets[tile];
  uint32_t word_nx = 0, toff_nx = 0;
  if (tile + stride < total_tiles) { word_nx = as_global(masks)[(tile + stride) * 64 + lane]; toff_nx = s_offsets[tile + stride]; }

  auto issue = [&](CompactUnit<W>& U) {  // the loads of step (tile, q); then (tile, q) moves on
    if (tile >= rec_end) {
      int lo = rec < 0 ? 0 : rec, hi = n_recs - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_recs[mid].tile_begin <= tile) lo = mid; else hi = mid - 1; }
      rec = lo;
      rec_begin = s_recs[rec].tile_begin; rec_rows = s_recs[rec].n_rows;
      rec_end = rec + 1 < n_recs ? s_recs[rec + 1].tile_begin : total_tiles;
      rec_off = rec_base != nullptr ? (uint32_t)((const FDB_CONST unsigned long long*)rec_base)[rec] : 0u;  // (nullptr: the offsets are relative to their record already)
      last_vword = rec_rows > 0 ? (uint32_t)((rec_rows - 1) >> 6) : 0u;
      const FDB_CONST FdbCompactCol* cd = (const FDB_CONST FdbCompactCol*)cols + ((size_t)rec * n_cols + col);
      src = cd->src; src_valid = cd->src_valid; dst = cd->dst; dst_valid = cd->dst_valid;
    }
    const int64_t step_row0 = (tile - rec_begin) * FDB_COMPACT_TILE + (int64_t)q * STEP_ROWS;
    if (q == 0) {  // a new tile: bits of rows past the record's end are cleared once, in the per-lane copy (lane k: rows [32 k, 32 k + 32) of the tile)
      const int64_t lane_left = rec_rows - step_row0 - (int64_t)lane * 32;
      word &= lane_left >= 32 ? ~0u : lane_left > 0 ? (1u << (int)lane_left) - 1u : 0u;
    }
    // the step's loads are (scalar base + 32-bit lane offset); a step wholly past the record's end (the last tiles of a record's
    // last group of 4) selects nothing and reads the column's first value
    const char* step_base = reinterpret_cast<const char*>(src) + (step_row0 < rec_rows ? step_row0 * W : 0);
    // straight-line loads, one row per lane and group (lane l: row 64 j + l — a group is one contiguous 256- or 512-byte read): a lane
    // whose row is not selected reads the step's first value (one cached line for all such lanes) instead of being branched around —
    // with a branch per load the compiler waits for every load before it issues the next
#pragma unroll
    for (int j = 0; j < G; j++) {
      const bool on = __builtin_amdgcn_inverse_ballot_w64(compact_group_word(word, q * G + j));
      const uint32_t off = on ? (uint32_t)(j * 64 + lane) * (uint32_t)W : 0u;
      if (W == 4) U.x4[j] = __builtin_nontemporal_load(as_global(reinterpret_cast<const uint32_t*>(step_base + off)));
      else U.x8[j] = __builtin_nontemporal_load(as_global(reinterpret_cast<const unsigned long long*>(step_base + off)));
    }
    if (NULLABLE) {  // the step's G validity words, one per lane (bitmaps are 256-byte aligned and padded: fdb_plan.cpp kTailPad)
      uint32_t vw = (uint32_t)(step_row0 >> 6) + (lane < G ? (uint32_t)lane : 0u);
      vw = vw < last_vword ? vw : last_vword;
      U.vword = as_global(reinterpret_cast<const unsigned long long*>(src_valid))[vw];
    }
    U.word = word;
    U.dst = dst; U.dst_valid = dst_valid; U.out0 = toff - rec_off; U.q = q; U.rec = rec;
    if (++q == Q) {
      q = 0; tile += stride; word = word_nx; toff = toff_nx;
      if (tile + stride < total_tiles) { word_nx = as_global(masks)[(tile + stride) * 64 + lane]; toff_nx = s_offsets[tile + stride]; }
    }
  };

  unsigned long long out = 0;  // next output row of the tile being written (wave-uniform)
  uint32_t my_nulls = 0;
  int nulls_rec = -1;
  auto flush_nulls = [&]() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) my_nulls += __shfl_xor(my_nulls, o, 64);
    if (lane == 0 && my_nulls != 0u && nulls_rec >= 0) atomicAdd(null_counts + ((size_t)nulls_rec * n_cols + col) * 64 + spread, (unsigned long long)my_nulls);
    my_nulls = 0;
  };
  auto process = [&](const CompactUnit<W>& U) {  // positions → staging → coalesced stores
    if (U.q == 0) out = U.out0;
    if (U.rec != nulls_rec) { flush_nulls(); nulls_rec = U.rec; }
    // the selection words ARE the ballots: positions are scalar prefix sums + one mbcnt per group, ready before the loads are
    // NULLs are rare: a step in which every selected row is valid sets its run of output validity bits arithmetically, one 64-bit
    // word per lane, and stages no validity bytes at all
    bool all_valid = true;
    if (NULLABLE) {  // lane j < G looks at group j: its selection word (two lanes of the tile's bitmap) against its validity word
      const int k2 = 2 * (U.q * G + (lane & (G - 1)));
      const unsigned long long selw = (unsigned long long)(uint32_t)__shfl((int)U.word, k2, 64) | ((unsigned long long)(uint32_t)__shfl((int)U.word, k2 + 1, 64) << 32);
      const unsigned long long missing = lane < G ? selw & ~U.vword : 0ull;
      all_valid = __ballot(missing != 0ull) == 0ull;
      my_nulls += (uint32_t)__popcll(missing);
    }
    uint32_t base = 0;
    if (!NULLABLE || all_valid) {
#pragma unroll
      for (int j = 0; j < G; j++) {
        const unsigned long long sel = compact_group_word(U.word, U.q * G + j);
        const uint32_t p = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(sel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel, 0u));
        const uint32_t at = __builtin_amdgcn_inverse_ballot_w64(sel) ? p : DUMP + (uint32_t)lane;  // (no branch: rows that are not selected are parked in the lane's dump slot)
        if (W == 4) reinterpret_cast<uint32_t*>(stage)[at] = U.x4[j];
        else reinterpret_cast<unsigned long long*>(stage)[at] = U.x8[j];
        base += (uint32_t)__popcll(sel);
      }
    } else {
#pragma unroll
      for (int j = 0; j < G; j++) {
        const unsigned long long sel = compact_group_word(U.word, U.q * G + j);
        const uint32_t p = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(sel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel, 0u));
        const uint32_t at = __builtin_amdgcn_inverse_ballot_w64(sel) ? p : DUMP + (uint32_t)lane;
        const bool ok = __builtin_amdgcn_inverse_ballot_w64(readlane_u64(U.vword, j));
        if (W == 4) reinterpret_cast<uint32_t*>(stage)[at] = ok ? U.x4[j] : 0u;
        else reinterpret_cast<unsigned long long*>(stage)[at] = U.x8[j];
        stage_valid[at] = ok ? (uint8_t)1 : (uint8_t)0;
        base += (uint32_t)__popcll(sel);
      }
    }
    const uint32_t total = base;
    if (total == 0u) return;  // (wave-uniform; only dump slots were written)
    __builtin_amdgcn_wave_barrier();
    // the staged values leave 16 bytes per lane and instruction: every LDS read of the step is issued before the first store
    constexpr int CHUNKS = STEP_ROWS * W / 16 / 64;  // 4
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t chunk[CHUNKS];
#pragma unroll
    for (int k = 0; k < CHUNKS; k++) chunk[k] = reinterpret_cast<const u32x4_t*>(stage)[k * 64 + lane];
    if (W == 8) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(U.dst) + out;
#pragma unroll
      for (int k = 0; k < CHUNKS; k++) {
        const uint32_t i = (uint32_t)(k * 64 + lane) * 2u;  // first of the chunk's two values
        if (i + 1 < total) __builtin_nontemporal_store(chunk[k], reinterpret_cast<u32x4_t*>(d + i));  // (streamed out: nobody reads it back soon)
        else if (i < total) __builtin_nontemporal_store((unsigned long long)chunk[k].x | ((unsigned long long)chunk[k].y << 32), d + i);
      }
    } else {
      uint32_t* d = reinterpret_cast<uint32_t*>(U.dst) + out;
#pragma unroll
      for (int k = 0; k < CHUNKS; k++) {
        const uint32_t i = (uint32_t)(k * 64 + lane) * 4u;  // first of the chunk's four values
        if (i + 3 < total) __builtin_nontemporal_store(chunk[k], reinterpret_cast<u32x4_t*>(d + i));
        else {
          if (i < total) d[i] = chunk[k].x;
          if (i + 1 < total) d[i + 1] = chunk[k].y;
          if (i + 2 < total) d[i + 2] = chunk[k].z;
        }
      }
    }
    if (NULLABLE) {
      unsigned long long* bits = reinterpret_cast<unsigned long long*>(U.dst_valid);
      if (all_valid) {
        // bits [out, out + total) ← 1: lane k owns word (out / 64) + k of the run (≤ STEP_ROWS / 64 + 1 words)
        const unsigned long long w0 = out >> 6, lo = out, hi = out + total;  // [lo, hi)
        const unsigned long long wk = w0 + (unsigned long long)lane;
        const unsigned long long b0 = wk << 6, b1 = b0 + 64;
        if (b0 < hi && b1 > lo) {
          unsigned long long m = ~0ull;
          if (lo > b0) m &= ~0ull << (lo - b0);
          if (hi < b1) m &= ~0ull >> (b1 - hi);
          atomicOr(bits + wk, m);
        }
      } else {
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < total; i0 += 64) {
          const uint32_t i = i0 + lane;
          const unsigned long long w = __ballot(i < total && stage_valid[i] != 0);
          if (lane == 0 && w != 0ull) {
            const unsigned long long at = out + i0;
            const uint32_t sh = (uint32_t)(at & 63ull);
            atomicOr(bits + (at >> 6), w << sh);
            if (sh != 0u && (w >> (64u - sh)) != 0ull) atomicOr(bits + (at >> 6) + 1, w >> (64u - sh));
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    out += total;
  };

  CompactUnit<W> A, B;
  issue(A);
  for (;;) {
    const bool more_b = tile < total_tiles;
    if (more_b) issue(B);
    process(A);
    if (!more_b) break;
    const bool more_a = tile < total_tiles;
    if (more_a) issue(A);
    process(B);
    if (!more_a) break;
  }
  flush_nulls();
}

template <bool ANY_NULLABLE>  // (a launch without nullable columns runs the leaner code: fewer registers, one more wave per SIMD)
__global__ __launch_bounds__(FDB_COMPACT_BLOCK) void compact_multi_kernel(const FdbCompactRec* __restrict__ recs, const int n_recs, const FdbCompactCol* __restrict__ cols,
                                                                          const int n_cols, const int32_t* __restrict__ col_wave_begin, const uint32_t* __restrict__ masks,
                                                                          const uint32_t* __restrict__ tile_offsets, const unsigned long long* __restrict__ rec_base,
                                                                          const int64_t total_tiles, unsigned long long* __restrict__ null_counts) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NW = FDB_COMPACT_BLOCK / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* stage = smem + (size_t)wave * FDB_COMPACT_STREAM_WAVE_LDS;
  const int g = __builtin_amdgcn_readfirstlane((int)blockIdx.x * NW + wave);  // this wave's number; its column: col_wave_begin[col] ≤ g < col_wave_begin[col + 1]
  int col = 0;
  while (col + 1 < n_cols && g >= col_wave_begin[col + 1]) col++;
  col = __builtin_amdgcn_readfirstlane(col);
  const int begin = __builtin_amdgcn_readfirstlane(col_wave_begin[col]);
  const int first = g - begin;
  const int stride = __builtin_amdgcn_readfirstlane(col_wave_begin[col + 1]) - begin;
  if (stride <= 0 || first >= stride) return;
  // width and nullability are the same for every record of the launch (one schema; the host clears `nullable` unless EVERY record
  // carries a bitmap for the column and passes all-ones bitmaps otherwise — see Plan::filter_batches)
  const int width = __builtin_amdgcn_readfirstlane(cols[col].width), nullable = __builtin_amdgcn_readfirstlane(cols[col].nullable);
#define FDB_RUN(W, N) compact_stream<W, N>(recs, n_recs, cols, n_cols, col, first, stride, masks, tile_offsets, rec_base, total_tiles, null_counts, stage, lane, g & 63)
  if (width == 8) { if (ANY_NULLABLE && nullable) FDB_RUN(8, true); else FDB_RUN(8, false); }
  else { if (ANY_NULLABLE && nullable) FDB_RUN(4, true); else FDB_RUN(4, false); }
#undef FDB_RUN
}

int g_cu_count[16] = {0};

}  // namespace

int fdb_scan_default_grid(int device) {
  if (device < 0 || device >= 16) device = 0;
  if (g_cu_count[device] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 512;
    g_cu_count[device] = prop.multiProcessorCount;
  }
  return 2 * g_cu_count[device];
}

int fdb_slot_kernel_block(void) { return SLOT_BLOCK; }
int fdb_scan_grid(const FdbScanArgs& args, int grid_blocks, int rows_per_thread) {
  const int64_t tile_rows = rows_per_thread == 0 ? (int64_t)SLOT_BLOCK * 4 : (int64_t)FDB_BLOCK * rows_per_thread;
  const int64_t n_tiles = (args.n_rows + tile_rows - 1) / tile_rows;
  return (int)(grid_blocks > n_tiles ? n_tiles : grid_blocks);
}

hipError_t fdb_launch_reduce_partials(const unsigned long long* partials, int n_blocks, int n_arrays, uint32_t n_slots,
                                      unsigned long long* state, uint64_t state_stride, const int32_t* funcs, unsigned long long* host_out, hipStream_t stream) {
  if (n_blocks <= 0 || n_slots == 0) return hipSuccess;
  FillIdents f;
  for (int a = 0; a < 1 + FDB_MAX_AGGS; a++) f.v[a] = a < n_arrays ? (unsigned long long)funcs[a] : 0ull;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((n_slots + 63) / 64, n_arrays), dim3(1024), 0, stream, partials, n_blocks, n_arrays, n_slots, state, state_stride, f,
                     host_out);
  return hipGetLastError();
}

namespace {
typedef void (*SlotKernel)(const FdbScanArgs*, int, int64_t, FdbScanArgs);
struct SlotVariant { SlotKernel lds, nolds; int block; };
#define FDB_SV(NC4, NC8, L4, L8, BLK) {scan_slots_kernel<true, NC4, NC8, L4, L8, BLK>, scan_slots_kernel<false, NC4, NC8, L4, L8, BLK>, BLK}
// single phase: ≤ 2 four-byte + ≤ 1 eight-byte column in total, everything loaded up front
const SlotVariant kSingle[] = {FDB_SV(2, 1, 0, 0, 512), FDB_SV(2, 1, 0, 0, 256), FDB_SV(2, 1, 0, 0, 1024)};
// two phase: filter columns (≤ 4 + 2) first, then group-by / aggregate columns (≤ 2 + 3)
const SlotVariant kTwoPhase[] = {FDB_SV(FDB_MAX_C4, FDB_MAX_C8, FDB_MAX_L4, FDB_MAX_L8, 512), FDB_SV(FDB_MAX_C4, FDB_MAX_C8, FDB_MAX_L4, FDB_MAX_L8, 256),
                                  FDB_SV(FDB_MAX_C4, FDB_MAX_C8, FDB_MAX_L4, FDB_MAX_L8, 1024)};
#undef FDB_SV
// mode: 0 = default, 1 = 512 threads, 2 = 256 threads, 3 = 1024 threads
const SlotVariant& slot_variant(int two_phase, int mode) {
  const int m = (mode >= 1 && mode <= 3) ? mode - 1 : 0;
  return two_phase ? kTwoPhase[m] : kSingle[m];
}
}  // namespace

int fdb_slot_geometry(int two_phase, int mode, int lds_acc, size_t lds_bytes, int device, int* tile_rows, int* blocks_per_cu) {
  const SlotVariant& v = slot_variant(two_phase, mode);
  *tile_rows = v.block * 4;
  const void* fn = reinterpret_cast<const void*>(lds_acc ? v.lds : v.nolds);
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, v.block, lds_bytes) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 1; }
  if (occ > 2048 / v.block) occ = 2048 / v.block;  // 2048 threads per CU
  *blocks_per_cu = occ;
  return 0;
}

hipError_t fdb_launch_scan_slots(const FdbScanArgs* d_parts, int n_parts, const FdbScanArgs& common, int64_t total_tiles, int grid_blocks,
                                 size_t lds_bytes, int two_phase, int mode, hipStream_t stream) {
  if (total_tiles <= 0 || grid_blocks <= 0) return hipSuccess;
  const SlotVariant& v = slot_variant(two_phase, mode);
  SlotKernel fn = common.lds_acc ? v.lds : v.nolds;
  hipLaunchKernelGGL(fn, dim3(grid_blocks), dim3(v.block), lds_bytes, stream, d_parts, n_parts, total_tiles, common);
  return hipGetLastError();
}

hipError_t fdb_launch_scan_dense(const FdbScanArgs& args, int grid_blocks, size_t lds_bytes, int rows_per_thread, hipStream_t stream) {
  const int64_t tile_rows = (int64_t)FDB_BLOCK * rows_per_thread;
  const int64_t n_tiles = (args.n_rows + tile_rows - 1) / tile_rows;
  if (n_tiles == 0) return hipSuccess;
  if (grid_blocks > n_tiles) grid_blocks = (int)n_tiles;
  dim3 grid(grid_blocks), block(FDB_BLOCK);
  if (rows_per_thread == 8) {
    if (args.lds_acc) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_dense_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((scan_dense_kernel<8, true>), grid, block, lds_bytes, stream, args);
    } else {
      hipLaunchKernelGGL((scan_dense_kernel<8, false>), grid, block, lds_bytes, stream, args);
    }
  } else {
    if (args.lds_acc) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_dense_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((scan_dense_kernel<4, true>), grid, block, lds_bytes, stream, args);
    } else {
      hipLaunchKernelGGL((scan_dense_kernel<4, false>), grid, block, lds_bytes, stream, args);
    }
  }
  return hipGetLastError();
}

// Plain streaming read (measurement only, fdb_read_ceiling): one 4 KiB tile per workgroup, 16 bytes per lane, `nt` loads —
// the access pattern with the highest read rate found on this box (tools/bw_probe.hip).
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4* __restrict__ src, int64_t n_vec, unsigned long long* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_vec) return;
  const u32x4 v = __builtin_nontemporal_load(as_global(src + i));
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x9E3779B9u) out[0] = 1;  // keeps the load alive; never true for the zero-filled buffer
}

hipError_t fdb_launch_stream_read(const void* src, int64_t bytes, unsigned long long* out, hipStream_t stream) {
  const int64_t n_vec = bytes / 16;
  if (n_vec <= 0) return hipSuccess;
  hipLaunchKernelGGL(stream_read_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const u32x4*>(src), n_vec, out);
  return hipGetLastError();
}

namespace {
// v[i] ← the double whose order-preserving integer key v[i] holds (the inverse of f64_to_ordered: the same involution)
__global__ void ordered_to_f64_kernel(unsigned long long* v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long k = (long long)v[i];
    v[i] = (unsigned long long)(k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFFLL));
  }
}
}  // namespace

hipError_t fdb_launch_ordered_to_f64(unsigned long long* v, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ordered_to_f64_kernel, dim3(blocks), dim3(256), 0, stream, v, n);
  return hipGetLastError();
}

hipError_t fdb_launch_fill_u64(unsigned long long* dst, unsigned long long value, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_u64_kernel, dim3(blocks), dim3(256), 0, stream, dst, value, n);
  return hipGetLastError();
}

hipError_t fdb_launch_fill_state(unsigned long long* base, int64_t n, int n_arrays, const unsigned long long* idents, hipStream_t stream) {
  if (n <= 0 || n_arrays <= 0) return hipSuccess;
  FillIdents f;
  for (int a = 0; a < 1 + FDB_MAX_AGGS; a++) f.v[a] = a < n_arrays ? idents[a] : 0ull;
  int blocks = (int)((n * n_arrays + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_state_kernel, dim3(blocks), dim3(256), 0, stream, base, n, n_arrays, f);
  return hipGetLastError();
}

hipError_t fdb_launch_state_to_host(const unsigned long long* state, unsigned long long* host_out, uint32_t n_slots, uint64_t stride, int n_arrays, hipStream_t stream) {
  if (n_slots == 0 || n_arrays <= 0) return hipSuccess;
  hipLaunchKernelGGL(state_to_host_kernel, dim3((n_slots + 255) / 256, n_arrays), dim3(256), 0, stream, state, host_out, n_slots, stride);
  return hipGetLastError();
}

hipError_t fdb_launch_state_pack(const unsigned long long* state, unsigned long long* packed, uint32_t n_slots, uint64_t stride, int n_arrays, hipStream_t stream) {
  if (n_slots == 0 || n_arrays <= 0) return hipSuccess;
  hipLaunchKernelGGL(state_pack_kernel, dim3((n_slots + 255) / 256, n_arrays), dim3(256), 0, stream, state, packed, n_slots, stride);
  return hipGetLastError();
}
hipError_t fdb_launch_state_fold_ranks(const unsigned long long* gathered, int n_ranks, uint32_t n_slots, int n_arrays, const int32_t* ops, unsigned long long* state,
                                       uint64_t stride, unsigned long long* host_out, hipStream_t stream) {
  if (n_slots == 0 || n_arrays <= 0) return hipSuccess;
  FillIdents f;
  for (int a = 0; a < 1 + FDB_MAX_AGGS; a++) f.v[a] = a < n_arrays ? (unsigned long long)ops[a] : 0ull;
  hipLaunchKernelGGL(state_fold_ranks_kernel, dim3((n_slots + 255) / 256, n_arrays), dim3(256), 0, stream, gathered, n_ranks, n_slots, n_arrays, f, state, stride, host_out);
  return hipGetLastError();
}

hipError_t fdb_launch_merge_u64(unsigned long long* dst, const unsigned long long* src, const uint32_t* map, int64_t n, int32_t func,
                                int32_t is_f64, hipStream_t stream, const unsigned long long* src_cnt) {
  if (n <= 0) return hipSuccess;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(merge_u64_kernel, dim3(blocks), dim3(256), 0, stream, dst, src, map, n, func, is_f64, src_cnt);
  return hipGetLastError();
}

// ---- Snappy pages → bytes (see fdb_kernels.h) ---------------------------------------------------------------------------------------
namespace {
constexpr uint32_t SNAPPY_WIN = 2048;    // bytes of the compressed stream held in LDS
constexpr uint32_t SNAPPY_RING = 65536;  // the page's most recent output, in LDS: what copies read (a compressor's matches stay inside its 64 KiB fragment)
constexpr uint32_t SNAPPY_SEG = 16384;   // the ring goes to HBM a segment at a time, 16 bytes per lane
struct __attribute__((packed, aligned(1))) SnappyChunk { unsigned long long a, b; };  // 16 bytes at any address
typedef uint32_t snappy_u32x4 __attribute__((ext_vector_type(4)));

// First version: copies read the page's output back from HBM — correct, and 0.56 µs per ELEMENT (a dependent global round trip each):
// 11 MB/s per page of dictionary indices. Elements now touch LDS only: tags come out of the input window, copies read and write the
// output ring, and the ring leaves for HBM in 16 KiB segments.
__global__ __launch_bounds__(64) void snappy_decode_kernel(const uint8_t* __restrict__ src, const FdbSnappyPage* __restrict__ pages, const int n_pages,
                                                           uint8_t* __restrict__ dst, uint32_t* __restrict__ status) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint8_t* const win = smem;                      // [SNAPPY_WIN + 16]
  uint8_t* const ring = smem + SNAPPY_WIN + 16;   // [SNAPPY_RING]
  const uint32_t lane = threadIdx.x;
  for (int pg = blockIdx.x; pg < n_pages; pg += gridDim.x) {
    const FdbSnappyPage P = pages[pg];
    const uint8_t* in = src + P.src_off;
    uint8_t* out = dst + P.dst_off;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.src_len), cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.dst_len);
    uint32_t base = 0xFFFFFFFFu, ip = 0, op = 0, flushed = 0, err = 0;
    // bytes [at, at + 8) of the stream through the window (zeros past the end: the bounds are checked on ip, not here)
    auto fetch = [&](const uint32_t at) -> unsigned long long {
      if (base == 0xFFFFFFFFu || at < base || at + 8u > base + SNAPPY_WIN) {
        __builtin_amdgcn_wave_barrier();
        base = at;
        const uint32_t i = lane * 32u;  // 64 lanes × 32 bytes = the window
        SnappyChunk c0 = {0ull, 0ull}, c1 = {0ull, 0ull};
        if ((unsigned long long)base + i + 32u <= n) { c0 = *reinterpret_cast<const SnappyChunk*>(in + base + i); c1 = *reinterpret_cast<const SnappyChunk*>(in + base + i + 16); }
        else {
          uint8_t t[32];
          for (uint32_t k = 0; k < 32u; k++) t[k] = (unsigned long long)base + i + k < n ? in[base + i + k] : (uint8_t)0;
          __builtin_memcpy(&c0, t, 16); __builtin_memcpy(&c1, t + 16, 16);
        }
        *reinterpret_cast<SnappyChunk*>(win + i) = c0; *reinterpret_cast<SnappyChunk*>(win + i + 16) = c1;
        __builtin_amdgcn_wave_barrier();
      }
      // 8 bytes at any offset out of three aligned words (the window has a word of slack behind it)
      const uint32_t o = at - base, sh = (o & 3u) * 8u;
      const uint32_t* w32 = reinterpret_cast<const uint32_t*>(win + (o & ~3u));
      const uint32_t w0 = w32[0], w1 = w32[1], w2 = w32[2];
      unsigned long long v = ((unsigned long long)w1 << 32) | w0;
      if (sh != 0u) v = (v >> sh) | ((unsigned long long)w2 << (64u - sh));
      // every lane read the same bytes: say so, and the tag arithmetic and the branches on it run on the scalar unit
      return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
    };
    // ring → HBM: every segment that is complete (all == false), or everything up to op (the page's end)
    auto flush = [&](const bool all) {
      __builtin_amdgcn_wave_barrier();
      while (flushed + SNAPPY_SEG <= op) {
        const uint32_t r0 = flushed & (SNAPPY_RING - 1u);
        for (uint32_t i = lane * 16u; i < SNAPPY_SEG; i += 64u * 16u) {
          const snappy_u32x4 v = *reinterpret_cast<const snappy_u32x4*>(ring + r0 + i);
          SnappyChunk c; __builtin_memcpy(&c, &v, 16);
          *reinterpret_cast<SnappyChunk*>(out + flushed + i) = c;
        }
        flushed += SNAPPY_SEG;
      }
      if (all) { for (uint32_t i = flushed + lane; i < op; i += 64u) out[i] = ring[i & (SNAPPY_RING - 1u)]; flushed = op; }
      __builtin_amdgcn_wave_barrier();
    };
    // preamble: the uncompressed length as a varint
    {
      unsigned long long len = 0;
      const unsigned long long w = fetch(0);
      int shift = 0, k = 0;
      for (;; k++, shift += 7) {
        if ((uint32_t)k >= n || k >= 5) { err = 1; break; }
        const uint32_t b = (uint32_t)(w >> (8 * k)) & 0xFFu;
        len |= (unsigned long long)(b & 0x7Fu) << shift;
        if (!(b & 0x80u)) { k++; break; }
      }
      ip = (uint32_t)k;
      if (!err && len != (unsigned long long)cap) err = 1;
    }
    while (err == 0 && ip < n) {
      const unsigned long long w = fetch(ip);
      const uint32_t tag = (uint32_t)w & 0xFFu;
      if ((tag & 3u) == 0u) {  // literal
        uint32_t l = (tag >> 2) + 1u, hdr = 1u;
        if (l > 60u) {
          const uint32_t extra = l - 60u;  // 1 … 4 length bytes
          if (ip + 1u + extra > n) { err = 2; break; }
          l = (uint32_t)((w >> 8) & (extra == 4u ? 0xFFFFFFFFull : ((1ull << (8u * extra)) - 1ull))) + 1u;
          hdr = 1u + extra;
          if (l == 0u) { err = 2; break; }  // (2^32: more than a page can hold)
        }
        if ((unsigned long long)ip + hdr + l > n) { err = 2; break; }
        if ((unsigned long long)op + l > cap) { err = 3; break; }
        const uint32_t from = ip + hdr;
        if (from >= base && from + l <= base + SNAPPY_WIN) {  // short and already in the window
          for (uint32_t i = lane; i < l; i += 64u) ring[(op + i) & (SNAPPY_RING - 1u)] = win[from - base + i];
          op += l;
        } else {  // from the stream in HBM, at most a segment at a time (the ring must not lap what has not been flushed)
          uint32_t done = 0;
          while (done < l) {
            const uint32_t room = SNAPPY_RING - (op - flushed), take = l - done < room ? l - done : room;
            const uint32_t body = take & ~15u;
            for (uint32_t i = lane * 16u; i < body; i += 64u * 16u) {
              const SnappyChunk c = *reinterpret_cast<const SnappyChunk*>(in + from + done + i);
              uint8_t t[16]; __builtin_memcpy(t, &c, 16);
              const uint32_t r = (op + i) & (SNAPPY_RING - 1u);
              if ((r & 15u) == 0u) { snappy_u32x4 v; __builtin_memcpy(&v, t, 16); *reinterpret_cast<snappy_u32x4*>(ring + r) = v; }
              else { for (uint32_t k = 0; k < 16u; k++) ring[(r + k) & (SNAPPY_RING - 1u)] = t[k]; }
            }
            for (uint32_t i = body + lane; i < take; i += 64u) ring[(op + i) & (SNAPPY_RING - 1u)] = in[from + done + i];
            op += take; done += take;
            if (op - flushed >= SNAPPY_SEG) flush(false);
          }
        }
        ip += hdr + l;
        if (op - flushed >= 2u * SNAPPY_SEG) flush(false);
        continue;
      }
      uint32_t l, off, hdr;
      if ((tag & 3u) == 1u) { hdr = 2u; l = 4u + ((tag >> 2) & 7u); off = ((tag >> 5) << 8) | ((uint32_t)(w >> 8) & 0xFFu); }
      else if ((tag & 3u) == 2u) { hdr = 3u; l = (tag >> 2) + 1u; off = (uint32_t)(w >> 8) & 0xFFFFu; }
      else { hdr = 5u; l = (tag >> 2) + 1u; off = (uint32_t)(w >> 8); }
      if (ip + hdr > n) { err = 2; break; }
      if (off == 0u || off > op) { err = 4; break; }
      if (off > SNAPPY_RING - 64u) { err = 6; break; }  // further back than the ring remembers (no compressor emits it)
      if ((unsigned long long)op + l > cap) { err = 3; break; }
      {  // l ≤ 64: one byte per lane; a pattern shorter than the copy repeats (every source byte lies before op). off and l are
         // wave-uniform: the three cases are BRANCHES (as one select the division of the rare case was paid by every element)
        const uint32_t i = lane;
        uint32_t at = i;
        if (off < l) {
          if ((off & (off - 1u)) == 0u) at = i & (off - 1u);
          else at = i % off;
        }
        uint8_t v = 0;
        if (i < l) v = ring[(op - off + at) & (SNAPPY_RING - 1u)];
        __builtin_amdgcn_wave_barrier();
        if (i < l) ring[(op + i) & (SNAPPY_RING - 1u)] = v;
      }
      ip += hdr; op += l;
      if (op - flushed >= 2u * SNAPPY_SEG) flush(false);
    }
    if (err == 0 && op != cap) err = 5;
    if (err == 0) flush(true);
    if (lane == 0) status[pg] = err;
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace

hipError_t fdb_launch_snappy_decode(const uint8_t* src, const FdbSnappyPage* pages, int32_t n_pages, uint8_t* dst, uint32_t* status, hipStream_t stream) {
  if (n_pages <= 0) return hipSuccess;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(snappy_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SNAPPY_WIN + 16 + SNAPPY_RING)); attr_set = true; }
  hipLaunchKernelGGL(snappy_decode_kernel, dim3((unsigned)std::min<int32_t>(n_pages, 8192)), dim3(64), SNAPPY_WIN + 16 + SNAPPY_RING, stream, src, pages, (int)n_pages, dst, status);
  return hipGetLastError();
}

hipError_t fdb_launch_pq_validity(const uint8_t* chunk, const FdbPqRun* def_runs, int32_t n_runs, int64_t n_rows, uint32_t* validity, uint32_t* counts,
                                  hipStream_t stream) {
  const int64_t n_words = (n_rows + 31) / 32;
  if (n_words == 0) return hipSuccess;
  int64_t blocks = (n_words + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pq_validity_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, chunk, def_runs, n_runs, n_rows, validity, counts);
  return hipGetLastError();
}

hipError_t fdb_launch_pq_decode(int kind, const uint8_t* chunk, const uint32_t* validity, const uint32_t* prefix, const FdbPqPlainPage* pages, int32_t n_pages,
                                const FdbPqRun* idx_runs, int32_t n_idx_runs, int64_t n_rows, void* out, hipStream_t stream) {
  if (n_rows == 0) return hipSuccess;
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (kind == 0) hipLaunchKernelGGL(pq_decode_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, chunk, validity, prefix, pages, n_pages, idx_runs, n_idx_runs, n_rows, out);
  else if (kind == 2) hipLaunchKernelGGL(pq_decode_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, chunk, validity, prefix, pages, n_pages, idx_runs, n_idx_runs, n_rows, out);
  else hipLaunchKernelGGL(pq_decode_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, chunk, validity, prefix, pages, n_pages, idx_runs, n_idx_runs, n_rows, out);
  return hipGetLastError();
}

hipError_t fdb_launch_pq_delta(const uint8_t* chunk, const FdbPqDeltaPage* pages, int32_t n_pages, const FdbPqDeltaMini* minis, unsigned long long* dense,
                               hipStream_t stream) {
  if (n_pages <= 0) return hipSuccess;
  hipLaunchKernelGGL(pq_delta_kernel, dim3((unsigned)std::min(n_pages, 8192)), dim3(256), 0, stream, chunk, pages, n_pages, minis, dense);
  return hipGetLastError();
}

hipError_t fdb_launch_validate_indices(const uint32_t* idx, const uint8_t* validity, int64_t n, uint32_t limit, uint32_t* flag, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(validate_indices_kernel, dim3(blocks), dim3(256), 0, stream, idx, validity, n, limit, flag);
  return hipGetLastError();
}

hipError_t fdb_launch_peer_reduce(unsigned long long* dst, const void* const* srcs, int n_srcs, int64_t lo, int64_t hi, int op, hipStream_t stream) {
  if (hi <= lo) return hipSuccess;
  if (n_srcs < 1 || n_srcs > FDB_MAX_PARTS) return hipErrorInvalidValue;
  PeerSrcs S;
  for (int r = 0; r < n_srcs; r++) S.p[r] = (const unsigned long long*)srcs[r];
  int blocks = (int)((hi - lo + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(peer_reduce_kernel, dim3(blocks), dim3(256), 0, stream, dst, S, n_srcs, lo, hi, op);
  return hipGetLastError();
}

hipError_t fdb_launch_filter_flags(const FdbScanArgs& args, uint8_t* masks, uint32_t* tile_counts, int device, hipStream_t stream) {
  const int64_t chunk_rows = (int64_t)FDB_COMPACT_BLOCK * 8;
  const int64_t n_chunks = (args.n_rows + chunk_rows - 1) / chunk_rows;
  if (n_chunks == 0) return hipSuccess;
  int64_t grid = (int64_t)fdb_scan_default_grid(device) * 4;  // 8 workgroups (32 waves) per CU: latency hiding by occupancy
  if (grid > n_chunks) grid = n_chunks;
  hipLaunchKernelGGL(filter_flags_kernel, dim3((unsigned)grid), dim3(FDB_COMPACT_BLOCK), args.lds_lut_bytes, stream, args, masks, tile_counts);
  return hipGetLastError();
}


hipError_t fdb_launch_exclusive_scan(uint32_t* counts, int64_t n, uint32_t* block_sums, unsigned long long* total, hipStream_t stream) {
  if (n <= 0) return hipMemsetAsync(total, 0, 8, stream);
  if (n <= 4096 || block_sums == nullptr) {  // small: one workgroup
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(FDB_BLOCK), 0, stream, counts, n, total);
    return hipGetLastError();
  }
  const int64_t n_blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, counts, n, block_sums);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(FDB_BLOCK), 0, stream, block_sums, n_blocks, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, counts, n, block_sums);
  return hipGetLastError();
}

hipError_t fdb_launch_compact_col(int width, const void* src, const uint8_t* src_valid, void* dst, uint8_t* dst_valid, const uint8_t* masks,
                                  const uint32_t* tile_offsets, int64_t n_rows, unsigned long long* null_count, int device, hipStream_t stream) {
  const int64_t n_tiles = (n_rows + FDB_COMPACT_TILE - 1) / FDB_COMPACT_TILE;
  if (n_tiles == 0) return hipSuccess;
  // 5 workgroups (20 independent waves) per CU — 5 KiB of staging LDS per wave
  int64_t grid = (int64_t)fdb_scan_default_grid(device) * 4;
  const int64_t need = (n_tiles + 3) / 4;
  if (grid > need) grid = need;
  const size_t lds = (size_t)(FDB_COMPACT_BLOCK / 64) * FDB_COMPACT_WAVE_LDS;
  if (width == 4) hipLaunchKernelGGL(compact_col_kernel<4>, dim3((unsigned)grid), dim3(FDB_COMPACT_BLOCK), lds, stream, src, src_valid, dst, dst_valid, masks, tile_offsets, n_rows, null_count);
  else if (width == 8) hipLaunchKernelGGL(compact_col_kernel<8>, dim3((unsigned)grid), dim3(FDB_COMPACT_BLOCK), lds, stream, src, src_valid, dst, dst_valid, masks, tile_offsets, n_rows, null_count);
  else if (width == 0) hipLaunchKernelGGL(compact_col_kernel<0>, dim3((unsigned)grid), dim3(FDB_COMPACT_BLOCK), lds, stream, src, src_valid, dst, dst_valid, masks, tile_offsets, n_rows, null_count);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}


hipError_t fdb_launch_sel_scan(const uint32_t* tile_counts, const unsigned long long* block_sums, int64_t total_tiles, uint32_t* offsets, const FdbCompactRec* recs,
                               int n_recs, unsigned long long* rec_base, hipStream_t stream) {
  if (total_tiles <= 0 || n_recs <= 0) return hipSuccess;
  const int64_t n_blocks = (total_tiles + 1023) / 1024;
  hipLaunchKernelGGL(sel_scan_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, stream, tile_counts, block_sums, total_tiles, offsets, recs, n_recs, rec_base);
  return hipGetLastError();
}

hipError_t fdb_launch_sel_block_sums(const uint32_t* tile_counts, int64_t total_tiles, unsigned long long* block_sums, hipStream_t stream) {
  if (total_tiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(sel_block_sums_kernel, dim3((unsigned)((total_tiles + 1023) / 1024)), dim3(256), 0, stream, tile_counts, total_tiles, block_sums);
  return hipGetLastError();
}

hipError_t fdb_launch_zero_regions(const FdbZeroRegion* regions, int n_regions, int64_t max_bytes, hipStream_t stream) {
  if (n_regions <= 0) return hipSuccess;
  int64_t gx = (max_bytes / 16 + 255) / 256;
  gx = gx < 1 ? 1 : gx > 64 ? 64 : gx;
  hipLaunchKernelGGL(zero_regions_kernel, dim3((unsigned)gx, (unsigned)n_regions), dim3(256), 0, stream, regions, n_regions);
  return hipGetLastError();
}

int fdb_compact_multi_blocks_per_cu(int any_nullable) {  // workgroups of compact_multi_kernel resident on one CU (registers and LDS): a wave's share of tiles is fixed at
  static const int n[2] = {[] {                          // launch, so the launch must not be larger than what runs at once
    int b = 0;
    const size_t lds = (size_t)(FDB_COMPACT_BLOCK / 64) * FDB_COMPACT_STREAM_WAVE_LDS;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, compact_multi_kernel<false>, FDB_COMPACT_BLOCK, lds) != hipSuccess || b < 1) { (void)hipGetLastError(); b = 4; }
    return b;
  }(), [] {
    int b = 0;
    const size_t lds = (size_t)(FDB_COMPACT_BLOCK / 64) * FDB_COMPACT_STREAM_WAVE_LDS;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, compact_multi_kernel<true>, FDB_COMPACT_BLOCK, lds) != hipSuccess || b < 1) { (void)hipGetLastError(); b = 4; }
    return b;
  }()};
  return n[any_nullable ? 1 : 0];
}

hipError_t fdb_launch_compact_multi(const FdbCompactRec* recs, int n_recs, const FdbCompactCol* cols, int n_cols, int any_nullable, const int32_t* col_wave_begin, int n_waves,
                                    const uint32_t* masks, const uint32_t* tile_offsets, const unsigned long long* rec_base, int64_t total_tiles,
                                    unsigned long long* null_counts, hipStream_t stream) {
  if (total_tiles <= 0 || n_cols <= 0 || n_recs <= 0 || n_waves <= 0) return hipSuccess;
  const size_t lds = (size_t)(FDB_COMPACT_BLOCK / 64) * FDB_COMPACT_STREAM_WAVE_LDS;
  hipLaunchKernelGGL(any_nullable ? compact_multi_kernel<true> : compact_multi_kernel<false>, dim3((unsigned)((n_waves + 3) / 4)), dim3(FDB_COMPACT_BLOCK), lds, stream, recs, n_recs,
                     cols, n_cols, col_wave_begin, masks, tile_offsets, rec_base, total_tiles, null_counts);
  return hipGetLastError();
}

hipError_t fdb_launch_scan_hash(const FdbHashArgs& args, int grid_blocks, size_t lds_bytes, hipStream_t stream) {
  const int64_t tile_rows = (int64_t)FDB_HASH_BLOCK;
  const int64_t n_tiles = (args.row_end - args.row_begin + tile_rows - 1) / tile_rows;
  if (n_tiles <= 0) return hipSuccess;
  if (grid_blocks > n_tiles) grid_blocks = (int)n_tiles;
  lds_bytes = (size_t)args.base.lds_lut_bytes + (size_t)args.key_words * FDB_HASH_BLOCK * 4;  // LUT copies + key staging
  if (lds_bytes > 48 * 1024)  // (the kernel also has 4 bytes of static LDS: stay below the 160 KiB total)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_hash_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void)hipGetLastError();
  hipLaunchKernelGGL(scan_hash_kernel, dim3(grid_blocks), dim3(FDB_HASH_BLOCK), lds_bytes, stream, args);
  return hipGetLastError();
}

hipError_t fdb_launch_hash_init(unsigned long long* table, uint64_t capacity, int entry_words, int n_aggs, const unsigned long long* idents,
                                hipStream_t stream) {
  HashIdents id;
  for (int j = 0; j < FDB_MAX_AGGS; j++) id.v[j] = j < n_aggs ? idents[j] : 0ull;
  const uint64_t total = capacity * (uint64_t)entry_words;
  hipLaunchKernelGGL(hash_init_kernel, dim3(4096), dim3(256), 0, stream, table, total, entry_words, n_aggs, id);
  return hipGetLastError();
}

hipError_t fdb_launch_hash_rehash(const unsigned long long* old_table, const uint32_t* old_keys, uint64_t old_capacity, int old_key_words, int old_used_words,
                                  unsigned long long* new_table, uint32_t* new_keys, uint64_t new_mask, int entry_words, int new_key_words,
                                  hipStream_t stream) {
  hipLaunchKernelGGL(hash_rehash_kernel, dim3(4096), dim3(256), 0, stream, old_table, old_keys, old_capacity, old_key_words, old_used_words, new_table, new_keys,
                     new_mask, entry_words, new_key_words);
  return hipGetLastError();
}

hipError_t fdb_launch_hash_chunk_bases(const unsigned long long* table, uint64_t capacity, int entry_words, uint32_t* bases, uint32_t* block_sums,
                                       unsigned long long* n_out, hipStream_t stream) {
  const int64_t n_chunks = (int64_t)((capacity + 63) / 64);
  if (n_chunks == 0) return hipMemsetAsync(n_out, 0, 8, stream);
  hipLaunchKernelGGL(hash_chunk_counts_kernel, dim3(4096), dim3(256), 0, stream, table, capacity, entry_words, bases);
  return fdb_launch_exclusive_scan(bases, n_chunks, block_sums, n_out, stream);  // in place: counts → exclusive prefix sums
}

hipError_t fdb_launch_hash_compact(const unsigned long long* table, const uint32_t* keys, uint64_t capacity, int entry_words, int key_words,
                                   unsigned long long* out_entries, uint32_t* out_keys, const uint32_t* bases, hipStream_t stream) {
  hipLaunchKernelGGL(hash_compact_kernel, dim3(4096), dim3(256), 0, stream, table, keys, capacity, entry_words, key_words, out_entries, out_keys,
                     bases);
  return hipGetLastError();
}

hipError_t fdb_launch_present_ids(const FdbPresentArgs& args, int device, hipStream_t stream) {
  if (args.n_rows == 0 || args.n_cand <= 0) return hipSuccess;
  if (args.key_words % 4 != 0 || args.key_words > 256 || ((uintptr_t)args.dense_keys & 15u) != 0) return hipErrorInvalidValue;
  const int64_t cus = fdb_scan_default_grid(device) / 2;
  const uint64_t total = args.n_rows * (uint64_t)(args.key_words / 4);
  const int64_t blocks = (int64_t)((total + 1023) / 1024);
  hipLaunchKernelGGL(present_ids_kernel, dim3((unsigned)std::min<int64_t>(blocks, cus * 8)), dim3(256), 0, stream, args);
  return hipGetLastError();
}
hipError_t fdb_launch_rank_ids(const FdbPresentArgs& args, hipStream_t stream) {
  if (args.n_cand <= 0) return hipSuccess;
  hipLaunchKernelGGL(rank_ids_kernel, dim3((unsigned)args.n_cand), dim3(256), 0, stream, args);
  return hipGetLastError();
}

hipError_t fdb_launch_hash_gather_rows(const FdbHashColumnsArgs& args, int device, hipStream_t stream) {
  const int64_t n_chunks = (int64_t)((args.capacity + 63) / 64);
  if (n_chunks == 0 || args.n_rows == 0) return hipSuccess;
  const int64_t cus = fdb_scan_default_grid(device) / 2;
  hipLaunchKernelGGL(hash_gather_rows_kernel, dim3((unsigned)std::min<int64_t>((n_chunks + 3) / 4, cus * 16)), dim3(256), 0, stream, args.table, args.keys, args.bases, args.dense_keys, args);
  return hipGetLastError();
}

hipError_t fdb_launch_hash_rows_to_columns(const FdbHashColumnsArgs& args, int device, hipStream_t stream) {
  const uint64_t end = std::min<uint64_t>(args.row_end, args.n_rows);
  if (args.row_begin >= end) return hipSuccess;
  if ((args.row_begin & 63u) != 0u) return hipErrorInvalidValue;
  const size_t lds = (size_t)4 * 64 * (size_t)(args.key_words | 1) * 4;  // one tile per wave
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&group_rows_to_columns_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
  }
  const int64_t cus = fdb_scan_default_grid(device) / 2;
  const int64_t n_groups = (int64_t)((end - args.row_begin + 63) / 64);
  hipLaunchKernelGGL(group_rows_to_columns_kernel, dim3((unsigned)std::min<int64_t>((n_groups + 3) / 4, cus * 16)), dim3(256), lds, stream, (const uint32_t*)args.dense_keys, args);
  return hipGetLastError();
}


// ---- Finish of a run store (table-free OrderedAggregate; fdb_kernels.h "run store") ------------------------------------------------
// Exclusive prefix sums of a (strided) u32 array in three launches: sums of 1 024-element blocks, a one-workgroup scan of those,
// the blocks again with their base. (A one-workgroup scan of everything is the trap DESIGN §9 names.)
__global__ __launch_bounds__(256) void scan_u32_sums_kernel(const uint32_t* __restrict__ in, int stride, int64_t n, unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long s_w[4];
  const int64_t b = blockIdx.x;
  unsigned long long v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int64_t i = b * 1024 + (int64_t)k * 256 + threadIdx.x; if (i < n) v += in[i * stride]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) sums[b] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void scan_u64_single_kernel(unsigned long long* __restrict__ sums, int64_t n_blocks, unsigned long long* __restrict__ total) {
  __shared__ unsigned long long s_w[16];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
    const int64_t i = b0 + threadIdx.x;
    const unsigned long long x = i < n_blocks ? sums[i] : 0ull;
    unsigned long long incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += y; }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long wbase = s_carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += s_w[w];
    if (i < n_blocks) sums[i] = wbase + incl - x;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}
__global__ __launch_bounds__(256) void scan_u32_apply_kernel(const uint32_t* __restrict__ in, int stride, int64_t n, const unsigned long long* __restrict__ sums, uint32_t* __restrict__ out) {
  __shared__ uint32_t s_w[4];
  const int64_t b = blockIdx.x;
  uint32_t x[4], mine = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int64_t i = b * 1024 + (int64_t)threadIdx.x * 4 + k; x[k] = i < n ? in[i * stride] : 0u; mine += x[k]; }
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += y; }
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t ex = (uint32_t)sums[b] + incl - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) ex += s_w[w];
#pragma unroll
  for (int k = 0; k < 4; k++) { const int64_t i = b * 1024 + (int64_t)threadIdx.x * 4 + k; if (i < n) out[i] = ex; ex += x[k]; }
}
hipError_t fdb_launch_scan_u32(const uint32_t* in, int stride, uint32_t* out, int64_t n, unsigned long long* scratch, unsigned long long* total, hipStream_t stream) {
  if (n <= 0) return hipMemsetAsync(total, 0, 8, stream);
  const int64_t n_blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(scan_u32_sums_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, in, stride, n, scratch);
  hipLaunchKernelGGL(scan_u64_single_kernel, dim3(1), dim3(1024), 0, stream, scratch, n_blocks, total);
  hipLaunchKernelGGL(scan_u32_apply_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, in, stride, n, (const unsigned long long*)scratch, out);
  return hipGetLastError();
}

// logical run → (segment, index): a workgroup takes 256 directory entries, lays their runs end to end and lets every thread find
// its run's entry by bisection in LDS (the writes are coalesced; an entry holds ≤ 256 runs)
__global__ __launch_bounds__(256) void runs_map_kernel(const uint32_t* __restrict__ dir, const uint32_t* __restrict__ starts, int64_t n_entries, const FdbRunSegs segs,
                                                       unsigned long long* __restrict__ phys) {
  __shared__ uint32_t s_start[257], s_base[256];
  const int64_t e0 = (int64_t)blockIdx.x * 256;
  const int64_t e = e0 + threadIdx.x;
  const uint32_t n_here = e < n_entries ? dir[e * 2 + 1] : 0u;
  s_start[threadIdx.x] = e < n_entries ? starts[e] : 0u;
  s_base[threadIdx.x] = e < n_entries ? dir[e * 2] : 0u;
  const int64_t last = (e0 + 255 < n_entries ? e0 + 255 : n_entries - 1);
  if (e == last) s_start[256] = s_start[threadIdx.x] + n_here;  // (end of the block's range; entries past `last` hold 0 runs)
  __syncthreads();
  const uint32_t lo = s_start[0], hi = s_start[256];
  const int n_valid = (int)(last - e0 + 1);
  for (uint32_t r = lo + threadIdx.x; r < hi; r += 256) {
    int a = 0, b = n_valid - 1;  // the last entry whose start ≤ r (entries with 0 runs share a start: the LAST of them that has runs wins by the search below)
    while (a < b) { const int m = (a + b + 1) >> 1; if (s_start[m] <= r) a = m; else b = m - 1; }
    const int64_t entry = e0 + a;
    int seg = 0;
    while (seg + 1 < segs.n_segs && (int64_t)segs.first_entry[seg + 1] <= entry) seg++;
    phys[r] = ((unsigned long long)seg << 32) | (unsigned long long)(s_base[a] + (r - s_start[a]));
  }
}
hipError_t fdb_launch_runs_map(const uint32_t* dir, const uint32_t* starts, int64_t n_entries, const FdbRunSegs& segs, unsigned long long* phys, hipStream_t stream) {
  if (n_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(runs_map_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, stream, dir, starts, n_entries, segs, phys);
  return hipGetLastError();
}

typedef uint32_t run_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void runs_flags_kernel(const unsigned long long* __restrict__ phys, int64_t n_runs, const FdbRunSegs segs, const unsigned char* __restrict__ rank,
                                                         int n_cols, uint32_t* __restrict__ flags, unsigned int* __restrict__ violation) {
  __shared__ unsigned char s_rank[FDB_RUN_TUPLE_BYTES * 256];
  for (int i = threadIdx.x; i < n_cols * 256; i += 256) s_rank[i] = rank[i];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_runs) return;
  const unsigned long long p = phys[i];
  const run_u32x4* t = reinterpret_cast<const run_u32x4*>(segs.tuples[p >> 32] + (p & 0xFFFFFFFFull) * FDB_RUN_BYTES);
  const run_u32x4 a0 = t[0], a1 = t[1];
  if (i == 0) { flags[0] = 1u; return; }
  const unsigned long long q = phys[i - 1];
  const run_u32x4* u = reinterpret_cast<const run_u32x4*>(segs.tuples[q >> 32] + (q & 0xFFFFFFFFull) * FDB_RUN_BYTES);
  const run_u32x4 b0 = u[0], b1 = u[1];
  const bool same = a0.x == b0.x && a0.y == b0.y && a0.z == b0.z && a0.w == b0.w && a1.x == b1.x && a1.y == b1.y && a1.z == b1.z && a1.w == b1.w;
  flags[i] = same ? 0u : 1u;
  if (same) return;
  // a new key must sort after its predecessor: first column whose ranks differ decides (ascending, NULLs last)
  const uint32_t cur[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, prev[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  for (int c = 0; c < n_cols; c++) {
    const uint32_t ic = (cur[c >> 2] >> (8 * (c & 3))) & 0xFFu, ip = (prev[c >> 2] >> (8 * (c & 3))) & 0xFFu;
    if (ic == ip) continue;
    if (s_rank[c * 256 + ic] < s_rank[c * 256 + ip]) atomicOr(violation, 1u);
    return;
  }
}
hipError_t fdb_launch_runs_flags(const unsigned long long* phys, int64_t n_runs, const FdbRunSegs& segs, const unsigned char* rank, int n_cols, uint32_t* flags,
                                 unsigned int* violation, hipStream_t stream) {
  if (n_runs <= 0) return hipSuccess;
  hipLaunchKernelGGL(runs_flags_kernel, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, stream, phys, n_runs, segs, rank, n_cols, flags, violation);
  return hipGetLastError();
}

// Wide segments (or a mix): the two runs' columns are compared one by one in plan order. A narrow run's id of column c is byte c of
// its tuple (columns the plan gained later: 0), a wide run's the word `cols[c].word` of its key tuple (past the tuple's end: 0 —
// the record of that launch did not know the column); int64 columns exist in wide runs only, NULL = bit `gi` of the valid mask clear.
__device__ __forceinline__ uint64_t run_bytes(int rw) { return rw == 0 ? (uint64_t)FDB_RUN_BYTES : rw == FDB_RUN_MEDIUM_WORDS ? (uint64_t)FDB_RUN_MEDIUM_BYTES : (uint64_t)rw * 4; }
struct RunRef { const uint32_t* t; int kw; };  // kw: key words of a wide run, 0 = narrow (a byte per id), -1 = medium (two bytes per id)
__device__ __forceinline__ RunRef run_ref(const FdbRunSegs& segs, unsigned long long p) {
  const int seg = (int)(p >> 32);
  const int rw = segs.run_words[seg];
  RunRef r;
  r.t = reinterpret_cast<const uint32_t*>(segs.tuples[seg] + (p & 0xFFFFFFFFull) * run_bytes(rw));
  r.kw = rw == 0 ? 0 : rw == FDB_RUN_MEDIUM_WORDS ? -1 : rw - 4;
  return r;
}
__device__ __forceinline__ uint32_t run_dict_id(const RunRef& r, int c, int word) {
  if (r.kw == 0) return c < FDB_RUN_TUPLE_BYTES ? (r.t[c >> 2] >> (8 * (c & 3))) & 0xFFu : 0u;
  if (r.kw < 0) return c < FDB_RUN_TUPLE_BYTES ? (r.t[c >> 1] >> (16 * (c & 1))) & 0xFFFFu : 0u;
  return word < r.kw ? r.t[word] : 0u;
}
__device__ __forceinline__ bool run_i64(const RunRef& r, int word, int gi, unsigned long long* v) {
  if (r.kw <= 0 || word + 1 >= r.kw) return false;
  const unsigned long long vm = (unsigned long long)r.t[0] | ((unsigned long long)r.t[1] << 32);
  if (!((vm >> gi) & 1ull)) return false;
  *v = (unsigned long long)r.t[word] | ((unsigned long long)r.t[word + 1] << 32);
  return true;
}
__global__ __launch_bounds__(256) void runs_flags_wide_kernel(const unsigned long long* __restrict__ phys, int64_t n_runs, const FdbRunSegs segs, const FdbRunCol* __restrict__ cols,
                                                              const uint32_t* __restrict__ rank32, int n_cols, uint32_t* __restrict__ flags, unsigned int* __restrict__ violation) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_runs) return;
  if (i == 0) { flags[0] = 1u; return; }
  const RunRef a = run_ref(segs, phys[i]), b = run_ref(segs, phys[i - 1]);
  typedef const __attribute__((address_space(4))) FdbRunCol* ConstRunCols;
  ConstRunCols q = (ConstRunCols)cols;
  if (a.kw == -1 && b.kw == -1) {
    // two medium records (32 two-byte ids each): four 16-byte loads per record instead of a 4-byte load per column and record — 80 bytes
    // apart from lane to lane, every one of those was a sector of its own (0.92 ms per 10 M runs); equal tuples, the usual case inside a
    // group that a wave or record boundary cut, are decided by 8 compares
    const run_u32x4* ta = reinterpret_cast<const run_u32x4*>(a.t);
    const run_u32x4* tb = reinterpret_cast<const run_u32x4*>(b.t);
    uint32_t wa[16], wb[16];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const run_u32x4 x = ta[v], y = tb[v];
      wa[4 * v] = x.x; wa[4 * v + 1] = x.y; wa[4 * v + 2] = x.z; wa[4 * v + 3] = x.w;
      wb[4 * v] = y.x; wb[4 * v + 1] = y.y; wb[4 * v + 2] = y.z; wb[4 * v + 3] = y.w;
    }
    uint32_t diff = 0;
#pragma unroll
    for (int v = 0; v < 16; v++) diff |= (wa[v] != wb[v]) ? (1u << v) : 0u;
    if (diff == 0u) { flags[i] = 0u; return; }
    flags[i] = 1u;
    int c = 2 * __builtin_ctz(diff);  // the first word that differs holds columns c and c + 1
    uint32_t xa = 0, xb = 0;
#pragma unroll
    for (int v = 0; v < 16; v++) if (v == (c >> 1)) { xa = wa[v]; xb = wb[v]; }
    if ((xa & 0xFFFFu) == (xb & 0xFFFFu)) { c++; xa >>= 16; xb >>= 16; }
    const uint32_t ic = xa & 0xFFFFu, ip = xb & 0xFFFFu;
    if (c < n_cols && q[c].kind == 0) {
      const uint32_t off = q[c].rank_off;
      if (rank32[off + ic] < rank32[off + ip]) atomicOr(violation, 1u);
    }
    return;
  }
  if (a.kw == 0 && b.kw == 0) {
    // two narrow records (32 one-byte ids each): two 16-byte loads per record; after a sort the records of neighbouring positions lie
    // anywhere, and the column-by-column walk below cost a dependent byte load per column and record (0.69 ms per 10 M runs)
    const run_u32x4* ta = reinterpret_cast<const run_u32x4*>(a.t);
    const run_u32x4* tb = reinterpret_cast<const run_u32x4*>(b.t);
    uint32_t wa[8], wb[8];
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const run_u32x4 x = ta[v], y = tb[v];
      wa[4 * v] = x.x; wa[4 * v + 1] = x.y; wa[4 * v + 2] = x.z; wa[4 * v + 3] = x.w;
      wb[4 * v] = y.x; wb[4 * v + 1] = y.y; wb[4 * v + 2] = y.z; wb[4 * v + 3] = y.w;
    }
    uint32_t diff = 0;
#pragma unroll
    for (int v = 0; v < 8; v++) diff |= (wa[v] != wb[v]) ? (1u << v) : 0u;
    if (diff == 0u) { flags[i] = 0u; return; }
    flags[i] = 1u;
    const int wv = __builtin_ctz(diff);  // the first word that differs holds columns 4 wv … 4 wv + 3
    uint32_t xa = 0, xb = 0;
#pragma unroll
    for (int v = 0; v < 8; v++) if (v == wv) { xa = wa[v]; xb = wb[v]; }
    const int byte = __builtin_ctz(xa ^ xb) >> 3;
    const int c = 4 * wv + byte;
    const uint32_t ic = (xa >> (8 * byte)) & 0xFFu, ip = (xb >> (8 * byte)) & 0xFFu;
    if (c < n_cols && q[c].kind == 0) {
      const uint32_t off = q[c].rank_off;
      if (rank32[off + ic] < rank32[off + ip]) atomicOr(violation, 1u);
    }
    return;
  }
  for (int c = 0; c < n_cols; c++) {
    const int kind = q[c].kind, word = q[c].word, gi = q[c].gi;
    if (kind == 0) {
      const uint32_t ic = run_dict_id(a, c, word), ip = run_dict_id(b, c, word);
      if (ic == ip) continue;
      flags[i] = 1u;
      const uint32_t off = q[c].rank_off;
      if (rank32[off + ic] < rank32[off + ip]) atomicOr(violation, 1u);  // ascending by value, NULL (rank 0xFFFFFFFF) last
      return;
    }
    unsigned long long vc = 0, vp = 0;
    const bool hc = run_i64(a, word, gi, &vc), hp = run_i64(b, word, gi, &vp);
    if (hc == hp && (!hc || vc == vp)) continue;
    flags[i] = 1u;
    bool before;  // does the new key sort BEFORE its predecessor?
    if (hc != hp) before = hc;  // a value after a NULL
    else before = kind == 3 ? vc < vp : (long long)vc < (long long)vp;
    if (before) atomicOr(violation, 1u);
    return;
  }
  flags[i] = 0u;
}
hipError_t fdb_launch_runs_flags_wide(const unsigned long long* phys, int64_t n_runs, const FdbRunSegs& segs, const FdbRunCol* cols, const uint32_t* rank32, int n_cols,
                                      uint32_t* flags, unsigned int* violation, hipStream_t stream) {
  if (n_runs <= 0) return hipSuccess;
  hipLaunchKernelGGL(runs_flags_wide_kernel, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, stream, phys, n_runs, segs, cols, rank32, n_cols, flags, violation);
  return hipGetLastError();
}

// The sort key of every run for one pass of the LSD sort that restores key order (FdbRunKeyPass, fdb_kernels.h). `rows` != nullptr: the
// things being sorted are not runs but dense key rows of `row_kw` words (an ordered plan's groups out of the hash table: a row has the
// layout of a wide run's key tuple) and phys[i] is a row number.
__global__ __launch_bounds__(256) void runs_sort_keys_kernel(const unsigned long long* __restrict__ phys, int64_t n_runs, const FdbRunSegs segs, const uint32_t* __restrict__ rows, int row_kw,
                                                             const FdbRunCol* __restrict__ cols, const uint32_t* __restrict__ rank32, const FdbRunKeyPass ps, unsigned long long* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_runs) return;
  RunRef a;
  if (rows != nullptr) { a.t = rows + phys[i] * (unsigned long long)row_kw; a.kw = row_kw; }
  else a = run_ref(segs, phys[i]);
  typedef const __attribute__((address_space(4))) FdbRunCol* ConstRunCols;
  ConstRunCols q = (ConstRunCols)cols;
  unsigned long long key = 0;
  if (ps.mode == 0 && a.kw <= 0) {
    // narrow / medium records: the 32 / 64 bytes of key ids come in as two / four 16-byte loads (the record of position i lies anywhere:
    // a byte load per column made every column a round trip of its own — 1.33 ms per 10 M runs and pass), wait in LDS word by word so
    // that a column can index them, and the ranks are looked up four columns at a time
    __shared__ uint32_t s_words[16][256];
    const int nq = a.kw == 0 ? 2 : 4;
    const run_u32x4* t = reinterpret_cast<const run_u32x4*>(a.t);
#pragma unroll
    for (int v = 0; v < 4; v++) {
      if (v < nq) {
        const run_u32x4 x = t[v];
        s_words[4 * v][threadIdx.x] = x.x; s_words[4 * v + 1][threadIdx.x] = x.y; s_words[4 * v + 2][threadIdx.x] = x.z; s_words[4 * v + 3][threadIdx.x] = x.w;
      }
    }
    const bool narrow = a.kw == 0;
    auto id_of = [&](int c) -> uint32_t {
      if (c >= FDB_RUN_TUPLE_BYTES) return 0u;
      return narrow ? (s_words[c >> 2][threadIdx.x] >> (8 * (c & 3))) & 0xFFu : (s_words[c >> 1][threadIdx.x] >> (16 * (c & 1))) & 0xFFFFu;
    };
    for (int k0 = 0; k0 < ps.n; k0 += 4) {
      uint32_t r[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u < ps.n ? k0 + u : ps.n - 1;
        const int c = ps.col[k];
        r[u] = rank32[q[c].rank_off + id_of(c)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (k0 + u < ps.n) {
          uint32_t x = r[u];
          if (x == 0xFFFFFFFFu) x = ps.null_rank[k0 + u];  // NULL sorts after every value
          key |= (unsigned long long)x << ps.shift[k0 + u];
        }
      }
    }
  } else if (ps.mode == 0) {
    for (int k = 0; k < ps.n; k++) {
      const int c = ps.col[k];
      const uint32_t id = run_dict_id(a, c, q[c].word);
      uint32_t r = rank32[q[c].rank_off + id];
      if (r == 0xFFFFFFFFu) r = ps.null_rank[k];  // NULL sorts after every value
      key |= (unsigned long long)r << ps.shift[k];
    }
  } else {
    const int c = ps.col[0];
    unsigned long long v = 0;
    const bool has = run_i64(a, q[c].word, q[c].gi, &v);
    if (ps.mode == 1) key = has ? (q[c].kind == 3 ? v : v ^ 0x8000000000000000ull) : 0ull;
    else key = has ? 0ull : 1ull;
  }
  keys[i] = key;
}
hipError_t fdb_launch_runs_sort_keys(const unsigned long long* phys, int64_t n_runs, const FdbRunSegs* segs, const uint32_t* rows, int row_kw, const FdbRunCol* cols,
                                     const uint32_t* rank32, const FdbRunKeyPass& pass, unsigned long long* keys, hipStream_t stream) {
  if (n_runs <= 0) return hipSuccess;
  if ((segs == nullptr) == (rows == nullptr)) return hipErrorInvalidValue;
  FdbRunSegs none;
  if (segs == nullptr) std::memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(runs_sort_keys_kernel, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, stream, phys, n_runs, segs != nullptr ? *segs : none, rows, row_kw, cols, rank32, pass, keys);
  return hipGetLastError();
}

// order[i] = i;  dst row i = src row order[i] (rows of `words` 32-bit words; one thread per word);  dst[i] = src[order[i]]
__global__ __launch_bounds__(256) void iota_u64_kernel(unsigned long long* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (unsigned long long)i;
}
__global__ __launch_bounds__(256) void gather_rows_u32_kernel(const uint32_t* __restrict__ src, int words, const unsigned long long* __restrict__ order, int64_t n, uint32_t* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * words) return;
  const int64_t i = t / words;
  const int w = (int)(t - i * words);
  dst[t] = src[order[i] * (unsigned long long)words + (unsigned long long)w];
}
__global__ __launch_bounds__(256) void gather_u64_kernel(const unsigned long long* __restrict__ src, const unsigned long long* __restrict__ order, int64_t n, unsigned long long* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[order[i]];
}
hipError_t fdb_launch_iota_u64(unsigned long long* p, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(iota_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n);
  return hipGetLastError();
}
hipError_t fdb_launch_gather_rows_u32(const uint32_t* src, int words, const unsigned long long* order, int64_t n, uint32_t* dst, hipStream_t stream) {
  if (n <= 0 || words <= 0) return hipSuccess;
  const int64_t total = n * words;
  if ((total + 255) / 256 > 0x7FFFFFFFll) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_rows_u32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, words, order, n, dst);
  return hipGetLastError();
}
hipError_t fdb_launch_gather_u64(const unsigned long long* src, const unsigned long long* order, int64_t n, unsigned long long* dst, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, order, n, dst);
  return hipGetLastError();
}

// One wave per 64 consecutive logical runs. The key rows of the groups that START among them are consecutive in the output
// (out_idx is monotone), so they are assembled in the wave's LDS tile and leave as one contiguous copy.
// See FdbRunsTranslateArgs. A thread re-keys one run; narrow / medium records are composed in registers (column index = compile-time
// constant of an unrolled loop), a wide record is written word by word.
__global__ __launch_bounds__(256) void runs_translate_kernel(const FdbRunsTranslateArgs a, const FdbRunSegs segs) {
  typedef const __attribute__((address_space(4))) FdbHashCol* ConstHashCols;
  ConstHashCols q = (ConstHashCols)a.cols;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x == 0) {
    const int64_t first = (int64_t)blockIdx.x * 256;
    a.out_dir[(int64_t)blockIdx.x * 2] = (uint32_t)first;
    a.out_dir[(int64_t)blockIdx.x * 2 + 1] = (uint32_t)(a.n_runs - first < 256 ? a.n_runs - first : 256);
  }
  if (i >= a.n_runs) return;
  const RunRef r = run_ref(segs, a.phys[i]);
  const uint32_t* tail = r.kw == 0 ? r.t + 8 : r.kw < 0 ? r.t + 16 : r.t + r.kw;  // {rows of the run, its aggregate}
  const u64x2 ca = *reinterpret_cast<const u64x2*>(tail);
  auto dst_id = [&](int c) -> uint32_t {  // destination column c's key id in this run (0: NULL or the source lacks the column)
    if (q[c].src_word < 0) return 0u;
    uint32_t id = run_dict_id(r, (int)q[c].lut_len, q[c].src_word);
    const uint32_t* lut = q[c].lut;
    if (id != 0u && lut != nullptr) id = lut[id];
    return id;
  };
  const int orw = a.out_run_words;
  if (orw == 0) {
    uint32_t w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < FDB_RUN_TUPLE_BYTES; c++) if (c < a.n_cols) w[c >> 2] |= (dst_id(c) & 0xFFu) << (8 * (c & 3));
    run_u32x4* o = reinterpret_cast<run_u32x4*>(a.out_tuples + (uint64_t)i * FDB_RUN_BYTES);
    o[0] = run_u32x4{w[0], w[1], w[2], w[3]}; o[1] = run_u32x4{w[4], w[5], w[6], w[7]};
    *reinterpret_cast<u64x2*>(o + 2) = ca;
  } else if (orw == FDB_RUN_MEDIUM_WORDS) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = 0u;
#pragma unroll
    for (int c = 0; c < FDB_RUN_TUPLE_BYTES; c++) if (c < a.n_cols) w[c >> 1] |= (dst_id(c) & 0xFFFFu) << (16 * (c & 1));
    run_u32x4* o = reinterpret_cast<run_u32x4*>(a.out_tuples + (uint64_t)i * FDB_RUN_MEDIUM_BYTES);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = run_u32x4{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
    *reinterpret_cast<u64x2*>(o + 4) = ca;
  } else {
    const int okw = orw - 4;
    uint32_t* o = reinterpret_cast<uint32_t*>(a.out_tuples + (uint64_t)i * (uint64_t)orw * 4);
    for (int w = 0; w < okw; w++) o[w] = 0u;
    unsigned long long vm = 0;
    for (int c = 0; c < a.n_cols; c++) {
      if (q[c].kind == 0) {
        const uint32_t id = dst_id(c);
        if (id != 0u) vm |= 1ull << q[c].gi;
        o[q[c].word] = id;
      } else {
        unsigned long long v = 0;
        if (q[c].src_word >= 0 && run_i64(r, q[c].src_word, (int)q[c].lut_len, &v)) {
          vm |= 1ull << q[c].gi;
          o[q[c].word] = (uint32_t)v; o[q[c].word + 1] = (uint32_t)(v >> 32);
        }
      }
    }
    o[0] = (uint32_t)vm; o[1] = (uint32_t)(vm >> 32);
    *reinterpret_cast<u64x2*>(o + okw) = ca;
  }
}
hipError_t fdb_launch_runs_translate(const FdbRunsTranslateArgs& args, const FdbRunSegs& segs, hipStream_t stream) {
  if (args.n_runs <= 0) return hipSuccess;
  hipLaunchKernelGGL(runs_translate_kernel, dim3((unsigned)((args.n_runs + 255) / 256)), dim3(256), 0, stream, args, segs);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void runs_expand_kernel(const FdbRunsExpandArgs a, const FdbRunSegs segs) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kw = a.key_words;
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * 64 * kw;
  const int64_t i = ((int64_t)blockIdx.x * 4 + wave) * 64 + lane;
  const bool live = i < a.n_runs;
  uint32_t fl = 0, g = 0;
  unsigned long long cnt = 0, acc = 0;
  run_u32x4 t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0};
  const uint32_t* wide = nullptr;  // a wide run's key tuple (already a dense key row of `wide_kw` words)
  const uint32_t* medium = nullptr;  // a medium run's 32 two-byte ids
  int wide_kw = 0;
  if (live) {
    const unsigned long long p = a.phys[i];
    const int seg = (int)(p >> 32);
    const uint64_t at = p & 0xFFFFFFFFull;
    const int rw = segs.run_words[seg];
    if (rw == 0) {
      const run_u32x4* t = reinterpret_cast<const run_u32x4*>(segs.tuples[seg] + at * FDB_RUN_BYTES);
      t0 = t[0]; t1 = t[1];
      { const u64x2 ca = *reinterpret_cast<const u64x2*>(t + 2); cnt = ca.x; acc = ca.y; }
    } else if (rw == FDB_RUN_MEDIUM_WORDS) {
      medium = reinterpret_cast<const uint32_t*>(segs.tuples[seg] + at * FDB_RUN_MEDIUM_BYTES);
      { const u64x2 ca = *reinterpret_cast<const u64x2*>(medium + 16); cnt = ca.x; acc = ca.y; }
    } else {
      wide = reinterpret_cast<const uint32_t*>(segs.tuples[seg] + at * (uint64_t)rw * 4);
      wide_kw = rw - 4;
      { const u64x2 ca = *reinterpret_cast<const u64x2*>(wide + wide_kw); cnt = ca.x; acc = ca.y; }
    }
    if (a.flags != nullptr) {
      fl = a.flags[i];
      g = a.out_idx[i] - (fl ? 0u : 1u);
    } else { fl = 1u; g = (uint32_t)i; }
  }
  const unsigned long long starts = __ballot(live && fl != 0u);
  // Runs of one group that sit next to each other in this wave are folded across its lanes first (g is monotone over the lanes, so a
  // Kogge-Stone scan with "same group at distance d" is a segmented scan) and the LAST lane of such a stretch applies the total: a plain
  // store when the whole group lies inside the wave, else one atomic per (wave, group). Without this a group that spans hundreds of runs
  // — long groups: every wave-tile of the scan emits a run of it — was hundreds of atomics on ONE address from waves that run at the same
  // time: 78 ns each, one after the other (cfg 2's query over a table sorted by labels.path: 390 k runs of 1 023 groups, 60 ms).
  {
    const uint32_t gg = live ? g : 0xFFFFFFFFu;
    uint32_t has_start = live && fl != 0u ? 1u : 0u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t pg = __shfl_up(gg, off, 64), ps = __shfl_up(has_start, off, 64);
      const unsigned long long pc = __shfl_up(cnt, off, 64), pa = __shfl_up(acc, off, 64);
      if (lane >= off && pg == gg && live) {
        cnt += pc;
        has_start |= ps;
        if (a.func == 1) acc += pa;
        else if (a.func == 2) acc = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)acc) + __longlong_as_double((long long)pa));
        else if (a.func == 3) acc = (unsigned long long)min((long long)acc, (long long)pa);
        else if (a.func == 4) acc = (unsigned long long)max((long long)acc, (long long)pa);
      }
    }
    const uint32_t ng = __shfl_down(gg, 1, 64);
    const bool seg_end = live && (lane == 63 || ng != gg);
    if (seg_end) {
      // does the group end with this run? (the next run starts a new key, or there is none)
      const bool ends = a.flags == nullptr || i + 1 >= a.n_runs || a.flags[i + 1] != 0u;
      const size_t vg = (size_t)g * (size_t)(a.val_stride > 0 ? a.val_stride : 1);
      if (has_start != 0u && ends) { a.vals_cnt[vg] = cnt; a.vals_acc[vg] = acc; }
      else {
        atomicAdd(a.vals_cnt + vg, cnt);
        if (a.func == 1) atomicAdd(a.vals_acc + vg, acc);
        else if (a.func == 2) atomicAdd(reinterpret_cast<double*>(a.vals_acc) + vg, __longlong_as_double((long long)acc));
        else if (a.func == 3) atomicMin(reinterpret_cast<long long*>(a.vals_acc) + vg, (long long)acc);
        else if (a.func == 4) atomicMax(reinterpret_cast<long long*>(a.vals_acc) + vg, (long long)acc);
      }
    }
  }
  if (starts == 0ull) return;
  const uint32_t first_g = __shfl(g, __builtin_ctzll(starts), 64);  // (g of the first starting lane = the wave's first output row)
  const uint32_t n_new = (uint32_t)__popcll(starts);
  if (live && fl != 0u) {
    const uint32_t r = g - first_g;
    uint32_t* row = tile + (size_t)r * kw;
    if (wide != nullptr) {
      for (int w = 0; w < kw; w++) row[w] = w < wide_kw ? wide[w] : 0u;  // (columns the plan gained after this run's launch: NULL)
      row[2] = 0u; row[3] = 0u;
    } else if (medium != nullptr) {
      for (int w = 0; w < kw; w++) row[w] = 0u;
      unsigned long long vm = 0;
      const int nc = a.n_cols < FDB_RUN_TUPLE_BYTES ? a.n_cols : FDB_RUN_TUPLE_BYTES;
      for (int c = 0; c < nc; c++) {
        const uint32_t id = (medium[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
        row[a.col_word[c]] = id;
        if (id != 0u) vm |= 1ull << c;
      }
      row[0] = (uint32_t)vm; row[1] = (uint32_t)(vm >> 32);
    } else {
      for (int w = 0; w < kw; w++) row[w] = 0u;
      const uint32_t ids[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      unsigned long long vm = 0;
      const int nc = a.n_cols < FDB_RUN_TUPLE_BYTES ? a.n_cols : FDB_RUN_TUPLE_BYTES;
      for (int c = 0; c < nc; c++) {
        const uint32_t id = (ids[c >> 2] >> (8 * (c & 3))) & 0xFFu;
        row[a.col_word[c]] = id;
        if (id != 0u) vm |= 1ull << c;
      }
      row[0] = (uint32_t)vm; row[1] = (uint32_t)(vm >> 32);
    }
  }
  __builtin_amdgcn_wave_barrier();
  uint32_t* dst = a.dense_keys + (size_t)first_g * kw;
  for (uint32_t t = lane; t < n_new * (uint32_t)kw; t += 64) dst[t] = tile[t];
}
hipError_t fdb_launch_runs_expand(const FdbRunsExpandArgs& args, const FdbRunSegs& segs, hipStream_t stream) {
  if (args.n_runs <= 0) return hipSuccess;
  const size_t lds = (size_t)4 * 64 * (size_t)args.key_words * 4;
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&runs_expand_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(runs_expand_kernel, dim3((unsigned)((args.n_runs + 255) / 256)), dim3(256), lds, stream, args, segs);
  return hipGetLastError();
}

__global__ void fill_u64_kernel(unsigned long long* p, int64_t n, unsigned long long v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
hipError_t fdb_launch_fill_u64(unsigned long long* p, int64_t n, unsigned long long v, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, p, n, v);
  return hipGetLastError();
}
