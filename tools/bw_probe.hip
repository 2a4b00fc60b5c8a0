// bw_probe.hip — streaming-read ceiling of this box for the scan kernel's access pattern (tuning aid).
// Reads the same column mix as cfg 2 (two uint32 index columns, one float64 column, two bitmaps) with no
// aggregation work, for several workgroup sizes / grid sizes / load flavours. Prints GB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <bool NT, typename T> __device__ __forceinline__ T ld(const T* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// R rows per lane per tile; tile = BLOCK * R rows; grid-stride over tiles.
template <int BLOCK, int R, bool NT>
__global__ __launch_bounds__(BLOCK) void probe(const uint32_t* a, const uint32_t* b, const unsigned long long* v, const uint8_t* ba,
                                               const uint8_t* bb, int64_t n_rows, unsigned long long* out) {
  unsigned long long acc = 0;
  const int64_t tile_rows = (int64_t)BLOCK * R;
  const int64_t n_tiles = n_rows / tile_rows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * tile_rows + (int64_t)threadIdx.x * R;
#pragma unroll
    for (int j = 0; j < R / 4; j++) {
      u32x4 x = ld<NT>(reinterpret_cast<const u32x4*>(a + row0) + j);
      u32x4 y = ld<NT>(reinterpret_cast<const u32x4*>(b + row0) + j);
      acc += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w;
    }
#pragma unroll
    for (int j = 0; j < R / 2; j++) {
      u64x2 z = ld<NT>(reinterpret_cast<const u64x2*>(v + row0) + j);
      acc += z.x + z.y;
    }
    acc += ba[row0 >> 3] + bb[row0 >> 3];
  }
  if (acc == 0x123456789ull) out[0] = acc;
}

// V1: two tiles in flight per wave (software-pipelined loads).
template <int BLOCK, int R, bool NT>
__global__ __launch_bounds__(BLOCK) void probe_unroll2(const uint32_t* a, const uint32_t* b, const unsigned long long* v, const uint8_t* ba,
                                               const uint8_t* bb, int64_t n_rows, unsigned long long* out) {
  unsigned long long acc = 0;
  const int64_t tile_rows = (int64_t)BLOCK * R;
  const int64_t n_tiles = n_rows / tile_rows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += 2 * gridDim.x) {
    const int64_t t2 = tile + gridDim.x < n_tiles ? tile + gridDim.x : tile;
    const int64_t row0 = tile * tile_rows + (int64_t)threadIdx.x * R;
    const int64_t row1 = t2 * tile_rows + (int64_t)threadIdx.x * R;
    u32x4 x0 = ld<NT>(reinterpret_cast<const u32x4*>(a + row0)), x1 = ld<NT>(reinterpret_cast<const u32x4*>(a + row1));
    u32x4 y0 = ld<NT>(reinterpret_cast<const u32x4*>(b + row0)), y1 = ld<NT>(reinterpret_cast<const u32x4*>(b + row1));
    u64x2 z0 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0)), z1 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0) + 1);
    u64x2 z2 = ld<NT>(reinterpret_cast<const u64x2*>(v + row1)), z3 = ld<NT>(reinterpret_cast<const u64x2*>(v + row1) + 1);
    acc += ba[row0 >> 3] + bb[row0 >> 3] + ba[row1 >> 3] + bb[row1 >> 3];
    acc += x0.x + x0.y + x0.z + x0.w + y0.x + y0.y + y0.z + y0.w + x1.x + x1.y + x1.z + x1.w + y1.x + y1.y + y1.z + y1.w;
    acc += z0.x + z0.y + z1.x + z1.y + z2.x + z2.y + z3.x + z3.y;
  }
  if (acc == 0x123456789ull) out[0] = acc;
}

// V2: persistent workgroups pull SUPER-tiles (SUB tiles each) from a global counter, next index prefetched.
template <int BLOCK, int R, bool NT, int SUB>
__global__ __launch_bounds__(BLOCK) void probe_dynamic(const uint32_t* a, const uint32_t* b, const unsigned long long* v, const uint8_t* ba,
                                               const uint8_t* bb, int64_t n_rows, unsigned long long* out, unsigned int* counter) {
  __shared__ unsigned int s_next[2];
  unsigned long long acc = 0;
  const int64_t tile_rows = (int64_t)BLOCK * R;
  const int64_t n_super = n_rows / (tile_rows * SUB);
  unsigned int cur = blockIdx.x;
  if (threadIdx.x == 0) s_next[0] = atomicAdd(counter, 1u) + gridDim.x;
  int ph = 0;
  while (cur < n_super) {
    __syncthreads();
    const unsigned int nxt = s_next[ph];
    if (threadIdx.x == 0) s_next[ph ^ 1] = atomicAdd(counter, 1u) + gridDim.x;  // prefetch the one after
    ph ^= 1;
#pragma unroll
    for (int s = 0; s < SUB; s++) {
      const int64_t row0 = ((int64_t)cur * SUB + s) * tile_rows + (int64_t)threadIdx.x * R;
      u32x4 x = ld<NT>(reinterpret_cast<const u32x4*>(a + row0));
      u32x4 y = ld<NT>(reinterpret_cast<const u32x4*>(b + row0));
      u64x2 z0 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0)), z1 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0) + 1);
      acc += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z0.x + z0.y + z1.x + z1.y + ba[row0 >> 3] + bb[row0 >> 3];
    }
    cur = nxt;
  }
  if (acc == 0x123456789ull) out[0] = acc;
}

// V4: the f64 column read as 16 B per lane, lane-contiguous (rows 2*lane..2*lane+1 of two half tiles) — checks
// whether the 32 B-stride dwordx4 pairs of V0 cost bandwidth.
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void probe_split(const uint32_t* a, const uint32_t* b, const unsigned long long* v, const uint8_t* ba,
                                               const uint8_t* bb, int64_t n_rows, unsigned long long* out) {
  unsigned long long acc = 0;
  const int64_t tile_rows = (int64_t)BLOCK * 4;
  const int64_t n_tiles = n_rows / tile_rows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t base = tile * tile_rows;
    const int64_t row0 = base + (int64_t)threadIdx.x * 4;
    u32x4 x = ld<NT>(reinterpret_cast<const u32x4*>(a + row0));
    u32x4 y = ld<NT>(reinterpret_cast<const u32x4*>(b + row0));
    u64x2 z0 = ld<NT>(reinterpret_cast<const u64x2*>(v + base) + threadIdx.x);
    u64x2 z1 = ld<NT>(reinterpret_cast<const u64x2*>(v + base) + BLOCK + threadIdx.x);
    acc += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z0.x + z0.y + z1.x + z1.y + ba[row0 >> 3] + bb[row0 >> 3];
  }
  if (acc == 0x123456789ull) out[0] = acc;
}

// V5: true software pipeline — tile i+1's loads are issued before tile i's data is consumed.
template <int BLOCK, bool NT, int CHUNK>
__global__ __launch_bounds__(BLOCK) void probe_pipe(const uint32_t* a, const uint32_t* b, const unsigned long long* v, const uint8_t* ba,
                                               const uint8_t* bb, int64_t n_rows, unsigned long long* out) {
  unsigned long long acc = 0;
  const int64_t tile_rows = (int64_t)BLOCK * 4;
  const int64_t n_tiles = n_rows / tile_rows;
  // CHUNK == 0: grid-stride; CHUNK > 0: each workgroup owns runs of CHUNK consecutive tiles
  auto tile_of = [&](int64_t k) -> int64_t {
    if (CHUNK == 0) return (int64_t)blockIdx.x + k * gridDim.x;
    return ((k / CHUNK) * gridDim.x + blockIdx.x) * CHUNK + (k % CHUNK);
  };
  int64_t k = 0;
  int64_t tile = tile_of(0);
  if (tile >= n_tiles) return;
  int64_t row0 = tile * tile_rows + (int64_t)threadIdx.x * 4;
  u32x4 x = ld<NT>(reinterpret_cast<const u32x4*>(a + row0));
  u32x4 y = ld<NT>(reinterpret_cast<const u32x4*>(b + row0));
  u64x2 z0 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0)), z1 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0) + 1);
  uint32_t m = ba[row0 >> 3] + bb[row0 >> 3];
  for (;;) {
    k++;
    const int64_t nt = tile_of(k);
    const bool more = nt < n_tiles;
    u32x4 nx = x, ny = y; u64x2 nz0 = z0, nz1 = z1; uint32_t nm = 0;
    if (more) {
      const int64_t r1 = nt * tile_rows + (int64_t)threadIdx.x * 4;
      nx = ld<NT>(reinterpret_cast<const u32x4*>(a + r1));
      ny = ld<NT>(reinterpret_cast<const u32x4*>(b + r1));
      nz0 = ld<NT>(reinterpret_cast<const u64x2*>(v + r1)); nz1 = ld<NT>(reinterpret_cast<const u64x2*>(v + r1) + 1);
      nm = ba[r1 >> 3] + bb[r1 >> 3];
    }
    acc += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z0.x + z0.y + z1.x + z1.y + m;
    if (!more) break;
    x = nx; y = ny; z0 = nz0; z1 = nz1; m = nm;
  }
  if (acc == 0x123456789ull) out[0] = acc;
}

// V6: static grid-stride for the first `n_static` tiles, then the remaining tiles are pulled from a counter
// (work stealing for the tail only: fast CUs/XCDs take more tiles).
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void probe_hybrid(const uint32_t* a, const uint32_t* b, const unsigned long long* v, const uint8_t* ba,
                                               const uint8_t* bb, int64_t n_rows, unsigned long long* out, unsigned int* counter, int64_t n_static) {
  __shared__ unsigned int s_t;
  unsigned long long acc = 0;
  const int64_t tile_rows = (int64_t)BLOCK * 4;
  const int64_t n_tiles = n_rows / tile_rows;
  auto body = [&](int64_t tile) {
    const int64_t row0 = tile * tile_rows + (int64_t)threadIdx.x * 4;
    u32x4 x = ld<NT>(reinterpret_cast<const u32x4*>(a + row0));
    u32x4 y = ld<NT>(reinterpret_cast<const u32x4*>(b + row0));
    u64x2 z0 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0)), z1 = ld<NT>(reinterpret_cast<const u64x2*>(v + row0) + 1);
    acc += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z0.x + z0.y + z1.x + z1.y + ba[row0 >> 3] + bb[row0 >> 3];
  };
  for (int64_t tile = blockIdx.x; tile < n_static; tile += gridDim.x) body(tile);
  for (;;) {
    if (threadIdx.x == 0) s_t = atomicAdd(counter, 1u);
    __syncthreads();
    const int64_t tile = n_static + s_t;
    __syncthreads();
    if (tile >= n_tiles) break;
    body(tile);
  }
  if (acc == 0x123456789ull) out[0] = acc;
}

// V7: hand-specialised cfg-2 scan (what a JIT-specialised plan kernel would look like): code ∈ bitmask, path →
// LDS LUT → slot, sum(value) into an LDS table, occupancy flags, plain-store flush. Measures the instruction-issue
// headroom of specialisation over the interpreting kernels.
template <int BLK>
__global__ __launch_bounds__(BLK) void probe_cfg2(const uint32_t* __restrict__ code, const uint32_t* __restrict__ path, const double* __restrict__ val,
                                                  const uint8_t* __restrict__ bcode, const uint8_t* __restrict__ bpath, const uint32_t* __restrict__ lut_g,
                                                  uint32_t lut_len, unsigned long long code_bits, int64_t n_rows, unsigned long long* partials, uint32_t n_slots) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
  uint32_t* l_cnt = lut + ((lut_len + 3) & ~3u);
  double* l_sum = reinterpret_cast<double*>(l_cnt + ((n_slots + 3) & ~3u));
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < lut_len; i += BLK) lut[i] = lut_g[i];
  for (uint32_t i = tid; i < n_slots; i += BLK) { l_cnt[i] = 0; l_sum[i] = 0.0; }
  __syncthreads();
  const int64_t tile_rows = (int64_t)BLK * 4;
  const int64_t n_tiles = n_rows / tile_rows;
  const uint32_t lane_off = tid * 16u;       // byte offset of this lane's 4 uint32 inside a tile
  const uint32_t null_code = 63u, null_path = lut_len - 1u;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * tile_rows;     // wave-uniform
    const char* pc = reinterpret_cast<const char*>(code + r0);
    const char* pp = reinterpret_cast<const char*>(path + r0);
    const char* pv = reinterpret_cast<const char*>(val + r0);
    const u32x4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pc + lane_off));
    const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pp + lane_off));
    const u64x2 v0 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(pv + 2u * lane_off));
    const u64x2 v1 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(pv + 2u * lane_off + 16u));
    const uint32_t vc = (bcode[(r0 >> 3) + (tid >> 1)] >> ((tid & 1u) * 4u)) & 0xFu;
    const uint32_t vp = (bpath[(r0 >> 3) + (tid >> 1)] >> ((tid & 1u) * 4u)) & 0xFu;
    const uint32_t ci[4] = {c.x, c.y, c.z, c.w}, pi[4] = {q.x, q.y, q.z, q.w};
    const double vv[4] = {__longlong_as_double((long long)v0.x), __longlong_as_double((long long)v0.y), __longlong_as_double((long long)v1.x), __longlong_as_double((long long)v1.y)};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t ce = ((vc >> r) & 1u) ? ci[r] : null_code;
      if ((code_bits >> ce) & 1ull) {
        const uint32_t slot = lut[((vp >> r) & 1u) ? pi[r] : null_path];
        l_cnt[slot] = 1u;
        atomicAdd(&l_sum[slot], vv[r]);
      }
    }
  }
  __syncthreads();
  unsigned long long* out = partials + (size_t)blockIdx.x * 2 * n_slots;
  for (uint32_t i = tid; i < n_slots; i += BLK) { out[i] = l_cnt[i]; out[n_slots + i] = (unsigned long long)__double_as_longlong(l_sum[i]); }
}

// V8: hand-specialised cfg-3 scan: (code ∈ bits) AND (method ∈ bits) AND instance IS NOT NULL; COUNT, MIN(ts), MAX(ts),
// SUM(value) BY path. Two-phase: the three filter inputs first, then path / value / timestamp for lanes with a hit.
template <int BLK, bool TWO_PHASE>
__global__ __launch_bounds__(BLK) void probe_cfg3(const uint32_t* __restrict__ code, const uint32_t* __restrict__ method, const uint32_t* __restrict__ path,
                                                  const double* __restrict__ val, const long long* __restrict__ ts, const uint8_t* __restrict__ bcode,
                                                  const uint8_t* __restrict__ bmethod, const uint8_t* __restrict__ binst, const uint8_t* __restrict__ bpath,
                                                  const uint32_t* __restrict__ lut_g, uint32_t lut_len, unsigned long long code_bits,
                                                  unsigned long long method_bits, int64_t n_rows, unsigned long long* partials, uint32_t n_slots) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
  uint32_t* l_cnt = lut + ((lut_len + 3) & ~3u);
  double* l_sum = reinterpret_cast<double*>(l_cnt + ((n_slots + 3) & ~3u));
  long long* l_min = reinterpret_cast<long long*>(l_sum + n_slots);
  long long* l_max = l_min + n_slots;
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < lut_len; i += BLK) lut[i] = lut_g[i];
  for (uint32_t i = tid; i < n_slots; i += BLK) { l_cnt[i] = 0; l_sum[i] = 0.0; l_min[i] = 0x7FFFFFFFFFFFFFFFLL; l_max[i] = -0x7FFFFFFFFFFFFFFFLL - 1; }
  __syncthreads();
  const int64_t tile_rows = (int64_t)BLK * 4;
  const int64_t n_tiles = n_rows / tile_rows;
  const uint32_t lane_off = tid * 16u;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * tile_rows;
    const u32x4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(code + r0) + lane_off));
    const u32x4 m = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(method + r0) + lane_off));
    const uint32_t bsh = (tid & 1u) * 4u;
    const int64_t boff = (r0 >> 3) + (tid >> 1);
    const uint32_t vc = (bcode[boff] >> bsh) & 0xFu, vm = (bmethod[boff] >> bsh) & 0xFu, vi = (binst[boff] >> bsh) & 0xFu;
    u32x4 q; u64x2 v0, v1, t0, t1; uint32_t vp;
    if (!TWO_PHASE) {
      q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(path + r0) + lane_off));
      v0 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(val + r0) + 2u * lane_off));
      v1 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(val + r0) + 2u * lane_off + 16u));
      t0 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(ts + r0) + 2u * lane_off));
      t1 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(ts + r0) + 2u * lane_off + 16u));
      vp = (bpath[boff] >> bsh) & 0xFu;
    }
    const uint32_t ci[4] = {c.x, c.y, c.z, c.w}, mi[4] = {m.x, m.y, m.z, m.w};
    uint32_t sel = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t ce = ((vc >> r) & 1u) ? ci[r] : 63u, me = ((vm >> r) & 1u) ? mi[r] : 63u;
      sel |= (uint32_t)(((code_bits >> ce) & 1ull) & ((method_bits >> me) & 1ull) & ((vi >> r) & 1u)) << r;
    }
    if (sel == 0) continue;
    if (TWO_PHASE) {
      q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(path + r0) + lane_off));
      v0 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(val + r0) + 2u * lane_off));
      v1 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(val + r0) + 2u * lane_off + 16u));
      t0 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(ts + r0) + 2u * lane_off));
      t1 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(ts + r0) + 2u * lane_off + 16u));
      vp = (bpath[boff] >> bsh) & 0xFu;
    }
    const uint32_t pi[4] = {q.x, q.y, q.z, q.w};
    const double vv[4] = {__longlong_as_double((long long)v0.x), __longlong_as_double((long long)v0.y), __longlong_as_double((long long)v1.x), __longlong_as_double((long long)v1.y)};
    const long long tt[4] = {(long long)t0.x, (long long)t0.y, (long long)t1.x, (long long)t1.y};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if ((sel >> r) & 1u) {
        const uint32_t slot = lut[((vp >> r) & 1u) ? pi[r] : lut_len - 1u];
        atomicAdd(&l_cnt[slot], 1u);
        atomicAdd(&l_sum[slot], vv[r]);
        atomicMin(&l_min[slot], tt[r]);
        atomicMax(&l_max[slot], tt[r]);
      }
    }
  }
  __syncthreads();
  unsigned long long* out = partials + (size_t)blockIdx.x * 4 * n_slots;
  for (uint32_t i = tid; i < n_slots; i += BLK) {
    out[i] = l_cnt[i]; out[n_slots + i] = (unsigned long long)__double_as_longlong(l_sum[i]);
    out[2 * n_slots + i] = (unsigned long long)l_min[i]; out[3 * n_slots + i] = (unsigned long long)l_max[i];
  }
}

template <typename F>
float time_it(F&& launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) launch();
  CHECK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int i = 0; i < reps; i++) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int BLOCK, int R, bool NT>
float run(int grid, const uint32_t* a, const uint32_t* b, const unsigned long long* v, const uint8_t* ba, const uint8_t* bb, int64_t n,
          unsigned long long* out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((probe<BLOCK, R, NT>), dim3(grid), dim3(BLOCK), 0, 0, a, b, v, ba, bb, n, out);
  CHECK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe<BLOCK, R, NT>), dim3(grid), dim3(BLOCK), 0, 0, a, b, v, ba, bb, n, out);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000;
  uint32_t *a, *b; unsigned long long *v, *out; uint8_t *ba, *bb;
  CHECK(hipMalloc(&a, n * 4 + 4096)); CHECK(hipMalloc(&b, n * 4 + 4096)); CHECK(hipMalloc(&v, n * 8 + 4096));
  CHECK(hipMalloc(&ba, n / 8 + 4096)); CHECK(hipMalloc(&bb, n / 8 + 4096)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(a, 1, n * 4)); CHECK(hipMemset(b, 2, n * 4)); CHECK(hipMemset(v, 3, n * 8)); CHECK(hipMemset(ba, 0xff, n / 8)); CHECK(hipMemset(bb, 0xff, n / 8));
  const double bytes = (double)n * 16.25;
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s CUs=%d rows=%lld bytes=%.1f MB\n", prop.name, cus, (long long)n, bytes / 1e6);
#define RUN(BLOCK, R, NT, G) { float ms = run<BLOCK, R, NT>(G, a, b, v, ba, bb, n, out); printf("block=%4d R=%d nt=%d grid=%5d  %.4f ms  %.1f GB/s\n", BLOCK, R, NT, G, ms, bytes / ms / 1e6); }
  for (int mult : {1, 2, 4, 8, 16}) {
    RUN(256, 4, true, cus * mult * 1); RUN(256, 8, true, cus * mult);
  }
  for (int mult : {1, 2, 4}) { RUN(512, 4, true, cus * mult); RUN(512, 8, true, cus * mult); }
  for (int mult : {1, 2}) { RUN(1024, 4, true, cus * mult); RUN(1024, 8, true, cus * mult); RUN(1024, 4, false, cus * mult); RUN(1024, 8, false, cus * mult); }
  RUN(256, 4, false, cus * 8); RUN(256, 8, false, cus * 8); RUN(512, 4, false, cus * 4);
  // non-persistent: one tile per workgroup
  RUN(256, 4, true, (int)(n / (256 * 4))); RUN(256, 8, true, (int)(n / (256 * 8))); RUN(1024, 4, true, (int)(n / (1024 * 4)));
  unsigned int* counter; CHECK(hipMalloc(&counter, 4));
#define REPORT(name, G, ms) printf("%-28s grid=%5d  %.4f ms  %.1f GB/s\n", name, G, ms, bytes / ms / 1e6)
  for (int G : {cus, cus * 2, cus * 4}) {
    float ms = time_it([&] { hipLaunchKernelGGL((probe_unroll2<1024, 4, true>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("unroll2 block=1024", G, ms);
    ms = time_it([&] { hipLaunchKernelGGL((probe_unroll2<256, 4, true>), dim3(G * 4), dim3(256), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("unroll2 block=256", G * 4, ms);
  }
  for (int G : {cus, cus * 2}) {
    float ms = time_it([&] { CHECK(hipMemsetAsync(counter, 0, 4, 0)); hipLaunchKernelGGL((probe_dynamic<1024, 4, true, 1>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out, counter); });
    REPORT("dynamic sub=1 block=1024", G, ms);
    ms = time_it([&] { CHECK(hipMemsetAsync(counter, 0, 4, 0)); hipLaunchKernelGGL((probe_dynamic<1024, 4, true, 4>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out, counter); });
    REPORT("dynamic sub=4 block=1024", G, ms);
    ms = time_it([&] { hipLaunchKernelGGL((probe_split<1024, true>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("split-f64 block=1024", G, ms);
    ms = time_it([&] { hipLaunchKernelGGL((probe_split<1024, false>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("split-f64 nt=0 block=1024", G, ms);
  }
  for (int G : {cus, cus * 2}) {
    float ms = time_it([&] { hipLaunchKernelGGL((probe_pipe<1024, true, 0>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("pipe block=1024 stride", G, ms);
    ms = time_it([&] { hipLaunchKernelGGL((probe_pipe<1024, true, 4>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("pipe block=1024 chunk4", G, ms);
    ms = time_it([&] { hipLaunchKernelGGL((probe_pipe<1024, true, 16>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("pipe block=1024 chunk16", G, ms);
    ms = time_it([&] { hipLaunchKernelGGL((probe_pipe<256, true, 0>), dim3(G * 4), dim3(256), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("pipe block=256 stride", G * 4, ms);
    ms = time_it([&] { hipLaunchKernelGGL((probe_pipe<512, true, 0>), dim3(G * 2), dim3(512), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("pipe block=512 stride", G * 2, ms);
  }
  {
    // realistic index contents for the specialised scan: code ∈ [0,6), path ∈ [0,1024)
    std::vector<uint32_t> hc(n), hp(n), hl(1025);
    uint64_t x = 88172645463325252ull;
    for (int64_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hc[i] = (x % 10) < 7 ? 0 : 1 + (x >> 8) % 5; hp[i] = (uint32_t)((x >> 20) % 1024); }
    for (int i = 0; i < 1025; i++) hl[i] = i < 1024 ? i + 1 : 0;
    CHECK(hipMemcpy(a, hc.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, hp.data(), n * 4, hipMemcpyHostToDevice));
    uint32_t* lut; CHECK(hipMalloc(&lut, 1025 * 4)); CHECK(hipMemcpy(lut, hl.data(), 1025 * 4, hipMemcpyHostToDevice));
    unsigned long long* partials; CHECK(hipMalloc(&partials, (size_t)4096 * 2 * 1025 * 8));
    const size_t lds = (1028 + 1028) * 4 + 1025 * 8;
    for (int G : {cus * 2, cus * 4}) {
      float ms = time_it([&] { hipLaunchKernelGGL((probe_cfg2<512>), dim3(G), dim3(512), lds, 0, a, b, (const double*)v, ba, bb, lut, 1025u, 1ull, n, partials, 1025u); });
      REPORT("specialised cfg2 blk=512", G, ms);
      ms = time_it([&] { hipLaunchKernelGGL((probe_cfg2<1024>), dim3(G / 2), dim3(1024), lds, 0, a, b, (const double*)v, ba, bb, lut, 1025u, 1ull, n, partials, 1025u); });
      REPORT("specialised cfg2 blk=1024", G / 2, ms);
      ms = time_it([&] { hipLaunchKernelGGL((probe_cfg2<256>), dim3(G * 2), dim3(256), lds, 0, a, b, (const double*)v, ba, bb, lut, 1025u, 1ull, n, partials, 1025u); });
      REPORT("specialised cfg2 blk=256", G * 2, ms);
    }
    {
      // cfg 3: + method (4 values), instance validity (5 % NULL), timestamp
      uint32_t* meth; long long* tsd; uint8_t* binst;
      CHECK(hipMalloc(&meth, n * 4 + 4096)); CHECK(hipMalloc(&tsd, n * 8 + 4096)); CHECK(hipMalloc(&binst, n / 8 + 4096));
      std::vector<uint32_t> hm(n); std::vector<uint8_t> hb(n / 8 + 1, 0xFF);
      uint64_t y = 0x9E3779B97F4A7C15ull;
      for (int64_t i = 0; i < n; i++) { y ^= y << 13; y ^= y >> 7; y ^= y << 17; hm[i] = (uint32_t)(y & 3); if ((y >> 10) % 20 == 0) hb[i >> 3] &= ~(1u << (i & 7)); }
      CHECK(hipMemcpy(meth, hm.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(binst, hb.data(), n / 8, hipMemcpyHostToDevice));
      CHECK(hipMemset(tsd, 5, n * 8));
      const double bytes3 = (double)n * 28.5;
      const size_t lds3 = (1028 + 1028) * 4 + 1025 * 8 * 3;
      for (int G : {cus * 2, cus * 3, cus * 4}) {
        float ms = time_it([&] { hipLaunchKernelGGL((probe_cfg3<512, true>), dim3(G), dim3(512), lds3, 0, a, meth, b, (const double*)v, tsd, ba, bb, binst, bb, lut, 1025u, 5ull, 1ull, n, partials, 1025u); });
        printf("%-28s grid=%5d  %.4f ms  %.1f GB/s\n", "specialised cfg3 2-phase", G, ms, bytes3 / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL((probe_cfg3<512, false>), dim3(G), dim3(512), lds3, 0, a, meth, b, (const double*)v, tsd, ba, bb, binst, bb, lut, 1025u, 5ull, 1ull, n, partials, 1025u); });
        printf("%-28s grid=%5d  %.4f ms  %.1f GB/s\n", "specialised cfg3 1-phase", G, ms, bytes3 / ms / 1e6);
      }
    }
    CHECK(hipMemset(a, 1, n * 4)); CHECK(hipMemset(b, 2, n * 4));
  }
  for (int G : {cus, cus * 2}) {
    const int64_t nt = n / 4096;
    for (int pct : {50, 80, 90, 95}) {
      const int64_t ns = nt * pct / 100 / G * G;
      float ms = time_it([&] { CHECK(hipMemsetAsync(counter, 0, 4, 0)); hipLaunchKernelGGL((probe_hybrid<1024, true>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out, counter, ns); });
      char nm[64]; snprintf(nm, sizeof nm, "hybrid static=%d%% b=1024", pct);
      REPORT(nm, G, ms);
    }
  }
  for (int G : {cus * 8, cus * 16, cus * 32, cus * 64}) {
    float ms = run<256, 4, true>(G, a, b, v, ba, bb, n, out);
    REPORT("static block=256 R=4", G, ms);
  }
  { int G = (int)(n / 4096); float ms = time_it([&] { hipLaunchKernelGGL((probe_split<1024, true>), dim3(G), dim3(1024), 0, 0, a, b, v, ba, bb, n, out); });
    REPORT("split-f64 non-persistent", G, ms); }
  return 0;
}
