// Measurement aid: what HBM delivers for MIXED read/write streams on this box — the ceiling of anything that compacts or copies
// (filter(), Finish of big tables). Pure reads reach ≈7 TB/s (tools/bw_probe.hip); a stream kernel that also writes shares the bus
// with its own write-backs. Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o tools/copy_probe && tools/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define G1 __attribute__((address_space(1)))

// reads R input streams of n 16-byte elements, writes W output streams (W ≤ R): out_w[i] = in_w[i] ^ in_{w+1}[i] …
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void mix(const u64x2* __restrict__ in, u64x2* __restrict__ out, long long n) {
  u64x2 sink = {0, 0};  // (W == 0: every loaded value is folded into this and the fold is stored under a condition the compiler cannot
                        // decide — the round-3 version xor-ed v[0] with itself, the loads were dead and the "read only" rows read 92–96 TB/s)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    u64x2 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = NT ? __builtin_nontemporal_load((const G1 u64x2*)(in + (long long)r * n + i)) : in[(long long)r * n + i];
    u64x2 acc = v[0];
#pragma unroll
    for (int r = (W > 0 ? W : 1); r < R; r++) acc ^= v[r];
#pragma unroll
    for (int w = 0; w < W; w++) { u64x2 o = w == 0 ? acc : v[w]; if (NT) __builtin_nontemporal_store(o, (G1 u64x2*)(out + (long long)w * n + i)); else out[(long long)w * n + i] = o; }
    if (W == 0) sink += acc;
  }
  if (W == 0 && sink.x == 0x1234567ull && sink.y == 0x7654321ull) out[0] = sink;
}

template <int R, int W, bool NT>
void run(const char* name, const u64x2* in, u64x2* out, long long n, int cus) {
  for (int per_cu : {2, 4, 8, 16}) {
    const int grid = cus * per_cu;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((mix<R, W, NT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    CHECK(hipEventRecord(a));
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL((mix<R, W, NT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double bytes = (double)(R + W) * n * 16;
    printf("%-34s nt=%d blocks/CU=%2d  %.4f ms  %.0f GB/s total (%d read : %d written)\n", name, (int)NT, per_cu, ms, bytes / ms / 1e6, R, W);
  }
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const long long n = (1024ll << 20) / 16;  // 1 GiB per stream: well past the 256 MB Infinity Cache
  u64x2 *in, *out;
  CHECK(hipMalloc(&in, (size_t)n * 16 * 3)); CHECK(hipMalloc(&out, (size_t)n * 16 * 2));
  CHECK(hipMemset(in, 1, (size_t)n * 16 * 3)); CHECK(hipMemset(out, 0, (size_t)n * 16 * 2));
  printf("# %s CUs=%d, %lld MiB per stream\n", prop.name, prop.multiProcessorCount, n * 16 >> 20);
  const int cus = prop.multiProcessorCount;
  run<1, 0, true>("read only", in, out, n, cus);
  run<2, 0, true>("read only, two streams", in, out, n, cus);
  run<1, 1, false>("copy (1 read : 1 written)", in, out, n, cus);
  run<1, 1, true>("copy (1 read : 1 written)", in, out, n, cus);
  run<2, 1, false>("compaction-like (2 read : 1 written)", in, out, n, cus);
  run<2, 1, true>("compaction-like (2 read : 1 written)", in, out, n, cus);
  run<3, 1, true>("3 read : 1 written", in, out, n, cus);
  return 0;
}
