// mapped_flag_probe.hip — how long does the host wait for a small result? (tuning aid for DESIGN.md §4, "next candidate")
//
// Two ways of getting a 16 KiB table from the last kernel of a stream to the host:
//   A. what Plan::fetch_state does today: hipMemcpyAsync(device → pinned) + hipStreamSynchronize;
//   B. the kernel writes the table into MAPPED pinned host memory, the last workgroup to finish (device-scope counter) issues a
//      system-scope fence and writes an epoch flag there; the host spins on the flag — no copy command, no driver wait.
// Prints the mean host-observed latency from launch to "result readable" for both, over a kernel of ~5 µs.
//   hipcc --offload-arch=gfx950 -O2 tools/mapped_flag_probe.hip -o tools/mapped_flag_probe && ./tools/mapped_flag_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kSlots = 1024;           // unsigned long long per array
constexpr int kArrays = 2;             // cnt + one accumulator, like cfg 2
constexpr int kBlocks = 8, kThreads = 256;

__global__ void fold_to_device(const unsigned long long* partials, unsigned long long* state) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < kSlots * kArrays; s += gridDim.x * blockDim.x) state[s] += partials[s];
}

__global__ void fold_to_host(const unsigned long long* partials, unsigned long long* state, unsigned long long* host_mirror,
                             unsigned int* done_counter, volatile unsigned long long* host_flag, unsigned long long epoch) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < kSlots * kArrays; s += gridDim.x * blockDim.x) {
    const unsigned long long v = state[s] + partials[s];
    state[s] = v;
    __builtin_nontemporal_store(v, &host_mirror[s]);
  }
  __threadfence_system();  // this workgroup's host writes are visible before it reports in
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = atomicAdd(done_counter, 1u);
    if (prev == gridDim.x - 1) {  // last workgroup: everybody's data is out
      *done_counter = 0;
      __threadfence_system();
      *host_flag = epoch;
    }
  }
}

int main() {
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  unsigned long long *d_part, *d_state, *h_pinned, *h_mapped, *d_mapped;
  unsigned int* d_counter;
  const size_t bytes = (size_t)kSlots * kArrays * 8;
  CK(hipMalloc(&d_part, bytes));
  CK(hipMalloc(&d_state, bytes));
  CK(hipMalloc(&d_counter, 64));
  CK(hipMemset(d_part, 1, bytes));
  CK(hipMemset(d_state, 0, bytes));
  CK(hipMemset(d_counter, 0, 64));
  CK(hipHostMalloc((void**)&h_pinned, bytes, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&h_mapped, bytes + 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&d_mapped, h_mapped, 0));
  std::memset(h_mapped, 0, bytes + 64);
  volatile unsigned long long* h_flag = h_mapped + kSlots * kArrays;
  unsigned long long* d_flag = d_mapped + kSlots * kArrays;
  const int reps = 2000;
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int warm = 0; warm < 2; warm++) {
    double a_us = 0, b_us = 0;
    for (int r = 0; r < reps; r++) {
      auto t0 = now();
      hipLaunchKernelGGL(fold_to_device, dim3(kBlocks), dim3(kThreads), 0, stream, d_part, d_state);
      CK(hipMemcpyAsync(h_pinned, d_state, bytes, hipMemcpyDeviceToHost, stream));
      CK(hipStreamSynchronize(stream));
      a_us += std::chrono::duration<double, std::micro>(now() - t0).count();
    }
    for (int r = 0; r < reps; r++) {
      const unsigned long long epoch = (unsigned long long)warm * reps + r + 1;
      auto t0 = now();
      hipLaunchKernelGGL(fold_to_host, dim3(kBlocks), dim3(kThreads), 0, stream, d_part, d_state, d_mapped, d_counter, d_flag, epoch);
      while (*h_flag != epoch) { /* spin */ }
      b_us += std::chrono::duration<double, std::micro>(now() - t0).count();
    }
    CK(hipStreamSynchronize(stream));
    if (warm == 1) {
      std::printf("A copy + stream wait : %7.2f us per result\n", a_us / reps);
      std::printf("B mapped table + flag: %7.2f us per result\n", b_us / reps);
      // the two paths fold the same partials the same number of times: the mirror must equal the device table
      CK(hipMemcpy(h_pinned, d_state, bytes, hipMemcpyDeviceToHost));
      std::printf("mirror %s the device table\n", std::memcmp(h_pinned, h_mapped, bytes) == 0 ? "matches" : "DIFFERS FROM");
    }
  }
  return 0;
}
