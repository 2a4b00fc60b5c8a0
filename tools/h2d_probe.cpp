// Measurement aid (round 4): what does it cost to get a 1 MB host record (a 65 536-row Callback) into HBM, per strategy?
//   A  memcpy into a pinned slab by the calling thread, one DMA per 8 records (what Plan::push does)
//   B  the same memcpy split over k spinning helper threads
//   C  hipMemcpyAsync straight from the caller's pageable buffer (the runtime stages it), one per record
//   D  the caller's buffer registered once (hipHostRegister), DMA straight from it, one wait per record (the buffer is only borrowed)
// g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/h2d_probe.cpp -o tools/h2d_probe -L/opt/rocm/lib -lamdhip64 -lpthread
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct SpinPool {  // helpers spin on a generation counter: no wake-up latency (they burn their cores while the chain is pushing)
  std::vector<std::thread> th;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> left{0};
  std::atomic<bool> stop{false};
  const unsigned char* src = nullptr; unsigned char* dst = nullptr; size_t bytes = 0; int parts = 1;
  explicit SpinPool(int k) {
    for (int i = 0; i < k; i++) th.emplace_back([this, i] {
      uint64_t seen = 0;
      for (;;) {
        uint64_t g;
        while ((g = gen.load(std::memory_order_acquire)) == seen) { if (stop.load()) return; __builtin_ia32_pause(); }
        seen = g;
        copy_part(i + 1);
        left.fetch_sub(1, std::memory_order_release);
      }
    });
  }
  void copy_part(int i) { const size_t piece = (bytes / parts + 63) & ~(size_t)63, off = (size_t)i * piece; if (off < bytes) memcpy(dst + off, src + off, std::min(piece, bytes - off)); }
  void copy(void* d, const void* s, size_t n) {
    src = (const unsigned char*)s; dst = (unsigned char*)d; bytes = n; parts = (int)th.size() + 1;
    left.store((int)th.size(), std::memory_order_relaxed);
    gen.fetch_add(1, std::memory_order_release);
    copy_part(0);
    while (left.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }
  ~SpinPool() { stop.store(true); for (auto& t : th) t.join(); }
};

int main() {
  const size_t rec = 1064960, n_rec = 512, total = rec * n_rec;  // 65 536 rows × 16.25 B
  unsigned char* src = (unsigned char*)malloc(total);
  memset(src, 7, total);
  unsigned char *pinned, *dev;
  CHECK(hipHostMalloc((void**)&pinned, total, hipHostMallocDefault));
  CHECK(hipMalloc((void**)&dev, total));
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto report = [&](const char* what, double dt) { printf("%-70s %7.1f us per 1 MB record = %5.1f GB/s\n", what, dt / n_rec * 1e6, total / dt / 1e9); };
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (size_t i = 0; i < n_rec; i++) {
      memcpy(pinned + i * rec, src + i * rec, rec);
      if ((i & 7) == 7) CHECK(hipMemcpyAsync(dev + (i - 7) * rec, pinned + (i - 7) * rec, 8 * rec, hipMemcpyHostToDevice, s));
    }
    CHECK(hipStreamSynchronize(s));
    if (rep) report("A memcpy into the pinned slab (1 thread) + one DMA per 8 records", now() - t0);
  }
  for (int k : {1, 3, 7}) {
    SpinPool pool(k);
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now();
      for (size_t i = 0; i < n_rec; i++) {
        pool.copy(pinned + i * rec, src + i * rec, rec);
        if ((i & 7) == 7) CHECK(hipMemcpyAsync(dev + (i - 7) * rec, pinned + (i - 7) * rec, 8 * rec, hipMemcpyHostToDevice, s));
      }
      CHECK(hipStreamSynchronize(s));
      char buf[128]; snprintf(buf, sizeof buf, "B memcpy split over the caller + %d spinning helpers + one DMA per 8 records", k);
      if (rep) report(buf, now() - t0);
    }
  }
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (size_t i = 0; i < n_rec; i++) CHECK(hipMemcpyAsync(dev + i * rec, src + i * rec, rec, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    if (rep) report("C hipMemcpyAsync from the pageable buffer, one per record", now() - t0);
  }
  {
    double t0 = now();
    CHECK(hipHostRegister(src, total, hipHostRegisterDefault));
    printf("  (hipHostRegister of %.0f MB: %.1f ms)\n", total / 1e6, (now() - t0) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
      t0 = now();
      for (size_t i = 0; i < n_rec; i++) { CHECK(hipMemcpyAsync(dev + i * rec, src + i * rec, rec, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s)); }
      if (rep) report("D registered buffer: DMA straight from it, one wait per record", now() - t0);
    }
    for (int rep = 0; rep < 2; rep++) {
      t0 = now();
      for (size_t i = 0; i < n_rec; i++) { CHECK(hipMemcpyAsync(dev + i * rec, src + i * rec, rec, hipMemcpyHostToDevice, s)); }
      CHECK(hipStreamSynchronize(s));
      if (rep) report("D' registered buffer, no per-record wait (a caller that keeps the record until Finish)", now() - t0);
    }
    CHECK(hipHostUnregister(src));
  }
  return 0;
}
