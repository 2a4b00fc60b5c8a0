// fdb_context.cpp — see fdb_context.h.
#include "fdb_context.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <string>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "fdb_plan.h"

namespace fdb {

namespace {
std::mutex g_mu;
std::vector<Context*> g_free[16];
std::atomic<int64_t> g_live_dev_blocks{0}, g_live_dev_bytes{0}, g_live_pinned{0};
inline size_t round_block(size_t b) {
  size_t r = 4096;
  while (r < b) r <<= 1;
  return r;
}
}  // namespace

Context* Context::acquire(int device) {
  if (device < 0 || device >= 16) throw Error(FDB_ERR_INVALID, "device index out of range");
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_free[device].empty()) {
      Context* c = g_free[device].back();
      g_free[device].pop_back();
      return c;
    }
  }
  hip_check(hipSetDevice(device), "hipSetDevice");
  Context* c = new Context();
  c->device = device;
  // The reference runs GOMAXPROCS chains (physicalplan.go:22: 256 on the bench host). Every chain gets a context, but not every context a
  // stream of its own: the contexts of a device share $FDB_MAX_STREAMS streams (default 4; 0 = one stream per context). With a stream per
  // chain the host→device rate of N chains pushing 65 536-row records fell from the link's 3.1 G rows/s to 1.4 beyond 32 chains (the
  // runtime spreads the streams' copies over the same few copy engines and hardware queues); with 4 shared streams it stays at 2.9–3.2
  // up to 128 chains (profiles/round6_push_bench.txt). Contexts that share a stream merely wait for each other's work in their own
  // stream waits; nothing in the library makes one stream's kernel wait for another kernel on the device across plans.
  static const int max_streams = std::getenv("FDB_MAX_STREAMS") ? std::atoi(std::getenv("FDB_MAX_STREAMS")) : 4;
  if (max_streams > 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    static std::vector<hipStream_t> shared[16];
    static size_t created[16];
    const size_t k = created[device]++ % (size_t)max_streams;
    if (k < shared[device].size()) { c->stream = shared[device][k]; return c; }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { created[device]--; delete c; hip_check(e, "hipStreamCreate"); }
    shared[device].push_back(c->stream);
    return c;
  }
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; hip_check(e, "hipStreamCreate"); }
  return c;
}

void Context::free_retired() {
  for (auto& r : retired_) { if (r.first) (void)hipHostFree(r.first); if (r.second) (void)hipFree(r.second); }
  retired_.clear();
}

void Context::release(Context* c) {
  if (c == nullptr) return;
  c->free_retired();
  c->stage_off_ = 0;
  std::lock_guard<std::mutex> lk(g_mu);
  g_free[c->device].push_back(c);
}

void* Context::dev_alloc(size_t bytes) {
  bytes = round_block(bytes ? bytes : 1);
  Block* best = nullptr;
  for (Block& b : dev_blocks_)
    if (!b.used && b.bytes >= bytes && (best == nullptr || b.bytes < best->bytes)) best = &b;
  if (best != nullptr && best->bytes <= bytes * 4) { best->used = true; note_device_alloc(best->bytes); return best->p; }
  void* p = nullptr;
  static const bool prof = std::getenv("FDB_PROFILE_ALLOC") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&p, bytes);
  if (prof && bytes >= ((size_t)1 << 20)) std::fprintf(stderr, "[fdb] hipMalloc(%zu) %.1f us\n", bytes, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  if (e == hipErrorOutOfMemory) {  // drop the cache and retry once
    (void)hipGetLastError();
    for (Block& b : dev_blocks_) if (!b.used) { (void)hipFree(b.p); b.p = nullptr; }
    dev_blocks_.erase(std::remove_if(dev_blocks_.begin(), dev_blocks_.end(), [](const Block& b) { return b.p == nullptr; }), dev_blocks_.end());
    e = hipMalloc(&p, bytes);
  }
  hip_check(e, "hipMalloc");
  dev_blocks_.push_back(Block{p, bytes, true});
  note_device_alloc(bytes);
  return p;
}

void Context::dev_free(void* p) {
  if (p == nullptr) return;
  for (Block& b : dev_blocks_) if (b.p == p) { if (b.used) note_device_free(b.bytes); b.used = false; return; }
}

void note_device_alloc(size_t bytes) { g_live_dev_blocks++; g_live_dev_bytes += (int64_t)bytes; }
void note_device_free(size_t bytes) { g_live_dev_blocks--; g_live_dev_bytes -= (int64_t)bytes; }
void live_allocations(int64_t* device_blocks, int64_t* device_bytes, int64_t* pinned_blocks) {
  if (device_blocks) *device_blocks = g_live_dev_blocks.load();
  if (device_bytes) *device_bytes = g_live_dev_bytes.load();
  if (pinned_blocks) *pinned_blocks = g_live_pinned.load();
}

// ---- pinned host memory next to the GPU ---------------------------------------------------------------------------------------
// hipHostMalloc places its pages by the CALLING thread's memory policy — on the node the thread happens to run on. The reference runs
// GOMAXPROCS chains (physicalplan.go:22), spread over every socket of the host; a chain on the far socket then gets pinned slabs on the
// far node and the GPU's DMA engines read them across the socket link (N chains pushing host records: the host→device rate falls once
// the GPU's own socket is full). The pinned blocks of this library are therefore allocated under a PREFERRED policy for the GPU's NUMA
// node (sysfs numa_node of its PCI function; $FDB_PINNED_NUMA=0: leave the policy alone — A/B aid). Raw syscalls: no libnuma here.
#include <sys/syscall.h>
#include <unistd.h>
namespace {
int gpu_numa_node(int device) {
  static std::mutex mu;
  static int cached[16];
  static bool known[16];
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || device >= 16) return -1;
  if (known[device]) return cached[device];
  known[device] = true; cached[device] = -1;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
  for (char* c = bus; *c; c++) *c = (char)std::tolower((unsigned char)*c);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
  if (FILE* f = std::fopen(path.c_str(), "r")) {
    int node = -1;
    if (std::fscanf(f, "%d", &node) == 1 && node >= 0 && node < 1024) cached[device] = node;
    std::fclose(f);
  }
  return cached[device];
}
struct PreferNode {  // the calling thread's memory policy = "prefer `node`" for the lifetime of the object
  bool set = false;
  explicit PreferNode(int device) {
    static const bool off = std::getenv("FDB_PINNED_NUMA") != nullptr && std::atoi(std::getenv("FDB_PINNED_NUMA")) == 0;
    const int node = off ? -1 : gpu_numa_node(device);
    if (node < 0) return;
    unsigned long mask[16] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    set = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, (unsigned long)(sizeof(mask) * 8)) == 0;
  }
  ~PreferNode() { if (set) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul); }
};
}  // namespace
hipError_t pinned_malloc_near(void** p, size_t bytes, int device) {
  PreferNode prefer(device);
  return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

void* Context::host_alloc(size_t bytes) {
  bytes = round_block(bytes ? bytes : 1);
  Block* best = nullptr;
  for (Block& b : host_blocks_)
    if (!b.used && b.bytes >= bytes && (best == nullptr || b.bytes < best->bytes)) best = &b;
  if (best != nullptr) { best->used = true; return best->p; }
  void* p = nullptr;
  static const bool prof = std::getenv("FDB_PROFILE_ALLOC") != nullptr;  // (tuning aid: what the driver's allocators cost when many chains start at once)
  const auto t0 = std::chrono::steady_clock::now();
  hip_check(pinned_malloc_near(&p, bytes, device), "hipHostMalloc");
  if (prof) std::fprintf(stderr, "[fdb] hipHostMalloc(%zu) %.1f us\n", bytes, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  host_blocks_.push_back(Block{p, bytes, true});
  return p;
}

void Context::host_free(void* p) {
  if (p == nullptr) return;
  for (Block& b : host_blocks_) if (b.p == p) { b.used = false; return; }
}

namespace {
struct PinnedBlock { void* p; size_t bytes; bool used; };
std::mutex g_pin_mu;
std::vector<PinnedBlock> g_pinned;
constexpr size_t kPinnedCacheBudget = (size_t)8 << 30;  // idle pinned bytes kept for re-use
}  // namespace

void* pinned_pool_alloc(size_t bytes) {
  const size_t want = (std::max<size_t>(bytes, 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);  // 2 MiB granules
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    PinnedBlock* best = nullptr;
    for (PinnedBlock& b : g_pinned)
      if (!b.used && b.bytes >= want && b.bytes <= want * 2 && (best == nullptr || b.bytes < best->bytes)) best = &b;
    if (best != nullptr) { best->used = true; g_live_pinned++; return best->p; }
  }
  void* p = nullptr;
  int cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  hip_check(pinned_malloc_near(&p, want, cur_dev), "hipHostMalloc(result)");
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pinned.push_back(PinnedBlock{p, want, true});
  g_live_pinned++;
  return p;
}

void pinned_pool_free(void* p) {
  if (p == nullptr) return;
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    size_t idle = 0;
    for (PinnedBlock& b : g_pinned) {
      if (b.p == p) { if (b.used) g_live_pinned--; b.used = false; }
      if (!b.used) idle += b.bytes;
    }
    // over budget: give the largest idle blocks back to the OS
    while (idle > kPinnedCacheBudget) {
      size_t k = g_pinned.size();
      for (size_t i = 0; i < g_pinned.size(); i++)
        if (!g_pinned[i].used && (k == g_pinned.size() || g_pinned[i].bytes > g_pinned[k].bytes)) k = i;
      if (k == g_pinned.size()) break;
      idle -= g_pinned[k].bytes;
      drop.push_back(g_pinned[k].p);
      g_pinned.erase(g_pinned.begin() + (long)k);
    }
  }
  for (void* q : drop) (void)hipHostFree(q);
}

namespace {
struct PoolBlock { void* p; size_t bytes; bool used; uint64_t freed_at; };  // freed_at: tick of the release that made it idle
std::atomic<uint64_t> g_pool_tick{0};
std::mutex g_dpool_mu;
std::vector<PoolBlock> g_dpool[16];
constexpr size_t kDevicePoolIdleBudget = (size_t)16 << 30;
}  // namespace

void* device_pool_alloc(int device, size_t bytes) {
  if (device < 0 || device >= 16) throw Error(FDB_ERR_INVALID, "device index out of range");
  const size_t want = (std::max<size_t>(bytes, 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  {
    std::lock_guard<std::mutex> lk(g_dpool_mu);
    PoolBlock* best = nullptr;
    for (PoolBlock& b : g_dpool[device])
      if (!b.used && b.bytes >= want && b.bytes <= want + want / 4 && (best == nullptr || b.bytes < best->bytes)) best = &b;
    if (best != nullptr) { best->used = true; note_device_alloc(best->bytes); return best->p; }
  }
  hip_check(hipSetDevice(device), "hipSetDevice");
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e == hipErrorOutOfMemory) {  // give the idle blocks back and retry once
    (void)hipGetLastError();
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> lk(g_dpool_mu);
      auto& v = g_dpool[device];
      for (PoolBlock& b : v) if (!b.used) drop.push_back(b.p);
      v.erase(std::remove_if(v.begin(), v.end(), [](const PoolBlock& b) { return !b.used; }), v.end());
    }
    for (void* q : drop) (void)hipFree(q);
    e = hipMalloc(&p, want);
  }
  hip_check(e, "hipMalloc(resident record)");
  std::lock_guard<std::mutex> lk(g_dpool_mu);
  g_dpool[device].push_back(PoolBlock{p, want, true, 0});
  note_device_alloc(want);
  return p;
}

void device_pool_free(int device, void* p) {
  if (p == nullptr || device < 0 || device >= 16) return;
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lk(g_dpool_mu);
    auto& v = g_dpool[device];
    size_t idle = 0;
    for (PoolBlock& b : v) {
      if (b.p == p && b.used) { b.used = false; b.freed_at = ++g_pool_tick; note_device_free(b.bytes); }
      if (!b.used) idle += b.bytes;
    }
    // over budget: the blocks that have been idle LONGEST go back to the driver (dropping the largest first threw away the one
    // big block a loop re-allocates every iteration while a pile of older mid-sized blocks stayed: a 1.7 GB result arena cost a
    // hipFree + hipMalloc per query, 50 ms, next to 16 GiB of idle parts)
    while (idle > kDevicePoolIdleBudget) {
      size_t k = v.size();
      for (size_t i = 0; i < v.size(); i++) if (!v[i].used && (k == v.size() || v[i].freed_at < v[k].freed_at)) k = i;
      if (k == v.size()) break;
      idle -= v[k].bytes;
      drop.push_back(v[k].p);
      v.erase(v.begin() + (long)k);
    }
  }
  if (!drop.empty()) { (void)hipSetDevice(device); for (void* q : drop) (void)hipFree(q); }
}

hipEvent_t Context::get_event() {
  if (!events_.empty()) { hipEvent_t e = events_.back(); events_.pop_back(); return e; }
  hipEvent_t e;
  hip_check(hipEventCreate(&e), "hipEventCreate");
  return e;
}

void Context::put_event(hipEvent_t e) { events_.push_back(e); }

unsigned long long* Context::select_ctl(size_t want, uint32_t* epoch, unsigned long long* ticket_base, unsigned long long* arrival_base) {
  const size_t words = FDB_SELECT_CTL_WORDS + want;
  if (words > select_words_ || select_epoch_ >= (1u << 24) - 1u) {
    if (words > select_words_) {
      // (owned by the context like its staging ring: not a dev_alloc block, which would count as a live allocation of the plan)
      if (select_ctl_ != nullptr) { (void)hipStreamSynchronize(stream); (void)hipFree(select_ctl_); select_ctl_ = nullptr; select_words_ = 0; }
      const size_t n = std::max<size_t>(words + words / 2, 1u << 16);
      void* p = nullptr;
      hip_check(hipMalloc(&p, n * 8), "hipMalloc(select control block)");
      select_ctl_ = (unsigned long long*)p;
      select_words_ = n;
    }
    hip_check(hipMemsetAsync(select_ctl_, 0, select_words_ * 8, stream), "hipMemsetAsync(select control block)");
    select_epoch_ = 0;
    select_ticket_ = 0;
    select_arrival_ = 0;
  }
  // (test hook: jump to the end of the 24-bit epoch range, so that the wrap — clear the block, start over — is exercised by a
  // handful of calls instead of sixteen million)
  if (std::getenv("FDB_TEST_SELECT_EPOCH_JUMP") != nullptr && select_epoch_ + 3 < (1u << 24) - 1u) select_epoch_ = (1u << 24) - 4u;
  *epoch = ++select_epoch_;
  *ticket_base = select_ticket_;
  *arrival_base = select_arrival_;
  return select_ctl_;
}

hipStream_t Context::aux_stream(int i) {
  if (i < 0 || i >= 3) i = 0;
  if (aux_[i] == nullptr) hip_check(hipStreamCreateWithFlags(&aux_[i], hipStreamNonBlocking), "hipStreamCreate(aux)");
  return aux_[i];
}

void Context::copy_out_parallel(void* host, const void* dev, size_t bytes) {
  constexpr size_t kMinSlice = (size_t)64 << 20;
  int parts = (int)std::min<size_t>(4, std::max<size_t>(1, bytes / kMinSlice));
  if (parts <= 1) {
    hip_check(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(result)");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    return;
  }
  for (int i = 0; i < parts - 1; i++)
    if (aux_[i] == nullptr) hip_check(hipStreamCreateWithFlags(&aux_[i], hipStreamNonBlocking), "hipStreamCreate(aux)");
  hipEvent_t ready = get_event();
  hip_check(hipEventRecord(ready, stream), "hipEventRecord");
  const size_t slice = ((bytes / (size_t)parts) + 4095) & ~(size_t)4095;
  hipError_t e = hipSuccess;
  for (int i = 0; i < parts && e == hipSuccess; i++) {
    const size_t off = (size_t)i * slice;
    if (off >= bytes) break;
    const size_t len = std::min(slice, bytes - off);
    hipStream_t s = i == 0 ? stream : aux_[i - 1];
    if (i > 0) e = hipStreamWaitEvent(s, ready, 0);
    if (e == hipSuccess) e = hipMemcpyAsync((unsigned char*)host + off, (const unsigned char*)dev + off, len, hipMemcpyDeviceToHost, s);
  }
  for (int i = 0; i < parts - 1; i++) { const hipError_t w = hipStreamSynchronize(aux_[i]); if (e == hipSuccess) e = w; }
  { const hipError_t w = hipStreamSynchronize(stream); if (e == hipSuccess) e = w; }
  put_event(ready);
  hip_check(e, "parallel device→host copy");
}

void* Context::stage(const void* host, size_t payload) {
  const size_t bytes = (std::max<size_t>(payload, 1) + 255) / 256 * 256;
  if (stage_off_ + bytes > stage_cap_) {
    // The ring is full. What it holds may still be in use — tables staged earlier for a launch that has not been queued yet
    // (a scan over 1 024 small records stages its predicate tables first and 1 024 argument blocks after them), or launches that
    // are queued but have not run — so it is neither restarted nor freed here: it is RETIRED (kept until the stream is next known
    // idle, reset_staging) and a bigger one takes over. (Until round 4 the ring restarted / was freed in place: the earlier tables
    // of the same launch were overwritten — or, with several chains in flight, their freed block was handed to another thread's
    // hipMalloc — and the scan read garbage LUTs.)
    flush_staging();  // (pending bytes go to the ring they were staged in)
    if (stage_h_ != nullptr) retired_.emplace_back(stage_h_, stage_d_);
    stage_h_ = nullptr; stage_d_ = nullptr;
    stage_off_ = 0;
    stage_sent_ = 0;
    shadow_valid_ = 0;
    stage_cap_ = std::max<size_t>(std::max<size_t>(round_block(bytes * 2), stage_cap_ * 2), 1 << 20);
    hip_check(pinned_malloc_near((void**)&stage_h_, stage_cap_, device), "hipHostMalloc(staging)");
    hip_check(hipMalloc((void**)&stage_d_, stage_cap_), "hipMalloc(staging)");
  }
  if (payload) std::memcpy(stage_h_ + stage_off_, host, payload);
  void* dst = stage_d_ + stage_off_;
  stage_off_ += bytes;
  if (!defer_) flush_staging();
  return dst;
}

void Context::flush_staging() {
  if (stage_sent_ >= stage_off_) return;
  static const bool skip_known = std::getenv("FDB_NO_STAGE_SKIP") == nullptr;  // (A/B aid)
  const size_t a = stage_sent_, b = stage_off_;
  if (skip_known && b <= shadow_valid_ && std::memcmp(stage_h_ + a, stage_shadow_.data() + a, b - a) == 0) {
    stage_sent_ = b;  // the device ring already holds exactly these bytes (shipped by an earlier flush on this stream)
    return;
  }
  hip_check(hipMemcpyAsync(stage_d_ + a, stage_h_ + a, b - a, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(staging)");
  if (stage_shadow_.size() < stage_cap_) stage_shadow_.resize(stage_cap_);
  if (a <= shadow_valid_) {  // (only a prefix is tracked: flushes start at 0 after every restart of the ring)
    std::memcpy(stage_shadow_.data() + a, stage_h_ + a, b - a);
    shadow_valid_ = std::max(shadow_valid_, b);
  }
  stage_sent_ = b;
}

unsigned char* Context::copy_reserve(size_t payload) {
  const size_t bytes = (std::max<size_t>(payload, 1) + 255) / 256 * 256;
  if (copy_off_ + bytes > copy_cap_) {
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(copy ring)");  // every DMA out of the ring has finished
    copy_off_ = 0;
    if (bytes > copy_cap_) {
      if (copy_h_) (void)hipHostFree(copy_h_);
      copy_cap_ = std::max<size_t>(round_block(bytes * 2), (size_t)32 << 20);
      hip_check(pinned_malloc_near((void**)&copy_h_, copy_cap_, device), "hipHostMalloc(copy ring)");
    }
  }
  unsigned char* p = copy_h_ + copy_off_;
  copy_off_ += bytes;
  return p;
}

void Context::copy_commit(void* dst, const unsigned char* ring_ptr, size_t bytes) {
  if (bytes == 0) return;
  hip_check(hipMemcpyAsync(dst, ring_ptr, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(copy ring)");
}

void Context::copy_in(void* dst, const void* host, size_t payload) {
  if (payload == 0) return;
  unsigned char* p = copy_reserve(payload);
  std::memcpy(p, host, payload);
  copy_commit(dst, p, payload);
}

}  // namespace fdb
