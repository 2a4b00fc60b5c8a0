// fdb_context.h — per-device execution contexts recycled across plans.
//
// A query creates and closes one operator chain per scan; creating a HIP stream, pinned staging memory and
// device scratch for each would cost more than the scan itself (a 100 M-row pass is ≈0.3 ms). A Context owns
// those resources; plans borrow one for their lifetime and hand it back on Close. Device and pinned blocks
// are cached by size (no hipFree/hipHostFree on the hot path — both synchronise the device).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <utility>
#include <vector>

namespace fdb {

class Context {
 public:
  static Context* acquire(int device);
  static void release(Context* ctx);  // the caller has synchronised ctx->stream

  int device = 0;
  hipStream_t stream = nullptr;

  void* dev_alloc(size_t bytes);
  void dev_free(void* p);
  void* host_alloc(size_t bytes);  // pinned
  void host_free(void* p);
  hipEvent_t get_event();
  void put_event(hipEvent_t e);
  // A big device→host copy split over `stream` and up to 3 auxiliary streams (several DMA engines work on one PCIe link more
  // evenly than one: 1.35 GB went from 41 to ≈52 GB/s); ordered after everything queued on `stream`, complete when this returns.
  void copy_out_parallel(void* host, const void* dev, size_t bytes);
  hipStream_t aux_stream(int i);  // a second queue of this context (copies that overlap kernels on `stream`); created on first use
  // filter()'s one-pass kernel (FdbSelectArgs, fdb_kernels.h): the control block — ticket counter, error word, one status word per
  // tile — lives as long as the context and is cleared only when it is created or grown (or its 24-bit epoch wraps): status words
  // carry the epoch of the launch that wrote them. Returns the block (≥ FDB_SELECT_CTL_WORDS + `words` words) and starts a new epoch;
  // `*ticket_base` / `*arrival_base` = what the ticket and arrival counters hold now; the caller reports what its launch draws with select_ctl_drawn().
  // The stream is idle between the calls of one filter() (it ends synchronised), which is what makes growing safe.
  unsigned long long* select_ctl(size_t words, uint32_t* epoch, unsigned long long* ticket_base, unsigned long long* arrival_base);
  void select_ctl_drawn(unsigned long long tickets, unsigned long long arrivals) { select_ticket_ += tickets; select_arrival_ += arrivals; }
  void select_ctl_reset() { select_epoch_ = (1u << 24); }  // a launch gave up: the counters are not what the host thinks — the next select_ctl() clears the block

  // Small host→device tables (LUTs, slot maps): staged in pinned memory, shipped with one async copy each.
  void* stage(const void* host, size_t bytes);
  void reset_staging() {  // the stream is idle (tables staged but not yet shipped — a deferred batch — stay where they are)
    if (stage_sent_ == stage_off_) stage_off_ = stage_sent_ = 0;
    copy_off_ = 0;
    if (!retired_.empty()) free_retired();  // rings that filled up while something could still read them (stage())
  }
  // Several tables that feed ONE launch: between defer_staging(true) and flush_staging() stage() only fills the pinned ring
  // (the device address it returns is final), and flush_staging() ships everything staged since with one copy command —
  // each small copy is a few µs on the stream's critical path in front of the kernel.
  void defer_staging(bool on) { if (!on) flush_staging(); defer_ = on; }
  void flush_staging();
  // Copies `bytes` of pageable host memory to `dst` (device) through the pinned ring: the source is fully read when this
  // returns (the caller may free it), the DMA runs asynchronously on `stream`.
  void copy_in(void* dst, const void* host, size_t bytes);
  // The same in two steps, for several host buffers that go to ONE device block: reserve ring space, fill it, commit one DMA.
  unsigned char* copy_reserve(size_t bytes);
  void copy_commit(void* dst, const unsigned char* ring_ptr, size_t bytes);

 private:
  Context() = default;
  struct Block { void* p; size_t bytes; bool used; };
  std::vector<Block> dev_blocks_, host_blocks_;
  std::vector<hipEvent_t> events_;
  unsigned char* stage_h_ = nullptr;
  unsigned char* stage_d_ = nullptr;
  std::vector<std::pair<unsigned char*, unsigned char*>> retired_;  // (pinned, device) halves of rings that filled up: freed when the stream is idle
  void free_retired();
  size_t stage_cap_ = 0, stage_off_ = 0, stage_sent_ = 0;  // [stage_sent_, stage_off_) is staged but not shipped yet
  // What the device half of the staging ring holds, as far as it is known ([0, shadow_valid_)): a flush whose bytes are already
  // there is not shipped again. A query repeated over resident data (same predicate tables, same launch descriptors, the same
  // pooled blocks) stages the same bytes at the same ring offsets every time — the ring restarts whenever the stream is idle —
  // and the copy command in front of the scan kernel was ≈10 µs of a 125 M-row shard's 360 µs step.
  std::vector<unsigned char> stage_shadow_;
  size_t shadow_valid_ = 0;
  bool defer_ = false;
  hipStream_t aux_[3] = {nullptr, nullptr, nullptr};
  unsigned long long* select_ctl_ = nullptr;
  size_t select_words_ = 0;
  uint32_t select_epoch_ = 0;
  unsigned long long select_ticket_ = 0, select_arrival_ = 0;
  unsigned char* copy_h_ = nullptr;  // pinned ring of copy_in (separate from the LUT staging ring: a wrap here never touches staged LUTs)
  size_t copy_cap_ = 0, copy_off_ = 0;
};

// Process-wide, thread-safe pool of pinned host blocks for RESULT records (their Arrow release callback can run on any
// thread, long after the plan and its Context are gone). Blocks are cached by size up to a byte budget; pinning 1 GB
// costs ≈0.2 s, re-using a cached block nothing.
void* pinned_pool_alloc(size_t bytes);
void pinned_pool_free(void* p);

// Process-wide, thread-safe pool of DEVICE blocks for records that outlive any plan (resident batches, resident filter results):
// hipMalloc / hipFree cost hundreds of microseconds and hipFree synchronises the whole device, which a chain that filters one
// record after another must not pay per record. Blocks are 2 MiB granules, re-used when an idle one is at most 25 % larger than
// the request; idle bytes per device are capped (the largest idle blocks go back to the driver first).
void* device_pool_alloc(int device, size_t bytes);
void device_pool_free(int device, void* p);

// Live-allocation accounting (fdb_live_allocations): device blocks handed out by dev_alloc and not returned, arenas of
// resident batches (note_device_alloc / _free), result blocks of the pinned pool not yet released.
void note_device_alloc(size_t bytes);
void note_device_free(size_t bytes);
void live_allocations(int64_t* device_blocks, int64_t* device_bytes, int64_t* pinned_blocks);

}  // namespace fdb
