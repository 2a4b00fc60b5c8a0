// fdb_comm.cpp — transports of the cross-GPU merge (see fdb_comm.h) and the plan-level merge operations built on them.
#include "fdb_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: every entry point is resolved with dlsym (no link-time dependency on librccl)

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>

#include "fdb_context.h"
#include "fdb_kernels.h"
#include "fdb_plan_internal.h"

namespace fdb {

namespace {

// Bytes one rank sends to one peer per grouped send/recv round. Measured on MI355X / RCCL 2.26 (round 1): a single exchange
// moving 1.4 GB between two buffers silently delivered only its first ≈0.69 GB, so big partitions travel in slices.
constexpr int64_t kExchangeSliceWordsDefault = (128 << 20) / 8;
// ($FDB_EXCHANGE_SLICE_BYTES: test hook — tests/test_gpu_fake_rccl.py forces 1 MiB slices so that the slicing itself is exercised
// with several ranks; every rank of a communicator must see the same value)
int64_t exchange_slice_words() {
  static const int64_t words = [] {
    const char* e = std::getenv("FDB_EXCHANGE_SLICE_BYTES");
    const long long b = e != nullptr ? std::atoll(e) : 0;
    return b >= 8 ? (int64_t)(b / 8) : kExchangeSliceWordsDefault;
  }();
  return words;
}

// ---- librccl, bound at run time -------------------------------------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  std::string why;  // why loading failed
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;  // optional: what the communicator itself says its size is (Comm::transport_ranks)

  RcclApi() {
    // The copy already mapped into the process wins (a host that also runs torch carries its own librccl; two RCCLs in one
    // process would each keep their own global state), then the system library, then an explicit path.
    const char* env = std::getenv("FDB_RCCL_LIB");
    if (env != nullptr && *env) handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (handle == nullptr) handle = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (handle == nullptr) handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (handle == nullptr) handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (handle == nullptr) handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (handle == nullptr) { const char* e = dlerror(); why = std::string("librccl not found (") + (e ? e : "?") + ")"; return; }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(handle, name);
      if (p == nullptr && why.empty()) why = std::string("librccl lacks ") + name;
      return p;
    };
    GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
    AllGather = (decltype(AllGather))sym("ncclAllGather");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    CommCount = (decltype(CommCount))dlsym(handle, "ncclCommCount");
  }
  bool ok() const { return handle != nullptr && why.empty(); }
};

RcclApi& rccl() {
  static RcclApi api;
  if (!api.ok()) throw Error(FDB_ERR_UNSUPPORTED, "RCCL is not available in this process: " + api.why);
  return api;
}

void nccl_check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw Error(FDB_ERR_DEVICE, std::string(what) + ": " + rccl().GetErrorString(r));
}

ncclRedOp_t red_op(int op) { return op == 3 ? ncclMin : op == 4 ? ncclMax : ncclSum; }
ncclDataType_t red_type(int op) { return op == 2 ? ncclFloat64 : ncclInt64; }

class RcclComm : public Comm {
 public:
  ncclComm_t comm = nullptr;
  hipStream_t ctl = nullptr;            // the communicator's own stream: control collectives run next to the plans' kernels
  unsigned long long* d_ctl = nullptr;  // device scratch of the control collectives
  size_t d_ctl_bytes = 0;
  unsigned long long* h_ctl = nullptr;  // pinned
  size_t h_ctl_bytes = 0;

  ~RcclComm() override {
    (void)hipSetDevice(device);
    if (ctl) (void)hipStreamSynchronize(ctl);
    if (comm) (void)rccl().CommDestroy(comm);
    if (ctl) (void)hipStreamDestroy(ctl);
    if (d_ctl) (void)hipFree(d_ctl);
    if (h_ctl) (void)hipHostFree(h_ctl);
  }
  void setup() {
    hip_check(hipSetDevice(device), "hipSetDevice");
    hip_check(hipStreamCreateWithFlags(&ctl, hipStreamNonBlocking), "hipStreamCreate(comm)");
    reserve(4096, 4096);
  }
  void reserve(size_t dev_bytes, size_t host_bytes) {
    if (dev_bytes > d_ctl_bytes) {
      if (d_ctl) (void)hipFree(d_ctl);
      d_ctl = nullptr;
      hip_check(hipMalloc((void**)&d_ctl, dev_bytes), "hipMalloc(comm scratch)");
      d_ctl_bytes = dev_bytes;
    }
    if (host_bytes > h_ctl_bytes) {
      if (h_ctl) (void)hipHostFree(h_ctl);
      h_ctl = nullptr;
      hip_check(hipHostMalloc((void**)&h_ctl, host_bytes, hipHostMallocDefault), "hipHostMalloc(comm scratch)");
      h_ctl_bytes = host_bytes;
    }
  }

  int transport_ranks() override {
    RcclApi& R = rccl();
    if (R.CommCount == nullptr || comm == nullptr) return -1;
    int n = -1;
    nccl_check(R.CommCount(comm, &n), "ncclCommCount");
    return n;
  }

  void probe_max(int64_t v[4]) override {
    hip_check(hipSetDevice(device), "hipSetDevice");
    std::memcpy(h_ctl, v, 32);
    hip_check(hipMemcpyAsync(d_ctl, h_ctl, 32, hipMemcpyHostToDevice, ctl), "hipMemcpyAsync(probe)");
    nccl_check(rccl().AllReduce(d_ctl, d_ctl, 4, ncclInt64, ncclMax, comm, ctl), "ncclAllReduce(probe)");
    hip_check(hipMemcpyAsync(h_ctl, d_ctl, 32, hipMemcpyDeviceToHost, ctl), "hipMemcpyAsync(probe)");
    hip_check(hipStreamSynchronize(ctl), "hipStreamSynchronize(comm)");
    std::memcpy(v, h_ctl, 32);
  }

  void all_reduce(const std::vector<Red>& reds, hipStream_t stream) override {
    if (reds.empty()) return;
    hip_check(hipSetDevice(device), "hipSetDevice");
    RcclApi& R = rccl();
    nccl_check(R.GroupStart(), "ncclGroupStart");
    ncclResult_t first = ncclSuccess;
    for (const Red& r : reds) {
      const ncclResult_t e = R.AllReduce(r.buf, r.buf, r.count, red_type(r.op), red_op(r.op), comm, stream);
      if (e != ncclSuccess && first == ncclSuccess) first = e;
    }
    const ncclResult_t end = R.GroupEnd();
    nccl_check(first, "ncclAllReduce");
    nccl_check(end, "ncclGroupEnd");
  }

  void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
    if (bytes == 0) return;
    hip_check(hipSetDevice(device), "hipSetDevice");
    nccl_check(rccl().AllGather(send, recv, bytes / 8, ncclUint64, comm, stream), "ncclAllGather");
  }

  std::vector<std::vector<uint8_t>> all_gather_host(const std::vector<uint8_t>& mine) override {
    hip_check(hipSetDevice(device), "hipSetDevice");
    RcclApi& R = rccl();
    // round 1: sizes
    reserve((size_t)size * 8 + 64, (size_t)size * 8 + 64);
    h_ctl[0] = (unsigned long long)mine.size();
    hip_check(hipMemcpyAsync(d_ctl + rank, h_ctl, 8, hipMemcpyHostToDevice, ctl), "hipMemcpyAsync(sizes)");
    nccl_check(R.AllGather(d_ctl + rank, d_ctl, 1, ncclUint64, comm, ctl), "ncclAllGather(sizes)");
    hip_check(hipMemcpyAsync(h_ctl, d_ctl, (size_t)size * 8, hipMemcpyDeviceToHost, ctl), "hipMemcpyAsync(sizes)");
    hip_check(hipStreamSynchronize(ctl), "hipStreamSynchronize(comm)");
    std::vector<size_t> sizes((size_t)size);
    size_t slot = 8;
    for (int p = 0; p < size; p++) { sizes[(size_t)p] = (size_t)h_ctl[p]; slot = std::max(slot, sizes[(size_t)p]); }
    slot = (slot + 7) / 8 * 8;
    // round 2: payloads, padded to the largest
    reserve(slot * (size_t)size, slot * (size_t)size);
    unsigned char* hb = (unsigned char*)h_ctl;
    unsigned char* db = (unsigned char*)d_ctl;
    if (!mine.empty()) std::memcpy(hb, mine.data(), mine.size());
    hip_check(hipMemcpyAsync(db + slot * (size_t)rank, hb, slot, hipMemcpyHostToDevice, ctl), "hipMemcpyAsync(blob)");
    nccl_check(R.AllGather(db + slot * (size_t)rank, db, slot, ncclUint8, comm, ctl), "ncclAllGather(blobs)");
    hip_check(hipMemcpyAsync(hb, db, slot * (size_t)size, hipMemcpyDeviceToHost, ctl), "hipMemcpyAsync(blobs)");
    hip_check(hipStreamSynchronize(ctl), "hipStreamSynchronize(comm)");
    std::vector<std::vector<uint8_t>> out((size_t)size);
    for (int p = 0; p < size; p++) out[(size_t)p].assign(hb + slot * (size_t)p, hb + slot * (size_t)p + sizes[(size_t)p]);
    return out;
  }

  void all_to_all(const unsigned long long* send, unsigned long long* recv, const std::vector<std::vector<int64_t>>& words,
                  hipStream_t stream) override {
    hip_check(hipSetDevice(device), "hipSetDevice");
    RcclApi& R = rccl();
    std::vector<int64_t> send_off((size_t)size, 0), recv_off((size_t)size, 0);
    int64_t max_words = 0;
    for (int p = 0; p < size; p++) {
      if (p > 0) {
        send_off[(size_t)p] = send_off[(size_t)p - 1] + words[(size_t)rank][(size_t)p - 1];
        recv_off[(size_t)p] = recv_off[(size_t)p - 1] + words[(size_t)p - 1][(size_t)rank];
      }
      for (int q = 0; q < size; q++) max_words = std::max(max_words, words[(size_t)p][(size_t)q]);
    }
    // every rank runs the same number of rounds (derived from the matrix all of them hold); a pair with nothing left skips
    const int64_t kExchangeSliceWords = exchange_slice_words();
    const int64_t rounds = (max_words + kExchangeSliceWords - 1) / kExchangeSliceWords;
    for (int64_t k = 0; k < rounds; k++) {
      const int64_t lo = k * kExchangeSliceWords;
      nccl_check(R.GroupStart(), "ncclGroupStart");
      ncclResult_t first = ncclSuccess;
      for (int p = 0; p < size; p++) {
        const int64_t ns = std::max<int64_t>(0, std::min(kExchangeSliceWords, words[(size_t)rank][(size_t)p] - lo));
        const int64_t nr = std::max<int64_t>(0, std::min(kExchangeSliceWords, words[(size_t)p][(size_t)rank] - lo));
        ncclResult_t e = ncclSuccess;
        if (ns > 0) e = R.Send(send + send_off[(size_t)p] + lo, (size_t)ns, ncclUint64, p, comm, stream);
        if (e == ncclSuccess && nr > 0) e = R.Recv(recv + recv_off[(size_t)p] + lo, (size_t)nr, ncclUint64, p, comm, stream);
        if (e != ncclSuccess && first == ncclSuccess) first = e;
      }
      const ncclResult_t end = R.GroupEnd();
      nccl_check(first, "ncclSend/ncclRecv");
      nccl_check(end, "ncclGroupEnd");
    }
  }
};

// ---- local transport: ranks are threads of one process ---------------------------------------------------------------------
struct LocalGroup {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool broken = false;  // a rank failed inside a collective: everybody leaves with an error instead of waiting forever
  std::vector<std::vector<Comm::Red>> reds;            // published per rank
  std::vector<const unsigned long long*> send_ptr;
  std::vector<std::vector<uint8_t>> blobs;
  std::vector<int64_t> probe;                          // [n][4]

  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) throw Error(FDB_ERR_STATE, "local communicator: another rank failed inside a collective");
    const uint64_t gen = generation;
    if (++arrived == n) { arrived = 0; generation++; cv.notify_all(); return; }
    cv.wait(lk, [&] { return generation != gen || broken; });
    if (generation == gen) throw Error(FDB_ERR_STATE, "local communicator: another rank failed inside a collective");
  }
  void fail() {
    std::lock_guard<std::mutex> lk(mu);
    broken = true;
    cv.notify_all();
  }
};

class LocalComm : public Comm {
 public:
  std::shared_ptr<LocalGroup> g;
  hipStream_t ctl = nullptr;

  ~LocalComm() override {
    if (ctl) { (void)hipSetDevice(device); (void)hipStreamDestroy(ctl); }
  }

  template <typename F>
  void guarded(F&& f) {
    try { f(); } catch (...) { g->fail(); throw; }
  }

  std::vector<std::vector<uint8_t>> all_gather_host(const std::vector<uint8_t>& mine) override {
    std::vector<std::vector<uint8_t>> out;
    guarded([&] {
      { std::lock_guard<std::mutex> lk(g->mu); g->blobs[(size_t)rank] = mine; }
      g->barrier();
      { std::lock_guard<std::mutex> lk(g->mu); out = g->blobs; }
      g->barrier();  // nobody overwrites its blob before everyone has copied
    });
    return out;
  }

  void probe_max(int64_t v[4]) override {
    guarded([&] {
      { std::lock_guard<std::mutex> lk(g->mu); for (int i = 0; i < 4; i++) g->probe[(size_t)rank * 4 + i] = v[i]; }
      g->barrier();
      {
        std::lock_guard<std::mutex> lk(g->mu);
        for (int i = 0; i < 4; i++) for (int p = 0; p < g->n; p++) v[i] = std::max(v[i], g->probe[(size_t)p * 4 + i]);
      }
      g->barrier();
    });
  }

  // Reduce-scatter then all-gather over direct loads of the peers' buffers: rank r reduces slice r of every array across all
  // ranks (rank order: deterministic float sums) into its own buffer — in that phase nobody else touches slice r of anything —
  // then copies the other slices from their owners.
  void all_reduce(const std::vector<Red>& reds, hipStream_t stream) override {
    guarded([&] {
      hip_check(hipSetDevice(device), "hipSetDevice");
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");  // this rank's arrays are complete
      { std::lock_guard<std::mutex> lk(g->mu); g->reds[(size_t)rank] = reds; }
      g->barrier();
      std::vector<std::vector<Red>> all;
      { std::lock_guard<std::mutex> lk(g->mu); all = g->reds; }
      for (int p = 0; p < size; p++)
        if (all[(size_t)p].size() != reds.size()) throw Error(FDB_ERR_INVALID, "local communicator: ranks disagree on the number of arrays");
      auto slice = [&](size_t count, int p, size_t* lo, size_t* hi) { *lo = count * (size_t)p / (size_t)size; *hi = count * (size_t)(p + 1) / (size_t)size; };
      for (size_t i = 0; i < reds.size(); i++) {
        const void* srcs[FDB_MAX_PARTS];
        for (int p = 0; p < size; p++) {
          if (all[(size_t)p][i].count != reds[i].count || all[(size_t)p][i].op != reds[i].op) throw Error(FDB_ERR_INVALID, "local communicator: ranks disagree on an array");
          srcs[p] = all[(size_t)p][i].buf;
        }
        size_t lo, hi;
        slice(reds[i].count, rank, &lo, &hi);
        if (hi > lo) hip_check(fdb_launch_peer_reduce((unsigned long long*)reds[i].buf, srcs, size, (int64_t)lo, (int64_t)hi, reds[i].op, stream), "peer reduce");
      }
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
      g->barrier();  // every slice is final at its owner
      for (size_t i = 0; i < reds.size(); i++)
        for (int p = 0; p < size; p++) {
          if (p == rank) continue;
          size_t lo, hi;
          slice(reds[i].count, p, &lo, &hi);
          if (hi > lo)
            hip_check(hipMemcpyAsync((unsigned long long*)reds[i].buf + lo, (const unsigned long long*)all[(size_t)p][i].buf + lo, (hi - lo) * 8, hipMemcpyDefault, stream),
                      "hipMemcpyAsync(peer slice)");
        }
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
      g->barrier();  // nobody's table changes while a peer still reads it
    });
  }

  void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
    if (bytes == 0) return;
    guarded([&] {
      hip_check(hipSetDevice(device), "hipSetDevice");
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");  // this rank's block is complete
      { std::lock_guard<std::mutex> lk(g->mu); g->send_ptr[(size_t)rank] = (const unsigned long long*)send; }
      g->barrier();
      std::vector<const unsigned long long*> ptrs;
      { std::lock_guard<std::mutex> lk(g->mu); ptrs = g->send_ptr; }
      for (int p = 0; p < size; p++)
        hip_check(hipMemcpyAsync((unsigned char*)recv + (size_t)p * bytes, ptrs[(size_t)p], bytes, hipMemcpyDefault, stream), "hipMemcpyAsync(peer block)");
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
      g->barrier();  // nobody's block changes while a peer still reads it
    });
  }

  void all_to_all(const unsigned long long* send, unsigned long long* recv, const std::vector<std::vector<int64_t>>& words,
                  hipStream_t stream) override {
    guarded([&] {
      hip_check(hipSetDevice(device), "hipSetDevice");
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
      { std::lock_guard<std::mutex> lk(g->mu); g->send_ptr[(size_t)rank] = send; }
      g->barrier();
      std::vector<const unsigned long long*> ptrs;
      { std::lock_guard<std::mutex> lk(g->mu); ptrs = g->send_ptr; }
      int64_t recv_off = 0;
      for (int p = 0; p < size; p++) {
        int64_t src_off = 0;
        for (int q = 0; q < rank; q++) src_off += words[(size_t)p][(size_t)q];
        const int64_t n = words[(size_t)p][(size_t)rank];
        if (n > 0) hip_check(hipMemcpyAsync(recv + recv_off, ptrs[(size_t)p] + src_off, (size_t)n * 8, hipMemcpyDefault, stream), "hipMemcpyAsync(peer rows)");
        recv_off += n;
      }
      hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
      g->barrier();  // the senders' buffers may be released now
    });
  }
};

}  // namespace

void rccl_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == FDB_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  nccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
  std::memcpy(id, &u, sizeof(u));
}

std::unique_ptr<Comm> rccl_init_rank(const uint8_t id[128], int n_ranks, int rank, int device) {
  if (n_ranks < 1 || n_ranks > FDB_MAX_PARTS || rank < 0 || rank >= n_ranks) throw Error(FDB_ERR_INVALID, "communicator: rank / size out of range (at most 64 ranks)");
  RcclApi& R = rccl();
  std::unique_ptr<RcclComm> c(new RcclComm());
  c->rank = rank; c->size = n_ranks; c->device = device;
  c->setup();
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  nccl_check(R.CommInitRank(&c->comm, n_ranks, u, rank), "ncclCommInitRank");
  return c;
}

std::vector<std::unique_ptr<Comm>> rccl_init_all(const int* devices, int n) {
  if (n < 1 || n > FDB_MAX_PARTS) throw Error(FDB_ERR_INVALID, "communicator: size out of range (at most 64 ranks)");
  RcclApi& R = rccl();
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  nccl_check(R.CommInitAll(comms.data(), n, devices), "ncclCommInitAll");
  // every raw handle gets its owner BEFORE anything that can fail: if rank r's setup throws, ~RcclComm destroys all n communicators
  std::vector<std::unique_ptr<RcclComm>> owned;
  for (int r = 0; r < n; r++) {
    owned.emplace_back(new RcclComm());
    owned.back()->rank = r; owned.back()->size = n; owned.back()->device = devices[r]; owned.back()->comm = comms[(size_t)r];
  }
  for (auto& c : owned) c->setup();
  std::vector<std::unique_ptr<Comm>> out;
  for (auto& c : owned) out.push_back(std::move(c));
  return out;
}

std::vector<std::unique_ptr<Comm>> local_init(const int* devices, int n) {
  if (n < 1 || n > FDB_MAX_PARTS) throw Error(FDB_ERR_INVALID, "communicator: size out of range (at most 64 ranks)");
  // peers on different devices read each other's memory directly
  for (int a = 0; a < n; a++)
    for (int b = 0; b < n; b++) {
      if (devices[a] == devices[b]) continue;
      int can = 0;
      hip_check(hipDeviceCanAccessPeer(&can, devices[a], devices[b]), "hipDeviceCanAccessPeer");
      if (!can) throw Error(FDB_ERR_UNSUPPORTED, "local communicator: devices " + std::to_string(devices[a]) + " and " + std::to_string(devices[b]) + " have no peer access");
      hip_check(hipSetDevice(devices[a]), "hipSetDevice");
      const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_check(e, "hipDeviceEnablePeerAccess");
      (void)hipGetLastError();
    }
  std::shared_ptr<LocalGroup> g(new LocalGroup());
  g->n = n;
  g->reds.resize((size_t)n); g->send_ptr.resize((size_t)n, nullptr); g->blobs.resize((size_t)n); g->probe.resize((size_t)n * 4, 0);
  std::vector<std::unique_ptr<Comm>> out;
  for (int r = 0; r < n; r++) {
    std::unique_ptr<LocalComm> c(new LocalComm());
    c->rank = r; c->size = n; c->device = devices[r]; c->g = g;
    out.push_back(std::move(c));
  }
  return out;
}

// ---- plan-level merges --------------------------------------------------------------------------------------------------------
namespace {
// Test hook ($FDB_TEST_FAIL_MERGE_RANK = r): rank r fails before the first collective of a merge, the way a pending record that
// raises when it is scanned would — tests/test_gpu_comm.py checks that the OTHER ranks then leave with an error instead of waiting.
void inject_merge_fault(int rank) {
  const char* e = std::getenv("FDB_TEST_FAIL_MERGE_RANK");
  if (e != nullptr && *e && std::atoi(e) == rank) throw Error(FDB_ERR_STATE, "injected failure before the merge (FDB_TEST_FAIL_MERGE_RANK)");
}
}  // namespace

bool Plan::comm_allreduce(Comm& comm) {
  // Whatever can fail on THIS rank before the first collective (a wrong device, a pending record that raises when it is scanned)
  // is caught and VOTED: a rank that threw here while its peers entered the probe would leave them blocked inside RCCL for ever.
  // The vote rides in the probe itself (the signature is 62 bits wide, so INT64_MAX in both of its words is no layout).
  std::exception_ptr local;
  int64_t n_slots = 0;
  int64_t v[4] = {INT64_MAX, INT64_MAX, 0, 0};
  try {
    if (comm.device != device_) throw Error(FDB_ERR_INVALID, "communicator endpoint lives on another device than the plan");
    inject_merge_fault(comm.rank);
    settle();
    runs_to_table();  // (an ordered plan's collected runs become table entries before anything is merged)
    hip_check(hipSetDevice(device_), "hipSetDevice");
    const uint64_t sig = state_signature(&n_slots) & ((1ull << 62) - 1);
    v[0] = (int64_t)sig; v[1] = -(int64_t)sig; v[2] = n_slots; v[3] = -n_slots;
  } catch (...) { local = std::current_exception(); }
  comm.probe_max(v);  // on the communicator's own stream: overlaps the scan still running on ours
  if (local) std::rethrow_exception(local);
  if (v[0] == INT64_MAX && v[1] == INT64_MAX) throw Error(FDB_ERR_STATE, "all-reduce merge abandoned: another rank failed before the collective");
  if (v[0] != -v[1] || v[2] != -v[3] || v[2] == 0) return false;
  materialize_state();  // (a rank that scanned nothing still holds an unfilled table)
  mirror_valid_ = false;
  std::vector<Comm::Red> reds;
  for (int32_t a = 0; a < num_state_arrays(); a++) {
    const int32_t op = state_array_op(a);
    if (op == 0) continue;  // COUNT is served by the row-count array
    reds.push_back(Comm::Red{d_state_ + (size_t)a * slots_alloc_, (size_t)n_slots_, op});
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
  // Small tables (cfg 2 / 3 / 4: 1 025 slots × a few arrays): ONE all-gather of the packed table and a local fold in rank order, which
  // also writes the host copy — one collective on the step's critical path instead of one all-reduce per array, and float64 sums that
  // are bit-identical on every rank and in every run whatever order the ranks arrived in (what fdb_plan_set_deterministic
  // promises for the scan now holds across GPUs too). Bigger tables keep the all-reduce. ($FDB_MERGE_ALLREDUCE: A/B aid)
  const int n_arrays = num_state_arrays();
  const size_t packed_bytes = (size_t)n_arrays * (size_t)n_slots_ * 8;
  if (packed_bytes * (size_t)comm.size <= ((size_t)32 << 20) && std::getenv("FDB_MERGE_ALLREDUCE") == nullptr) {
    unsigned long long* d_packed = (unsigned long long*)ctx_->dev_alloc(packed_bytes + 256);
    unsigned long long* d_all = (unsigned long long*)ctx_->dev_alloc(packed_bytes * (size_t)comm.size + 256);
    scratch_.push_back(d_packed); scratch_.push_back(d_all);
    int32_t ops[1 + FDB_MAX_AGGS] = {0};
    for (int32_t a = 0; a < n_arrays; a++) ops[a] = state_array_op(a);
    hip_check(fdb_launch_state_pack(d_state_, d_packed, n_slots_, slots_alloc_, n_arrays, stream_), "state pack");
    comm.all_gather(d_packed, d_all, packed_bytes, stream_);
    unsigned long long* host_out = mirror_target();
    hip_check(fdb_launch_state_fold_ranks(d_all, comm.size, n_slots_, n_arrays, ops, d_state_, slots_alloc_, host_out, stream_), "state fold (rank order)");
    if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); merge_events_.emplace_back(e0, e1); }
    state_dirty_ = true;
    mirror_valid_ = host_out != nullptr;
    return true;
  }
  comm.all_reduce(reds, stream_);  // ordered after the scan and the fold kernel; Finish / Close wait for this stream
  if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); merge_events_.emplace_back(e0, e1); }
  state_dirty_ = true;
  // the merged table goes to the host copy from a kernel on the same stream (Finish then waits once and reads it; a device→host
  // copy command behind the collective was 15 µs)
  if (unsigned long long* host_out = mirror_target()) {
    hip_check(fdb_launch_state_to_host(d_state_, host_out, n_slots_, slots_alloc_, (int)(1 + aggs_.size()), stream_), "state to host");
    mirror_valid_ = true;
  }
  return true;
}

namespace {
void put_u32(std::vector<uint8_t>* b, uint32_t v) { const size_t o = b->size(); b->resize(o + 4); std::memcpy(b->data() + o, &v, 4); }
void put_str(std::vector<uint8_t>* b, const char* p, size_t n) { put_u32(b, (uint32_t)n); b->insert(b->end(), (const uint8_t*)p, (const uint8_t*)p + n); }
struct Reader {
  const std::vector<uint8_t>& b;
  size_t at = 0;
  uint32_t u32() {
    if (at + 4 > b.size()) throw Error(FDB_ERR_INVALID, "group schema message truncated");
    uint32_t v; std::memcpy(&v, b.data() + at, 4); at += 4; return v;
  }
  std::string str() {
    const uint32_t n = u32();
    if (at + n > b.size()) throw Error(FDB_ERR_INVALID, "group schema message truncated");
    std::string s((const char*)b.data() + at, n); at += n; return s;
  }
};
}  // namespace

GroupSchema Plan::export_schema() const {
  GroupSchema s;
  for (const GroupColState& g : gcols_) {
    GroupSchemaCol c;
    c.name = g.name; c.kind = g.kind; c.is_bool = g.is_bool; c.is_u64 = g.is_u64; c.plain = g.plain; c.value_format = g.value_format;
    for (const std::string_view& v : g.values) c.values.emplace_back(v);
    s.cols.push_back(std::move(c));
  }
  for (const AggState& a : aggs_) s.agg_types.push_back(a.type);
  return s;
}

void Plan::adopt_schema(const GroupSchema& s) {
  runs_to_table();
  if (mode_ == TableMode::DENSE) switch_to_hash();  // (before the column set changes)
  if (s.agg_types.size() != aggs_.size()) throw Error(FDB_ERR_INVALID, "plans have different aggregations");
  for (size_t j = 0; j < aggs_.size(); j++) {
    if (s.agg_types[j] == FDB_T_NONE) continue;
    if (aggs_[j].type != FDB_T_NONE && aggs_[j].type != s.agg_types[j]) throw Error(FDB_ERR_INVALID, "aggregation types differ between plans");
    aggs_[j].type = s.agg_types[j];
  }
  for (const GroupSchemaCol& c : s.cols) {
    size_t gi = 0;
    for (; gi < gcols_.size(); gi++) if (gcols_[gi].name == c.name) break;
    if (gi == gcols_.size()) {
      GroupColState g;
      g.name = c.name; g.kind = c.kind; g.is_bool = c.is_bool; g.is_u64 = c.is_u64; g.plain = c.plain; g.value_format = c.value_format; g.cap = 1; g.stride = 0;
      gcols_.push_back(std::move(g));
    }
    GroupColState& g = gcols_[gi];
    if (g.kind != c.kind || g.is_bool != c.is_bool || g.is_u64 != c.is_u64 || g.plain != c.plain)
      throw Error(FDB_ERR_INVALID, "group column " + c.name + " has a different type in this plan");
    if (c.kind == 0 && !c.values.empty()) {
      std::shared_ptr<HostDict> d(new HostDict());
      d->values = c.values; d->value_format = c.value_format; d->plain = c.plain;
      g.owners.push_back(d);
      for (const std::string& v : d->values) g.intern(std::string_view(v));
    }
  }
  if (gcols_.size() > FDB_MAX_HASH_GCOLS) throw Error(FDB_ERR_UNSUPPORTED, "too many group columns");
  hash_layout();
  hash_reserve(0);
}

void Plan::comm_exchange(Comm& comm, Plan& shard) {
  PhaseTimer pt;
  // 1. one group schema for all ranks: columns and dictionary values in first-seen order, ranks in rank order (the
  //    Synchronizer's arrival order made deterministic); key ids assigned from it mean the same group everywhere.
  //    A rank that fails before this first collective still takes part in it — with an EMPTY message, which makes every rank
  //    leave with an error instead of waiting inside RCCL for a peer that is gone.
  std::vector<uint8_t> blob;
  std::exception_ptr local;
  try {
    if (comm.device != device_ || shard.device_ != device_) throw Error(FDB_ERR_INVALID, "communicator endpoint lives on another device than the plan");
    inject_merge_fault(comm.rank);
    settle();
    runs_to_table();  // (an ordered plan's collected runs become table entries before anything is merged)
    hip_check(hipSetDevice(device_), "hipSetDevice");
    const GroupSchema mine = export_schema();
    put_u32(&blob, (uint32_t)mine.cols.size());
    for (const GroupSchemaCol& c : mine.cols) {
      put_str(&blob, c.name.data(), c.name.size());
      put_u32(&blob, (uint32_t)c.kind | (c.is_bool ? 0x100u : 0u) | (c.is_u64 ? 0x200u : 0u) | (c.plain ? 0x400u : 0u));
      put_str(&blob, c.value_format.data(), c.value_format.size());
      put_u32(&blob, (uint32_t)c.values.size());
      for (const std::string& v : c.values) put_str(&blob, v.data(), v.size());
    }
    put_u32(&blob, (uint32_t)mine.agg_types.size());
    for (int32_t t : mine.agg_types) put_u32(&blob, (uint32_t)t);
  } catch (...) { local = std::current_exception(); blob.clear(); }
  const std::vector<std::vector<uint8_t>> all = comm.all_gather_host(blob);
  if (local) std::rethrow_exception(local);
  for (const std::vector<uint8_t>& b : all)
    if (b.empty()) throw Error(FDB_ERR_STATE, "exchange abandoned: another rank failed before the collective");
  GroupSchema uni;
  uni.agg_types.assign(aggs_.size(), FDB_T_NONE);
  std::vector<std::unordered_map<std::string, uint32_t>> seen;
  for (const std::vector<uint8_t>& b : all) {
    Reader r{b};
    const uint32_t n_cols = r.u32();
    for (uint32_t k = 0; k < n_cols; k++) {
      GroupSchemaCol c;
      c.name = r.str();
      const uint32_t flags = r.u32();
      c.kind = (int)(flags & 0xFF); c.is_bool = (flags & 0x100u) != 0; c.is_u64 = (flags & 0x200u) != 0; c.plain = (flags & 0x400u) != 0;
      c.value_format = r.str();
      const uint32_t n_vals = r.u32();
      size_t ui = 0;
      for (; ui < uni.cols.size(); ui++) if (uni.cols[ui].name == c.name) break;
      if (ui == uni.cols.size()) { uni.cols.push_back(c); seen.emplace_back(); }
      GroupSchemaCol& u = uni.cols[ui];
      if (u.kind != c.kind || u.is_bool != c.is_bool || u.is_u64 != c.is_u64 || u.plain != c.plain)
        throw Error(FDB_ERR_INVALID, "group column " + c.name + " has different types on different ranks");
      for (uint32_t v = 0; v < n_vals; v++) {
        std::string s = r.str();
        if (seen[ui].emplace(s, 0).second) u.values.push_back(std::move(s));
      }
    }
    const uint32_t n_aggs = r.u32();
    if (n_aggs != aggs_.size()) throw Error(FDB_ERR_INVALID, "ranks have different aggregations");
    for (uint32_t j = 0; j < n_aggs; j++) {
      const int32_t t = (int32_t)r.u32();
      if (t == FDB_T_NONE) continue;
      if (uni.agg_types[j] != FDB_T_NONE && uni.agg_types[j] != t) throw Error(FDB_ERR_INVALID, "aggregation types differ between ranks");
      uni.agg_types[j] = t;
    }
  }
  // 2. re-key + partition on the device
  void* rows = nullptr;
  int64_t counts[FDB_MAX_PARTS] = {0};
  int32_t rw = 0;
  std::vector<uint8_t> cb((size_t)comm.size * 8 + 4);
  try {
    shard.adopt_schema(uni);
    for (size_t j = 0; j < aggs_.size(); j++) if (aggs_[j].type == FDB_T_NONE) aggs_[j].type = uni.agg_types[j];
    pt.mark("exchange: schema");
    hash_export(shard, comm.size, &rows, counts, &rw);  // synchronised: the rows are complete
    pt.mark("exchange: export");
    std::memcpy(cb.data(), &rw, 4);
    std::memcpy(cb.data() + 4, counts, (size_t)comm.size * 8);
  } catch (...) { local = std::current_exception(); cb.clear(); }  // (a message of the wrong size: every rank refuses it below)
  // 3. who sends how much to whom
  const std::vector<std::vector<uint8_t>> call = comm.all_gather_host(cb);
  if (local) std::rethrow_exception(local);
  for (int p = 0; p < comm.size; p++)
    if (call[(size_t)p].empty()) throw Error(FDB_ERR_STATE, "exchange abandoned: another rank failed while exporting its table");
  std::vector<std::vector<int64_t>> words((size_t)comm.size, std::vector<int64_t>((size_t)comm.size, 0));
  int64_t recv_rows = 0;
  for (int p = 0; p < comm.size; p++) {
    if (call[(size_t)p].size() != cb.size()) throw Error(FDB_ERR_INVALID, "exchange: malformed counts message");
    int32_t prw; std::memcpy(&prw, call[(size_t)p].data(), 4);
    if (prw != rw) throw Error(FDB_ERR_INVALID, "exchange: ranks disagree on the packed row size");
    for (int q = 0; q < comm.size; q++) {
      int64_t c; std::memcpy(&c, call[(size_t)p].data() + 4 + (size_t)q * 8, 8);
      words[(size_t)p][(size_t)q] = c * (rw / 2);
    }
    recv_rows += words[(size_t)p][(size_t)comm.rank] / (rw / 2);
  }
  // 4. partitions travel to their owners, owners merge on the device
  unsigned long long* recv = nullptr;
  if (recv_rows > 0) { recv = (unsigned long long*)shard.ctx_->dev_alloc((size_t)recv_rows * rw * 4); shard.scratch_.push_back(recv); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timing) { e0 = ctx_->get_event(); e1 = ctx_->get_event(); hip_check(hipEventRecord(e0, stream_), "hipEventRecord"); }
  comm.all_to_all((const unsigned long long*)rows, recv, words, stream_);
  if (timing) { hip_check(hipEventRecord(e1, stream_), "hipEventRecord"); merge_events_.emplace_back(e0, e1); }
  sync();  // the rows have arrived (and this plan's timing events / scratch are settled)
  pt.mark("exchange: all-to-all");
  // The owners merge what they received RANK BY RANK (the regions of `recv` are in rank order): every launch holds a group at most once, and
  // a group's partial sums are added in rank order — the merged float64 sums are a function of the ranks' partial sums and the
  // communicator alone, whatever order the rows arrived in ("float sums merged in fixed rank order"; one launch over all
  // rows added them with atomics in whatever order the waves ran).
  {
    const unsigned long long* at = recv;
    for (int p = 0; p < comm.size; p++) {
      const int64_t w = words[(size_t)p][(size_t)comm.rank];
      if (w > 0) shard.hash_import(at, w / (rw / 2), /*unique_rows=*/true);
      at += w;
    }
  }
  pt.mark("exchange: import");
}

}  // namespace fdb
