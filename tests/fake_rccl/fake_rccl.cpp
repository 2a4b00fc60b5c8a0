// fake_rccl.cpp — TEST INFRASTRUCTURE, not product: a stand-in for librccl that accepts SEVERAL ranks of one communicator on ONE
// device, so that the RCCL transport of frostdb_amd/csrc/fdb_comm.cpp (unique id → ncclCommInitRank / ncclCommInitAll, grouped
// in-place all-reduces, all-gathers, grouped and sliced ncclSend / ncclRecv, the failure vote) can be driven with 2 / 4 / 8 ranks on
// the 1-GPU box of the test pool. The real RCCL refuses that ("invalid usage": two ranks on one device), so before this file the
// transport had only ever run with one rank.
//
// It exports the entry points fdb_comm.cpp binds (fdb_comm.cpp: RcclApi) with RCCL's signatures and RCCL's semantics where the
// caller can observe them: operations are ordered behind the work already queued on their stream, operations between
// ncclGroupStart and ncclGroupEnd are posted together (a rank may post its sends and receives to all peers in one group without
// deadlocking), reductions run in rank order, ranks may be threads of one process or separate processes. Transport: the ranks meet
// in a memory-mapped file named after the unique id (under $TMPDIR: no /dev/shm size limit); payloads are staged through it
// (device → file mapping → device). Collectives use one staging slot per rank and two barriers per round; sends and receives use
// one mailbox per ordered pair of ranks, so a rank with nothing to exchange in a round does not have to show up (like RCCL, an
// empty group is a no-op) — a message must fit its mailbox (the test forces fdb_comm.cpp's slice size down with
// FDB_EXCHANGE_SLICE_BYTES).
// Loaded through $FDB_RCCL_LIB by tests/test_gpu_fake_rccl.py. Built by tests/fake_rccl/build.py (g++, links libamdhip64).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types, enums and prototypes: the definitions below must match them

namespace {

constexpr uint32_t kMagic = 0xFDB0CC1u;
constexpr int kMaxOps = 256;
constexpr double kTimeoutSeconds = 90.0;

enum OpKind : uint32_t { OP_ALLREDUCE = 1, OP_ALLGATHER = 2, OP_SEND = 3, OP_RECV = 4 };

struct OpDesc {            // published by a rank for one round
  uint32_t kind, peer, dtype, red;
  uint64_t count, payload_off, payload_bytes;
};
struct RankSlot {          // in shared memory, one per rank, followed by its payload area
  uint32_t n_ops, failed;
  uint32_t pad[14];
  OpDesc ops[kMaxOps];
};
struct Mailbox {           // one per ordered pair (source, destination), followed by its payload area
  std::atomic<uint64_t> posted, consumed;  // messages written by the source / read by the destination
  uint64_t bytes;
  uint64_t pad[5];
};
struct Header {            // start of the segment
  std::atomic<uint32_t> magic, attached, arrived, generation, detached, broken;
  uint32_t n_ranks, pad;
  uint64_t slot_stride, payload_bytes, mailbox_stride, mailbox_bytes, mailboxes_at;
};

size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

template <typename T>
void reduce_into(T* acc, const T* in, size_t n, ncclRedOp_t op) {
  for (size_t i = 0; i < n; i++) acc[i] = op == ncclSum ? (T)(acc[i] + in[i]) : op == ncclMin ? (in[i] < acc[i] ? in[i] : acc[i]) : op == ncclMax ? (in[i] > acc[i] ? in[i] : acc[i]) : acc[i];
}
bool reduce_bytes(void* acc, const void* in, size_t count, ncclDataType_t t, ncclRedOp_t op) {
  if (op != ncclSum && op != ncclMin && op != ncclMax) return false;
  switch (t) {
    case ncclInt64: reduce_into((int64_t*)acc, (const int64_t*)in, count, op); return true;
    case ncclUint64: reduce_into((uint64_t*)acc, (const uint64_t*)in, count, op); return true;
    case ncclFloat64: reduce_into((double*)acc, (const double*)in, count, op); return true;
    case ncclInt32: reduce_into((int32_t*)acc, (const int32_t*)in, count, op); return true;
    case ncclUint32: reduce_into((uint32_t*)acc, (const uint32_t*)in, count, op); return true;
    case ncclFloat32: reduce_into((float*)acc, (const float*)in, count, op); return true;
    case ncclUint8: reduce_into((uint8_t*)acc, (const uint8_t*)in, count, op); return true;
    default: return false;
  }
}

struct Pending {  // an operation posted inside a group (or alone), executed at the group's end
  OpKind kind;
  const void* send;
  void* recv;
  size_t count;
  ncclDataType_t dtype;
  ncclRedOp_t red;
  int peer;
  struct ncclComm* comm;
  hipStream_t stream;
};

thread_local int g_group_depth = 0;
thread_local std::vector<Pending> g_pending;
thread_local std::string g_error;

}  // namespace

struct ncclComm {
  int rank = 0, n = 1, device = 0;
  Header* hdr = nullptr;
  size_t map_bytes = 0;
  std::string name;
  RankSlot* slot(int r) const { return (RankSlot*)((char*)hdr + 4096 + (size_t)r * hdr->slot_stride); }
  unsigned char* payload(int r) const { return (unsigned char*)slot(r) + sizeof(RankSlot); }
  Mailbox* mailbox(int src, int dst) const { return (Mailbox*)((char*)hdr + hdr->mailboxes_at + ((size_t)src * (size_t)n + (size_t)dst) * hdr->mailbox_stride); }
};

namespace {

ncclResult_t fail(ncclResult_t r, const std::string& why) { g_error = why; return r; }

std::string segment_name(const ncclUniqueId& id) {
  char buf[64];
  const unsigned char* b = (const unsigned char*)id.internal;
  std::snprintf(buf, sizeof buf, "/fdb_fake_rccl_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11]);
  const char* dir = std::getenv("TMPDIR");
  return std::string(dir != nullptr && *dir ? dir : "/tmp") + buf;
}

size_t payload_bytes_per_rank() {
  const char* e = std::getenv("FAKE_RCCL_SLOT_MB");
  const size_t mb = e != nullptr && std::atoi(e) > 0 ? (size_t)std::atoi(e) : 8;
  return mb << 20;
}
size_t mailbox_payload_bytes() {
  const char* e = std::getenv("FAKE_RCCL_MAILBOX_KB");
  const size_t kb = e != nullptr && std::atoi(e) > 0 ? (size_t)std::atoi(e) : 1088;
  return kb << 10;
}

template <typename F>
bool spin(F&& done) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; !done(); i++) {
    if (i < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((i & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds) return false;
  }
  return true;
}

// sense-reversing barrier over the segment; false = a peer never arrived (or somebody marked the communicator broken)
bool barrier(ncclComm* c) {
  Header* h = c->hdr;
  if (h->broken.load()) return false;
  const uint32_t gen = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == h->n_ranks) { h->arrived.store(0); h->generation.fetch_add(1); return true; }
  const bool ok = spin([&] { return h->generation.load() != gen || h->broken.load() != 0; });
  if (!ok) h->broken.store(1);
  return ok && h->generation.load() != gen;
}

ncclResult_t attach(ncclComm* c, const ncclUniqueId& id, int n, int rank, bool create_only) {
  c->name = segment_name(id);
  c->n = n; c->rank = rank;
  const size_t payload = payload_bytes_per_rank();
  const size_t stride = (sizeof(RankSlot) + payload + 4095) & ~(size_t)4095;
  const size_t mbox = mailbox_payload_bytes();
  const size_t mbox_stride = (sizeof(Mailbox) + mbox + 4095) & ~(size_t)4095;
  const size_t mailboxes_at = 4096 + stride * (size_t)n;
  c->map_bytes = mailboxes_at + mbox_stride * (size_t)n * (size_t)n;
  int fd = open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  bool creator = fd >= 0;
  if (!creator) {
    if (create_only) return fail(ncclSystemError, "fake rccl: segment " + c->name + " exists already");
    fd = open(c->name.c_str(), O_RDWR, 0600);
    if (fd < 0) return fail(ncclSystemError, "fake rccl: cannot open " + c->name);
  }
  if (creator && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); unlink(c->name.c_str()); return fail(ncclSystemError, "fake rccl: ftruncate failed"); }
  if (!creator) {  // the creator may not have sized the segment yet
    struct stat st;
    const bool sized = spin([&] { return fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes; });
    if (!sized) { close(fd); return fail(ncclSystemError, "fake rccl: segment never reached its size (ranks disagree on the communicator size?)"); }
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail(ncclSystemError, "fake rccl: mmap failed");
  c->hdr = (Header*)p;
  if (creator) {
    c->hdr->n_ranks = (uint32_t)n; c->hdr->slot_stride = stride; c->hdr->payload_bytes = payload;
    c->hdr->mailbox_stride = mbox_stride; c->hdr->mailbox_bytes = mbox; c->hdr->mailboxes_at = mailboxes_at;
    c->hdr->magic.store(kMagic);
  } else if (!spin([&] { return c->hdr->magic.load() == kMagic; })) {
    return fail(ncclSystemError, "fake rccl: segment never initialised");
  }
  if (c->hdr->n_ranks != (uint32_t)n) return fail(ncclInvalidArgument, "fake rccl: ranks disagree on the communicator size");
  c->hdr->attached.fetch_add(1);
  return ncclSuccess;
}

// Sends and receives of one group: every send is written to its mailbox first (the mailbox is free once the peer has consumed
// the previous message — which it did in an EARLIER group of its own, so nobody waits in a cycle), then every receive is read.
ncclResult_t run_p2p(ncclComm* c, std::vector<Pending>& ops) {
  (void)hipSetDevice(c->device);
  for (const Pending& p : ops) if (hipStreamSynchronize(p.stream) != hipSuccess) return fail(ncclUnhandledCudaError, "fake rccl: hipStreamSynchronize failed");
  for (const Pending& p : ops) {
    if (p.peer < 0 || p.peer >= c->n) return fail(ncclInvalidArgument, "fake rccl: peer out of range");
    if (p.kind != OP_SEND) continue;
    const size_t bytes = p.count * dtype_size(p.dtype);
    if (bytes > c->hdr->mailbox_bytes) return fail(ncclInternalError, "fake rccl: a message exceeds the mailbox (FAKE_RCCL_MAILBOX_KB; lower FDB_EXCHANGE_SLICE_BYTES)");
    Mailbox* m = c->mailbox(c->rank, p.peer);
    if (!spin([&] { return m->consumed.load() == m->posted.load() || c->hdr->broken.load() != 0; }) || c->hdr->broken.load()) {
      c->hdr->broken.store(1);
      return fail(ncclSystemError, "fake rccl: the peer never consumed the previous message (timeout)");
    }
    if (bytes > 0 && hipMemcpy((unsigned char*)(m + 1), p.send, bytes, hipMemcpyDefault) != hipSuccess) return fail(ncclUnhandledCudaError, "fake rccl: device → mailbox copy failed");
    m->bytes = bytes;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    m->posted.fetch_add(1);
  }
  for (const Pending& p : ops) {
    if (p.kind != OP_RECV) continue;
    const size_t bytes = p.count * dtype_size(p.dtype);
    Mailbox* m = c->mailbox(p.peer, c->rank);
    if (!spin([&] { return m->posted.load() > m->consumed.load() || c->hdr->broken.load() != 0; }) || c->hdr->broken.load()) {
      c->hdr->broken.store(1);
      return fail(ncclSystemError, "fake rccl: the matching send never arrived (timeout)");
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (m->bytes != bytes) { c->hdr->broken.store(1); return fail(ncclInvalidUsage, "fake rccl: send and receive sizes differ"); }
    if (bytes > 0 && hipMemcpy(p.recv, (const unsigned char*)(m + 1), bytes, hipMemcpyDefault) != hipSuccess) return fail(ncclUnhandledCudaError, "fake rccl: mailbox → device copy failed");
    m->consumed.fetch_add(1);
  }
  return ncclSuccess;
}

// One round: everything this thread posted since the outermost ncclGroupStart (or one lone operation).
ncclResult_t run_round(std::vector<Pending>& ops) {
  if (ops.empty()) return ncclSuccess;
  {
    bool any_p2p = false, any_coll = false;
    for (const Pending& p : ops) { if (p.kind == OP_SEND || p.kind == OP_RECV) any_p2p = true; else any_coll = true; }
    if (any_p2p && any_coll) return fail(ncclInvalidUsage, "fake rccl: collectives and sends / receives in one group are not supported");
    if (any_p2p) {
      for (const Pending& p : ops) if (p.comm != ops[0].comm) return fail(ncclInvalidUsage, "fake rccl: one group spanning several communicators is not supported");
      return run_p2p(ops[0].comm, ops);
    }
  }
  // rounds are per communicator; operations of one group all belong to one communicator here (fdb_comm.cpp never mixes them)
  ncclComm* c = ops[0].comm;
  for (const Pending& p : ops) if (p.comm != c) return fail(ncclInvalidUsage, "fake rccl: one group spanning several communicators is not supported");
  if ((int)ops.size() > kMaxOps) return fail(ncclInternalError, "fake rccl: more than 256 operations in one group");
  (void)hipSetDevice(c->device);
  // stream order: everything queued before these operations has run
  std::vector<hipStream_t> seen;
  for (const Pending& p : ops) {
    bool dup = false;
    for (hipStream_t s : seen) dup = dup || s == p.stream;
    if (!dup) { seen.push_back(p.stream); if (hipStreamSynchronize(p.stream) != hipSuccess) return fail(ncclUnhandledCudaError, "fake rccl: hipStreamSynchronize failed"); }
  }
  // publish descriptors + payloads
  RankSlot* mine = c->slot(c->rank);
  unsigned char* pay = c->payload(c->rank);
  size_t off = 0;
  mine->failed = 0;
  for (size_t i = 0; i < ops.size(); i++) {
    const Pending& p = ops[i];
    OpDesc& d = mine->ops[i];
    d.kind = p.kind; d.peer = (uint32_t)p.peer; d.dtype = (uint32_t)p.dtype; d.red = (uint32_t)p.red; d.count = p.count;
    d.payload_off = off; d.payload_bytes = 0;
    const size_t es = dtype_size(p.dtype);
    if (es == 0) { mine->failed = 1; g_error = "fake rccl: unsupported data type"; continue; }
    if (p.kind != OP_RECV) {
      const size_t bytes = p.count * es;
      if (off + bytes > c->hdr->payload_bytes) { mine->failed = 1; g_error = "fake rccl: a round's payload exceeds the staging slot (FAKE_RCCL_SLOT_MB)"; continue; }
      if (bytes > 0 && hipMemcpy(pay + off, p.send, bytes, hipMemcpyDefault) != hipSuccess) { mine->failed = 1; g_error = "fake rccl: device → staging copy failed"; continue; }
      d.payload_bytes = bytes;
      off += (bytes + 15) & ~(size_t)15;
    }
  }
  mine->n_ops = (uint32_t)ops.size();
  std::atomic_thread_fence(std::memory_order_seq_cst);
  if (!barrier(c)) return fail(ncclSystemError, "fake rccl: a rank did not arrive at the collective (timeout or broken communicator)");
  ncclResult_t result = ncclSuccess;
  for (int r = 0; r < c->n; r++) if (c->slot(r)->failed) result = ncclInternalError;
  std::vector<unsigned char> tmp;
  for (size_t i = 0; i < ops.size() && result == ncclSuccess; i++) {
    const Pending& p = ops[i];
    const size_t es = dtype_size(p.dtype);
    if (p.kind == OP_ALLREDUCE || p.kind == OP_ALLGATHER) {
      // the i-th collective of this rank pairs with the collective at the same position among the peers' collectives
      size_t ordinal = 0;
      for (size_t j = 0; j < i; j++) if (ops[j].kind == OP_ALLREDUCE || ops[j].kind == OP_ALLGATHER) ordinal++;
      const size_t bytes = p.count * es;
      tmp.assign(p.kind == OP_ALLREDUCE ? bytes : bytes * (size_t)c->n, 0);
      for (int r = 0; r < c->n && result == ncclSuccess; r++) {
        const RankSlot* s = c->slot(r);
        const OpDesc* d = nullptr;
        size_t k = 0;
        for (uint32_t j = 0; j < s->n_ops; j++)
          if (s->ops[j].kind == OP_ALLREDUCE || s->ops[j].kind == OP_ALLGATHER) { if (k == ordinal) { d = &s->ops[j]; break; } k++; }
        if (d == nullptr || d->kind != (uint32_t)p.kind || d->count != p.count || d->dtype != (uint32_t)p.dtype || (p.kind == OP_ALLREDUCE && d->red != (uint32_t)p.red)) {
          result = fail(ncclInvalidUsage, "fake rccl: ranks posted different collectives");
          break;
        }
        const unsigned char* src = c->payload(r) + d->payload_off;
        if (p.kind == OP_ALLGATHER) std::memcpy(tmp.data() + bytes * (size_t)r, src, bytes);
        else if (r == 0) std::memcpy(tmp.data(), src, bytes);
        else if (!reduce_bytes(tmp.data(), src, p.count, p.dtype, p.red)) result = fail(ncclInvalidArgument, "fake rccl: unsupported reduction");
      }
      if (result == ncclSuccess && !tmp.empty() && hipMemcpy(p.recv, tmp.data(), tmp.size(), hipMemcpyDefault) != hipSuccess)
        result = fail(ncclUnhandledCudaError, "fake rccl: staging → device copy failed");
    } else if (p.kind == OP_RECV) {
      // my k-th receive from `peer` pairs with the peer's k-th send to me
      size_t ordinal = 0;
      for (size_t j = 0; j < i; j++) if (ops[j].kind == OP_RECV && ops[j].peer == p.peer) ordinal++;
      const RankSlot* s = c->slot(p.peer);
      const OpDesc* d = nullptr;
      size_t k = 0;
      for (uint32_t j = 0; j < s->n_ops; j++)
        if (s->ops[j].kind == OP_SEND && s->ops[j].peer == (uint32_t)c->rank) { if (k == ordinal) { d = &s->ops[j]; break; } k++; }
      if (d == nullptr) { result = fail(ncclInvalidUsage, "fake rccl: a receive without a matching send in the same round"); break; }
      if (d->count * dtype_size((ncclDataType_t)d->dtype) != p.count * es) { result = fail(ncclInvalidUsage, "fake rccl: send and receive sizes differ"); break; }
      if (p.count > 0 && hipMemcpy(p.recv, c->payload(p.peer) + d->payload_off, p.count * es, hipMemcpyDefault) != hipSuccess)
        result = fail(ncclUnhandledCudaError, "fake rccl: staging → device copy failed");
    } else if (p.kind == OP_SEND) {
      // every send must be received in this round (RCCL would hang otherwise): checked from the receiver's side above
      if (p.peer < 0 || p.peer >= c->n) result = fail(ncclInvalidArgument, "fake rccl: peer out of range");
    }
  }
  if (result != ncclSuccess && g_error.empty()) g_error = "fake rccl: a peer failed in this round";
  // nobody overwrites its slot before every rank has read what it needs
  if (!barrier(c) && result == ncclSuccess) result = fail(ncclSystemError, "fake rccl: a rank did not leave the collective (timeout)");
  return result;
}

ncclResult_t post(Pending p) {
  if (p.comm == nullptr || p.comm->hdr == nullptr) return fail(ncclInvalidArgument, "fake rccl: null communicator");
  g_pending.push_back(p);
  if (g_group_depth > 0) return ncclSuccess;
  std::vector<Pending> ops;
  ops.swap(g_pending);
  return run_round(ops);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<uint64_t> counter{0};
  std::memset(id, 0, sizeof(*id));
  uint64_t w[4] = {(uint64_t)getpid(), (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(), counter.fetch_add(1) + 1, 0x9E3779B97F4A7C15ull};
  for (int i = 0; i < 4; i++) { w[i] ^= w[(i + 1) & 3] * 0xff51afd7ed558ccdULL; w[i] ^= w[i] >> 29; }
  std::memcpy(id->internal, w, sizeof(w));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (comm == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "fake rccl: bad rank / size");
  ncclComm* c = new ncclComm();
  (void)hipGetDevice(&c->device);
  const ncclResult_t r = attach(c, id, nranks, rank, false);
  if (r != ncclSuccess) { delete c; return r; }
  // like RCCL: returns once every rank has joined
  if (!spin([&] { return c->hdr->attached.load() >= (uint32_t)nranks; })) { delete c; return fail(ncclSystemError, "fake rccl: not every rank joined the communicator"); }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  if (comms == nullptr || ndev < 1) return fail(ncclInvalidArgument, "fake rccl: bad device count");
  ncclUniqueId id;
  ncclGetUniqueId(&id);
  for (int r = 0; r < ndev; r++) {
    ncclComm* c = new ncclComm();
    c->device = devlist != nullptr ? devlist[r] : r;
    const ncclResult_t e = attach(c, id, ndev, r, r == 0);
    if (e != ncclSuccess) { delete c; return e; }
    comms[r] = c;
  }
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (comm == nullptr) return ncclSuccess;
  if (comm->hdr != nullptr) {
    const bool last = comm->hdr->detached.fetch_add(1) + 1 >= comm->hdr->n_ranks;
    munmap(comm->hdr, comm->map_bytes);
    if (last) unlink(comm->name.c_str());
  }
  delete comm;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { if (comm == nullptr || count == nullptr) return ncclInvalidArgument; *count = comm->n; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) { if (comm == nullptr || rank == nullptr) return ncclInvalidArgument; *rank = comm->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return fail(ncclInvalidUsage, "fake rccl: ncclGroupEnd without ncclGroupStart");
  if (--g_group_depth > 0) return ncclSuccess;
  std::vector<Pending> ops;
  ops.swap(g_pending);
  return run_round(ops);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  return post(Pending{OP_ALLREDUCE, sendbuff, recvbuff, count, datatype, op, -1, comm, stream});
}
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  return post(Pending{OP_ALLGATHER, sendbuff, recvbuff, sendcount, datatype, ncclSum, -1, comm, stream});
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(Pending{OP_SEND, sendbuff, nullptr, count, datatype, ncclSum, peer, comm, stream});
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(Pending{OP_RECV, nullptr, recvbuff, count, datatype, ncclSum, peer, comm, stream});
}

const char* ncclGetErrorString(ncclResult_t result) {
  static thread_local std::string text;
  const char* base = result == ncclSuccess ? "no error" : result == ncclUnhandledCudaError ? "unhandled hip error" : result == ncclSystemError ? "unhandled system error" :
                     result == ncclInternalError ? "internal error" : result == ncclInvalidArgument ? "invalid argument" : result == ncclInvalidUsage ? "invalid usage" : "error";
  text = std::string(base) + (g_error.empty() || result == ncclSuccess ? "" : " — " + g_error);
  return text.c_str();
}

}  // extern "C"
