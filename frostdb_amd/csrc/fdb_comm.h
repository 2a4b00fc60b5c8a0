// fdb_comm.h — the cross-GPU merge of per-GPU partial tables behind the C ABI (include/frostdb_amd.h, "cross-GPU merge").
//
// ≙ Synchronizer + HashAggregate(final=true) (synchronize.go:31-53, physicalplan.go:438-471) when the chains of a query run
// on different GPUs. A `Comm` is one rank's endpoint of a communicator; two transports implement it:
//   * RCCL (librccl bound at run time): ncclCommInitRank (one process per GPU) or ncclCommInitAll (one process, N devices);
//   * local: the ranks are threads of one process and read each other's device buffers directly (peer-to-peer loads / copies
//     over xGMI, or plain loads when ranks share a device) — host rendezvous, no RCCL.
// The plan-level operations (Plan::comm_allreduce / comm_exchange, fdb_plan.h) are written against this interface only.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace fdb {

class Comm {
 public:
  virtual ~Comm() = default;
  int rank = 0, size = 1, device = 0;
  std::string error;  // text of the last failure on this handle (fdb_comm_last_error)
  // Ranks the TRANSPORT reports for this communicator (RCCL: ncclCommCount; the in-process transport: its group's size; -1 when the
  // library cannot say). bench.py prints it per rank so that a line claiming N GPUs shows N ranks inside the communicator itself.
  virtual int transport_ranks() { return size; }

  // Host-level control exchange (blocking, collective): every rank's blob, in rank order.
  virtual std::vector<std::vector<uint8_t>> all_gather_host(const std::vector<uint8_t>& mine) = 0;
  // Element-wise MAX of four int64 across ranks, result on the host. Runs next to (not behind) work queued on plan streams.
  virtual void probe_max(int64_t v[4]) = 0;
  // In-place all-reduce of arrays of 8-byte elements, enqueued on `stream` as ONE group.
  // op: 1 int64 sum, 2 float64 sum, 3 int64 min, 4 int64 max (Plan::state_array_op's numbering).
  struct Red { void* buf; size_t count; int op; };
  virtual void all_reduce(const std::vector<Red>& reds, hipStream_t stream) = 0;
  // All-gather of `bytes` bytes per rank (a multiple of 8): recv = every rank's send block, in rank order. Enqueued on `stream`.
  virtual void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) = 0;
  // Exchange of packed rows in 8-byte words. `send` holds this rank's `size` partitions back to back (words[rank][p] words for
  // rank p); `recv` receives words[p][rank] words from every rank p, grouped by source in rank order. The whole matrix is known
  // to every rank (all_gather_host). Complete on `stream` order; the caller synchronises before reading `recv`.
  virtual void all_to_all(const unsigned long long* send, unsigned long long* recv, const std::vector<std::vector<int64_t>>& words,
                          hipStream_t stream) = 0;
};

// RCCL
void rccl_unique_id(uint8_t id[128]);
std::unique_ptr<Comm> rccl_init_rank(const uint8_t id[128], int n_ranks, int rank, int device);
std::vector<std::unique_ptr<Comm>> rccl_init_all(const int* devices, int n);
// in-process peer-to-peer transport
std::vector<std::unique_ptr<Comm>> local_init(const int* devices, int n);

}  // namespace fdb
