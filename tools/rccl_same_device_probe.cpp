// Probe: does RCCL accept two ranks of one communicator on the SAME device (needed to test multi-rank merges on a 1-GPU box)?
// hipcc tools/rccl_same_device_probe.cpp -o tools/rccl_same_device_probe -lrccl -lpthread
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <thread>
#include <vector>
int main() {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) { std::printf("ncclGetUniqueId failed\n"); return 2; }
  int rc[2] = {0, 0};
  auto work = [&](int r) {
    hipSetDevice(0);
    ncclComm_t c;
    ncclResult_t e = ncclCommInitRank(&c, 2, id, r);
    if (e != ncclSuccess) { std::printf("rank %d: ncclCommInitRank: %s\n", r, ncclGetErrorString(e)); rc[r] = 1; return; }
    long long* d; hipMalloc(&d, 8192);
    std::vector<long long> h(1024, r + 1);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    e = ncclAllReduce(d, d, 1024, ncclInt64, ncclSum, c, s);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost);
    std::printf("rank %d: allreduce %s, h[0]=%lld (want 3)\n", r, ncclGetErrorString(e), h[0]);
    ncclCommDestroy(c);
  };
  std::thread t0(work, 0), t1(work, 1);
  t0.join(); t1.join();
  return rc[0] | rc[1];
}
