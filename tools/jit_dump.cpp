// Tuning aid (CPU-only): prints the run-time specialised hash-scan kernel for a cfg 5-like shape (N dictionary key columns with
// validity, SUM(float64) + COUNT) so that it can be compiled offline:
//   g++ -std=c++17 -I frostdb_amd/csrc -I include tools/jit_dump.cpp frostdb_amd/libfrostdb_amd.so -o /tmp/jit_dump
//   /tmp/jit_dump 32 > /tmp/k.hip && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DFDB_DEVICE_ONLY=1 \
//       -I frostdb_amd/csrc --cuda-device-only -Rpass-analysis=kernel-resource-usage -c /tmp/k.hip -o /tmp/k.o
#include <cstdio>
#include <cstdlib>

#include "frostdb_amd.h"
#include "fdb_kernels.h"
#include "fdb_jit.h"

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 32;
  const int kinds = argc > 2 ? std::atoi(argv[2]) : 0;  // 1: make the last column an int64 key
  fdb::JitHashShape s;
  for (int c = 0; c < n; c++) s.cols.push_back({kinds == 1 && c == n - 1 ? 1 : 0, true, c < 8, -1});
  s.aggs.push_back({FDB_AGG_SUM, FDB_T_F64, -1, 0});
  s.agg_validity.push_back(false);
  s.aggs.push_back({FDB_AGG_COUNT, FDB_T_I64, -1, 0});
  s.agg_validity.push_back(false);
  s.need_count = true;
  std::fputs(fdb::jit_hash_source(s).c_str(), stdout);
  return 0;
}
