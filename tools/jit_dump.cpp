// Tuning aid (CPU-only): prints the run-time specialised hash-scan kernel for a cfg 5-like shape (N dictionary key columns with
// validity, SUM(float64) + COUNT) so that it can be compiled offline:
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I frostdb_amd/csrc -I include -I /opt/rocm/include tools/jit_dump.cpp frostdb_amd/csrc/*.o \
//       -L/opt/rocm/lib -lamdhip64 -lhiprtc -ldl -lpthread -lz -Wl,-rpath,/opt/rocm/lib -o /tmp/jit_dump
//   (the library's objects, not the .so: the generators are internals and the .so exports the C ABI only)
//   /tmp/jit_dump 32 > /tmp/k.hip && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DFDB_DEVICE_ONLY=1 \
//       -I frostdb_amd/csrc --cuda-device-only -Rpass-analysis=kernel-resource-usage -c /tmp/k.hip -o /tmp/k.o
#include <cstdio>
#include <cstdlib>
#include <string>

#include "frostdb_amd.h"
#include "fdb_kernels.h"
#include "fdb_jit.h"

int main(int argc, char** argv) {
  if (argc > 1 && (std::string(argv[1]) == "flags" || std::string(argv[1]) == "select")) {
    const bool select = std::string(argv[1]) == "select";
    // the selection-bitmap kernel of filter(): `value > T AND labels.code == <one of a few>` (an 8-byte compare + a dictionary truth table)
    fdb::JitShape s;
    s.block = 512; s.two_phase = true; s.lds_acc = false;
    s.n_c8 = 1; s.c8[0].has_values = true; s.c8[0].has_validity = 0;
    s.n_c4 = 1; s.c4[0].has_values = true; s.c4[0].has_validity = 2;
    fdb::JitLeaf a; a.kind = FDB_LEAF_CMP_F64; a.slot = 0; a.wide = 1; a.op = 5;
    fdb::JitLeaf b; b.kind = FDB_LEAF_DICT_BITS; b.slot = 0; b.wide = 0;
    fdb::JitLeaf c; c.kind = FDB_LEAF_DICT_LUT; c.slot = 0; c.wide = 0; c.lut_in_lds = true;
    s.leaves = {a, b, c};
    s.code = {0, 1, FDB_CODE_AND, 2, FDB_CODE_OR};
    if (argc > 2) { s.n_c4 = 0; s.leaves = {a}; s.code = {0}; }  // `value > T` alone (bench.py's select line)
    if (select) { s.fuse8 = 1; if (argc > 3) s.fuse4 = 1; }  // the one-pass kernel: `value` (and the dictionary column) compacted by the predicate's own wave
    std::fputs((select ? fdb::jit_select_source(s) : fdb::jit_flags_source(s)).c_str(), stdout);
    return 0;
  }
  if (argc > 1 && std::string(argv[1]) == "plan") {
    // the dense scan kernel of a cfg 2-like shape (`labels.code == X` + SUM(float64) [+ MIN / COUNT] GROUP BY labels.path):
    //   plan [variant]   variant: lds (default) | wave (per-wave tables: fdb_plan_set_deterministic) | reg (≤ 8 slots in registers) | cache (table too big for LDS) | two (two-phase layout, 4 aggregates)
    const std::string v = argc > 2 ? argv[2] : "lds";
    fdb::JitShape s;
    s.block = v == "wave" ? 256 : 512;
    s.two_phase = v == "two";
    s.lds_acc = v != "cache"; s.cache = v == "cache"; s.need_count = v == "two";
    s.wave_tables = v == "wave";
    if (v == "reg") s.reg_slots = 6;
    fdb::JitLeaf a; a.kind = FDB_LEAF_DICT_BITS; a.slot = 0; a.wide = 0;
    s.leaves = {a};
    s.code = {0};
    if (s.two_phase) {
      s.n_c4 = 1; s.c4[0].has_values = true; s.c4[0].has_validity = 1;
      s.n_l4 = 1; s.l4[0].has_values = true; s.l4[0].has_validity = 2;
      s.n_l8 = 2; s.l8[0].has_values = true; s.l8[1].has_values = true;
      s.gcols.push_back({0, true});
      s.aggs = {{FDB_AGG_COUNT, FDB_T_I64, 1, 0}, {FDB_AGG_MIN, FDB_T_I64, 0, 0}, {FDB_AGG_MAX, FDB_T_I64, 0, 0}, {FDB_AGG_SUM, FDB_T_F64, 1, 0}};
    } else {
      s.n_c4 = 2; s.c4[0].has_values = true; s.c4[0].has_validity = 1; s.c4[1].has_values = true; s.c4[1].has_validity = 1;
      s.n_c8 = 1; s.c8[0].has_values = true;
      s.gcols.push_back({1, true});
      s.aggs = {{FDB_AGG_SUM, FDB_T_F64, 0, 0}, {FDB_AGG_MIN, FDB_T_F64, 0, 0}};
    }
    std::fputs(fdb::jit_source(s).c_str(), stdout);
    return 0;
  }
  const int n = argc > 1 ? std::atoi(argv[1]) : 32;
  const int kinds = argc > 2 ? std::atoi(argv[2]) : 0;  // 1: make the last column an int64 key; 2: run kernel, narrow records; 3: run kernel, wide records; 4: wide + an int64 key; 5: run kernel, medium records
  fdb::JitHashShape s;
  for (int c = 0; c < n; c++) s.cols.push_back({(kinds == 1 || kinds == 4) && c == n - 1 ? 1 : 0, true, c < 8, -1});
  s.aggs.push_back({FDB_AGG_SUM, FDB_T_F64, -1, 0});
  s.agg_validity.push_back(false);
  if (kinds >= 2) s.runs = kinds == 2 ? 1 : kinds == 5 ? 3 : 2;  // the table-free OrderedAggregate's run kernel (one aggregation); 5: medium records (two bytes per key id)
  else { s.aggs.push_back({FDB_AGG_COUNT, FDB_T_I64, -1, 0}); s.agg_validity.push_back(false); }
  s.need_count = true;
  std::fputs(fdb::jit_hash_source(s).c_str(), stdout);
  return 0;
}
