// fdb_plan_internal.h — per-record resolution state shared by fdb_plan.cpp and fdb_hash.cpp (not part of any API).
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "fdb_plan.h"

namespace fdb {

inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PhaseTimer {  // FDB_PROFILE=1: per-phase host microseconds on stderr (tuning aid)
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseTimer() : on(std::getenv("FDB_PROFILE") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[fdb] %-20s %8.1f us\n", what, std::chrono::duration<double, std::micro>(n - t).count());
    t = n;
  }
};

struct Blob {  // LUTs of one batch, shipped with one copy
  std::vector<uint8_t> bytes;
  size_t add(const void* p, size_t n) {
    const size_t off = align_up_sz(bytes.size(), 16);
    bytes.resize(off + std::max<size_t>(n, 1), 0);
    if (n) std::memcpy(bytes.data() + off, p, n);
    return off;
  }
};

struct PendingLut { int kind; int index; size_t blob_off; size_t len_bytes; };  // kind 0: leaf, 1: group col


struct GroupRes {
  int gi, ci, kind;      // kind 2: computed int64 key (ci = -1, expr_root = root node in the record's args.expr)
  int expr_root = -1;
  std::shared_ptr<const std::vector<uint32_t>> lut;
};

struct Plan::Resolved {
  TruthCache* truths = nullptr;           // the plan's cache
  int cur_node = -1;                      // filter node being resolved
  FdbScanArgs args;
  Blob blob;
  std::vector<PendingLut> luts;
  std::vector<char> counted;  // per batch column: bit 0 values, bit 1 validity already counted in algorithmic bytes
  int64_t bytes = 0;
  int leaf_col[FDB_MAX_LEAVES];           // batch column behind each leaf (-1: constant leaf)
  int gcol_col[FDB_MAX_DENSE_GCOLS];
  int agg_col[FDB_MAX_AGGS];
  int expr_col[FDB_MAX_EXPR_NODES];       // batch column behind each expression column node
  std::vector<GroupRes> groups;           // group-by columns of this record (plan-level index, record column, kind, key-id LUT)
  // AndExpr.Eval is lazy (filter.go:172-190): when the left side of an AND selects no row of the record the right side is not
  // evaluated — so a right side that cannot be evaluated on this record (an operator its column type does not support) only is
  // an error if the left side selects something. Set by the plan: rows of this record the sub-tree rooted at `node` selects.
  std::function<int64_t(int node)> count_selected;
  Resolved() {
    for (int& v : leaf_col) v = -1;
    for (int& v : gcol_col) v = -1;
    for (int& v : agg_col) v = -1;
    for (int& v : expr_col) v = -1;
  }
  // Algorithmic bytes: each referenced buffer once per row, whatever the number of references.
  void count(const DeviceBatch& b, int ci, bool values = true) {
    if (counted.empty()) counted.assign(b.cols.size(), 0);
    char& c = counted[(size_t)ci];
    if (values && !(c & 1)) { c |= 1; bytes += b.cols[(size_t)ci].value_bytes; }
    if (!(c & 2)) { c |= 2; bytes += b.cols[(size_t)ci].validity_bytes; }
  }
};


}  // namespace fdb
