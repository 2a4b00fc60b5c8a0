// Measurement aid: N chains pushing small HOST records concurrently — std::thread straight on the C ABI, no
// interpreter anywhere near the timed region. What the Go shim drives: one plan per chain (one goroutine per chain, physicalplan.go:337-347),
// records of 1 024 rows (the reference's batch floor, table.go:780) … 65 536 rows through fdb_plan_push_many, then the chains' plans merged
// (fdb_plan_merge ≙ Synchronizer + final stage) and one fdb_plan_finish. cfg 2's query: labels.code == '200', SUM(value) BY labels.path.
// The records are hand-built Arrow C Data Interface structs (dictionary<uint32, binary> labels, int64 timestamp, float64 value), every chain
// its own copies of the dictionaries' BUFFERS (equal content, different addresses: what per-record dictionaries of a Go producer look like).
// Build:  g++ -O2 -std=c++17 -Iinclude tools/push_bench.cpp frostdb_amd/libfrostdb_amd.so -Wl,-rpath,$PWD/frostdb_amd -lpthread -o tools/push_bench
// Run:    tools/push_bench [rows per record = 1024] [records per chain = 1024] [chains, comma separated = 1,8,16,32,64,128]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "arrow_c_data.h"
#include "frostdb_amd.h"

namespace {
void noop_release_array(ArrowArray* a) { a->release = nullptr; }
void noop_release_schema(ArrowSchema* s) { s->release = nullptr; }

struct Rng { uint64_t s; uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; } };

// One record's storage + C structs. Children: 0 labels.code, 1 labels.path, 2 timestamp, 3 value.
struct Record {
  std::vector<uint32_t> code_idx, path_idx;
  std::vector<int64_t> ts;
  std::vector<double> value;
  ArrowArray arr, kids[4], dicts[2];
  ArrowArray* kid_ptr[4];
  const void* buf_struct[1];
  const void* buf_kid[4][2];
  const void* buf_dict[2][3];
};
struct Dict {  // dictionary<binary> storage: offsets + bytes
  std::vector<int32_t> off;
  std::string bytes;
};
Dict make_dict(const std::vector<std::string>& values) {
  Dict d;
  d.off.push_back(0);
  for (const std::string& v : values) { d.bytes += v; d.off.push_back((int32_t)d.bytes.size()); }
  return d;
}
void fill_record(Record& r, int64_t rows, Rng& rng, const Dict& codes, const Dict& paths, int64_t n_codes, int64_t n_paths, int64_t row_base) {
  r.code_idx.resize((size_t)rows); r.path_idx.resize((size_t)rows); r.ts.resize((size_t)rows); r.value.resize((size_t)rows);
  for (int64_t i = 0; i < rows; i++) {
    const uint64_t x = rng.next();
    r.code_idx[(size_t)i] = (x % 10) < 7 ? 0u : (uint32_t)(1 + (x >> 8) % (uint64_t)(n_codes - 1));  // 70 % '200'
    r.path_idx[(size_t)i] = (uint32_t)((x >> 20) % (uint64_t)n_paths);
    r.ts[(size_t)i] = 1700000000000LL + 15000 * ((row_base + i) / 4096);
    r.value[(size_t)i] = (double)((x >> 11) % 1000000) / 1000.0;
  }
  auto prim = [&](int k, const void* data) {
    ArrowArray& a = r.kids[k];
    std::memset(&a, 0, sizeof(a));
    a.length = rows; a.null_count = 0; a.offset = 0; a.n_buffers = 2; a.n_children = 0;
    r.buf_kid[k][0] = nullptr; r.buf_kid[k][1] = data;
    a.buffers = r.buf_kid[k]; a.release = noop_release_array;
    r.kid_ptr[k] = &a;
  };
  prim(0, r.code_idx.data()); prim(1, r.path_idx.data()); prim(2, r.ts.data()); prim(3, r.value.data());
  auto dict = [&](int k, const Dict& d, int64_t n) {
    ArrowArray& a = r.dicts[k];
    std::memset(&a, 0, sizeof(a));
    a.length = n; a.null_count = 0; a.n_buffers = 3;
    r.buf_dict[k][0] = nullptr; r.buf_dict[k][1] = d.off.data(); r.buf_dict[k][2] = d.bytes.data();
    a.buffers = r.buf_dict[k]; a.release = noop_release_array;
    r.kids[k].dictionary = &a;
  };
  dict(0, codes, n_codes); dict(1, paths, n_paths);
  std::memset(&r.arr, 0, sizeof(r.arr));
  r.arr.length = rows; r.arr.null_count = 0; r.arr.n_buffers = 1; r.buf_struct[0] = nullptr; r.arr.buffers = r.buf_struct;
  r.arr.n_children = 4; r.arr.children = r.kid_ptr; r.arr.release = noop_release_array;
}

struct Schema {
  ArrowSchema root, kids[4], dicts[2];
  ArrowSchema* kid_ptr[4];
  Schema() {
    auto leaf = [&](int k, const char* fmt, const char* name) {
      ArrowSchema& s = kids[k];
      std::memset(&s, 0, sizeof(s));
      s.format = fmt; s.name = name; s.flags = 2 /* nullable */; s.release = noop_release_schema;
      kid_ptr[k] = &s;
    };
    leaf(0, "I", "labels.code"); leaf(1, "I", "labels.path"); leaf(2, "l", "timestamp"); leaf(3, "g", "value");
    for (int k = 0; k < 2; k++) {
      std::memset(&dicts[k], 0, sizeof(ArrowSchema));
      dicts[k].format = "z"; dicts[k].name = ""; dicts[k].release = noop_release_schema;
      kids[k].dictionary = &dicts[k];
    }
    std::memset(&root, 0, sizeof(root));
    root.format = "+s"; root.name = ""; root.n_children = 4; root.children = kid_ptr; root.release = noop_release_schema;
  }
};

void check(int rc, fdb_plan* p, const char* what) {
  if (rc == 0) return;
  std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, p ? fdb_plan_last_error(p) : fdb_last_error());
  std::exit(1);
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? std::atoll(argv[1]) : 1024;
  const int per_chain = argc > 2 ? std::atoi(argv[2]) : 1024;
  std::vector<int> chain_counts;
  { std::string s = argc > 3 ? argv[3] : "1,8,16,32,64,128"; size_t at = 0; while (at < s.size()) { chain_counts.push_back(std::atoi(s.c_str() + at)); at = s.find(',', at); if (at == std::string::npos) break; at++; } }
  const int64_t n_codes = 6, n_paths = 1024;
  std::vector<std::string> cv = {"200", "404", "500", "301", "201", "503"}, pv;
  for (int i = 0; i < n_paths; i++) { char b[32]; std::snprintf(b, sizeof b, "/api/v1/path/%04d", i); pv.push_back(b); }
  // the query
  fdb_expr leaf; std::memset(&leaf, 0, sizeof(leaf));
  leaf.op = FDB_OP_EQ; leaf.left = leaf.right = -1; leaf.column = "labels.code"; leaf.literal.type = FDB_LIT_STRING; leaf.literal.data = "200"; leaf.literal.len = 3;
  fdb_aggregation agg; std::memset(&agg, 0, sizeof(agg)); agg.func = FDB_AGG_SUM; agg.column = "value";
  fdb_group_expr grp; std::memset(&grp, 0, sizeof(grp)); grp.name = "labels.path";
  fdb_plan_desc desc; std::memset(&desc, 0, sizeof(desc));
  desc.filter = &leaf; desc.n_filter = 1; desc.filter_root = 0; desc.aggs = &agg; desc.n_aggs = 1; desc.groups = &grp; desc.n_groups = 1;
  Schema schema;
  std::printf("# push_bench: records of %lld rows, %d records per chain; labels.code == '200', SUM(value) BY labels.path (%lld paths)\n", (long long)rows, per_chain, (long long)n_paths);
  for (int chains : chain_counts) {
    if (chains <= 0) continue;
    // every chain: its own dictionaries (buffers) and records
    std::vector<Dict> cd((size_t)chains), pd((size_t)chains);
    std::vector<std::vector<std::unique_ptr<Record>>> recs((size_t)chains);
    std::vector<std::vector<ArrowArray*>> ap((size_t)chains);
    std::vector<std::vector<ArrowSchema*>> sp((size_t)chains);
    double expect = 0;
    for (int c = 0; c < chains; c++) {
      cd[(size_t)c] = make_dict(cv); pd[(size_t)c] = make_dict(pv);
      Rng rng{0x9E3779B97F4A7C15ull * (uint64_t)(c + 1)};
      for (int k = 0; k < per_chain; k++) {
        recs[(size_t)c].emplace_back(new Record());
        Record& r = *recs[(size_t)c].back();
        fill_record(r, rows, rng, cd[(size_t)c], pd[(size_t)c], n_codes, n_paths, (int64_t)k * rows);
        for (int64_t i = 0; i < rows; i++) if (r.code_idx[(size_t)i] == 0) expect += r.value[(size_t)i];
        ap[(size_t)c].push_back(&r.arr); sp[(size_t)c].push_back(&schema.root);
      }
    }
    double best_push = 1e30, best_merge = 0, best_finish = 0;
    for (int rep = 0; rep < 4; rep++) {
      std::vector<fdb_plan*> plans((size_t)chains, nullptr);
      for (int c = 0; c < chains; c++) check(fdb_plan_create(&desc, 0, &plans[(size_t)c]), nullptr, "fdb_plan_create");
      std::atomic<int> ready{0}, go{0};
      std::vector<std::thread> ts;
      for (int c = 0; c < chains; c++)
        ts.emplace_back([&, c] {
          ready.fetch_add(1);
          while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
          int32_t pushed = 0;
          check(fdb_plan_push_many(plans[(size_t)c], ap[(size_t)c].data(), sp[(size_t)c].data(), per_chain, &pushed), plans[(size_t)c], "fdb_plan_push_many");
          int64_t g = 0;
          check(fdb_plan_num_groups(plans[(size_t)c], &g), plans[(size_t)c], "fdb_plan_num_groups");  // (settle: the queued records are scanned)
        });
      while (ready.load() < chains) std::this_thread::yield();
      const double t0 = now_ms();
      go.store(1, std::memory_order_release);
      for (std::thread& t : ts) t.join();
      const double t1 = now_ms();
      for (int c = 1; c < chains; c++) check(fdb_plan_merge(plans[0], plans[(size_t)c]), plans[0], "fdb_plan_merge");
      const double t2 = now_ms();
      ArrowArray out; ArrowSchema outs; int64_t n = 0;
      std::memset(&out, 0, sizeof(out)); std::memset(&outs, 0, sizeof(outs));
      check(fdb_plan_finish(plans[0], &out, &outs, &n), plans[0], "fdb_plan_finish");
      const double t3 = now_ms();
      double got = 0;  // Σ over groups of sum(value): children = [labels.path, sum(value)]
      if (out.n_children == 2 && out.children[1]->n_buffers >= 2) { const double* v = (const double*)out.children[1]->buffers[1]; for (int64_t i = 0; i < n; i++) got += v[i]; }
      if (out.release) out.release(&out);
      if (outs.release) outs.release(&outs);
      for (fdb_plan* p : plans) fdb_plan_close(p);
      if (std::abs(got - expect) > 1e-6 * std::abs(expect)) { std::fprintf(stderr, "WRONG RESULT: %f vs %f\n", got, expect); return 1; }
      if (rep > 0 && t1 - t0 < best_push) { best_push = t1 - t0; best_merge = t2 - t1; best_finish = t3 - t2; }
    }
    const double total_rows = (double)chains * per_chain * rows;
    std::printf("%4d chains: push %8.2f ms = %7.2f us per record and chain, %6.3f G rows/s; %3d merges %6.2f ms; finish %5.2f ms; whole query %6.3f G rows/s\n", chains, best_push,
                best_push * 1e3 / per_chain, total_rows / best_push / 1e6, chains - 1, best_merge, best_finish, total_rows / (best_push + best_merge + best_finish) / 1e6);
    std::fflush(stdout);
  }
  return 0;
}
