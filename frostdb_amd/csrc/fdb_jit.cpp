// fdb_jit.cpp — plan-specialised scan kernels, generated as HIP source and compiled at run time with hiprtc.
//
// The ahead-of-time kernels interpret the plan: for every 256-row tile they walk the predicate program, the group
// columns and the aggregate list through wave-uniform branches and register-select chains. rocprofv3 showed that for a
// multi-predicate plan (cfg 3) this makes the scan instruction-issue-bound (SQ_ACTIVE_INST_ANY ≈ 92 % of SIMD time,
// ≈725 wave-instructions per tile, 4.4 TB/s), while a hand-specialised kernel of the same plan streams at 6.4 TB/s
// (tools/bw_probe.hip). This file produces that specialised kernel for any plan the slot kernel accepts: the plan's
// SHAPE (which columns, which leaf kinds, the boolean expression, the aggregate list, LDS layout decisions) is baked
// into straight-line code; everything that varies between queries or records of the same shape (pointers, literals,
// truth-table bits, LUT offsets, strides, row counts) stays a run-time argument, read from the same FdbScanArgs blocks
// the interpreting kernel reads. Compiled code objects are cached per process and on disk, keyed by the source text.
//
// Failure to compile (hiprtc missing, unexpected shape) is not an error: the caller falls back to the interpreting
// kernels — still on the GPU.
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "fdb_jit.h"
#include "../../include/frostdb_amd.h"

namespace fdb {

namespace {

// The argument-block definitions the generated kernel shares with the host (fdb_kernels.h, embedded at build time).
const char* kKernelsHeader =
#include "fdb_kernels_h.inc"
    ;

// Wide run records are written from RE-LOADED columns (the lanes that end a run): the first pass then has to leave the columns in the
// L2 — with non-temporal loads the second pass went back to HBM (30.8 % of the roofline; 37.9 % with plain loads,
// profiles/round6_wide_records_temporal_loads.txt; $FDB_RUNS_WIDE_NT=1 restores the non-temporal ones). The hash-table kernel's inserting
// lanes re-load too, but there plain loads cost 3 % (they compete with the table's entries for the L2): it keeps the non-temporal ones.
bool temporal_loads(int runs) {
  const char* nt = std::getenv("FDB_RUNS_WIDE_NT");
  return runs == 2 && !(nt != nullptr && std::atoi(nt) != 0);
}

const char* kPreamble = R"HIP(
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define G1 __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const G1 T* as_global(const T* p) { return (const G1 T*)p; }
__device__ __forceinline__ long long f64_to_ordered(double d) { long long b = __double_as_longlong(d); return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL); }
// (a NaN contributes the identity of MIN / MAX, −0.0 counts as +0.0: see f64_minmax_key in fdb_kernels.hip)
__device__ __forceinline__ long long f64_minmax_key(double d, bool is_min) {
  d += 0.0;
  const long long k = f64_to_ordered(d);
  return d != d ? (is_min ? 0x7FFFFFFFFFFFFFFFLL : (-0x7FFFFFFFFFFFFFFFLL - 1)) : k;
}
// (FDB_LD_TEMPORAL: the kernel reads its columns twice — wide run records — so the first pass must leave them in the L2)
#ifdef FDB_LD_TEMPORAL
__device__ __forceinline__ u32x4 ld4(const void* base, uint32_t byte_off) {
  return *as_global(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + byte_off));
}
__device__ __forceinline__ u64x2 ld8(const void* base, uint32_t byte_off) {
  return *as_global(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(base) + byte_off));
}
#else
__device__ __forceinline__ u32x4 ld4(const void* base, uint32_t byte_off) {
  return __builtin_nontemporal_load(as_global(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + byte_off)));
}
__device__ __forceinline__ u64x2 ld8(const void* base, uint32_t byte_off) {
  return __builtin_nontemporal_load(as_global(reinterpret_cast<const u64x2*>(reinterpret_cast<const char*>(base) + byte_off)));
}
#endif
__device__ __forceinline__ uint32_t ldv(const uint8_t* bm, uint32_t byte_off, uint32_t shift) { return (as_global(bm)[byte_off] >> shift) & 0xFu; }
// dictionary truth table in a 64-bit immediate: 4 rows → 4-bit mask (NULL rows look up entry `null_at`)
__device__ __forceinline__ uint32_t leaf_bits(u32x4 idx, uint32_t valid, unsigned long long bits, uint32_t null_at) {
  const uint32_t i0 = (valid & 1u) ? idx.x : null_at, i1 = (valid & 2u) ? idx.y : null_at, i2 = (valid & 4u) ? idx.z : null_at, i3 = (valid & 8u) ? idx.w : null_at;
  return (uint32_t)((bits >> i0) & 1ull) | ((uint32_t)((bits >> i1) & 1ull) << 1) | ((uint32_t)((bits >> i2) & 1ull) << 2) | ((uint32_t)((bits >> i3) & 1ull) << 3);
}
template <typename LUT>
__device__ __forceinline__ uint32_t leaf_lut(u32x4 idx, uint32_t valid, LUT lut, uint32_t null_at) {
  const uint32_t i0 = (valid & 1u) ? idx.x : null_at, i1 = (valid & 2u) ? idx.y : null_at, i2 = (valid & 4u) ? idx.z : null_at, i3 = (valid & 8u) ? idx.w : null_at;
  return (uint32_t)lut[i0] | ((uint32_t)lut[i1] << 1) | ((uint32_t)lut[i2] << 2) | ((uint32_t)lut[i3] << 3);
}
// pre-aggregate arithmetic (project.go:163-399): Go int64 semantics — wrap-around, quotient truncated toward zero,
// MinInt64 / -1 wraps; a division by zero yields NULL, whose raw slot reads 0.
__device__ __forceinline__ long long i64_add(long long a, long long b) { return (long long)((unsigned long long)a + (unsigned long long)b); }
__device__ __forceinline__ long long i64_sub(long long a, long long b) { return (long long)((unsigned long long)a - (unsigned long long)b); }
__device__ __forceinline__ long long i64_mul(long long a, long long b) { return (long long)((unsigned long long)a * (unsigned long long)b); }
__device__ __forceinline__ long long i64_div(long long a, long long b) { return b == 0 ? 0 : b == -1 ? (long long)(0ull - (unsigned long long)a) : a / b; }
__device__ __forceinline__ long long u64_div(long long a, long long b) { return b == 0 ? 0 : (long long)((unsigned long long)a / (unsigned long long)b); }  // DivUint64s (project.go:379-395)
__device__ __forceinline__ double f64_div(double a, double b) { return b == 0.0 ? 0.0 : a / b; }
template <int OP, typename T> __device__ __forceinline__ bool cmp1(T a, T b) {
  return OP == 1 ? a == b : OP == 2 ? a != b : OP == 3 ? a < b : OP == 4 ? a <= b : OP == 5 ? a > b : a >= b;
}
template <int OP, typename T> __device__ __forceinline__ uint32_t cmp4(T a, T b, T c, T d, T lit, uint32_t valid) {
  return ((uint32_t)cmp1<OP, T>(a, lit) | ((uint32_t)cmp1<OP, T>(b, lit) << 1) | ((uint32_t)cmp1<OP, T>(c, lit) << 2) | ((uint32_t)cmp1<OP, T>(d, lit) << 3)) & valid;
}
)HIP";

// device | shape key → function. An entry is published BEFORE it is built: the thread that created it builds it (disk load or hiprtc,
// 130–500 ms) without holding g_mu, threads that want the same kernel wait for that entry alone, and everybody else's lookups go
// straight through. (Until round 6 the build ran under g_mu: while one chain compiled a new shape, every other query of the process —
// cached shapes too — stood still for the length of a compilation.)
struct KernelEntry {
  std::mutex m;
  std::condition_variable cv;
  bool ready = false;
  hipFunction_t fn = nullptr;
};
std::mutex g_mu;
std::unordered_map<std::string, std::shared_ptr<KernelEntry>> g_cache;
bool g_disabled = false;

uint64_t fnv(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

// Directory of the on-disk code-object cache, or "" when there is none that can be trusted: cached .hsaco files are loaded as GPU
// code, so the directory must be a real directory (not a symlink) owned by this user and closed to everybody else — another
// local user must not be able to plant code objects under a predictable name. $FDB_JIT_CACHE, else $XDG_CACHE_HOME/frostdb_amd,
// else ~/.cache/frostdb_amd, else /tmp/frostdb_amd_jit_<uid> (all subject to the same check).
std::string cache_dir() {
  std::string d;
  if (const char* e = std::getenv("FDB_JIT_CACHE")) d = e;
  else if (const char* x = std::getenv("XDG_CACHE_HOME"); x != nullptr && *x) d = std::string(x) + "/frostdb_amd";
  else if (const char* h = std::getenv("HOME"); h != nullptr && *h) { ::mkdir((std::string(h) + "/.cache").c_str(), 0700); d = std::string(h) + "/.cache/frostdb_amd"; }
  else d = "/tmp/frostdb_amd_jit_" + std::to_string((long)getuid());
  if (d.empty()) return "";
  (void)::mkdir(d.c_str(), 0700);
  struct stat st;
  if (::lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) {
    static bool warned = false;
    if (!warned) { warned = true; std::fprintf(stderr, "[frostdb_amd] JIT disk cache disabled: %s is not a private directory of this user\n", d.c_str()); }
    return "";
  }
  return d;
}

bool compile(const std::string& src, std::vector<char>* code, std::string* log) {
  hiprtcProgram prog;
  const char* hdr_names[] = {"fdb_kernels.h"};
  const char* hdr_src[] = {kKernelsHeader};
  if (hiprtcCreateProgram(&prog, src.c_str(), "fdb_plan_kernel.hip", 1, hdr_src, hdr_names) != HIPRTC_SUCCESS) { *log = "hiprtcCreateProgram failed"; return false; }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-DFDB_DEVICE_ONLY=1"};
  const hiprtcResult r = hiprtcCompileProgram(prog, 5, opts);
  size_t n = 0;
  hiprtcGetProgramLogSize(prog, &n);
  if (n > 1) { log->resize(n); hiprtcGetProgramLog(prog, &(*log)[0]); }
  if (r != HIPRTC_SUCCESS) { hiprtcDestroyProgram(&prog); return false; }
  size_t sz = 0;
  hiprtcGetCodeSize(prog, &sz);
  code->resize(sz);
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  return true;
}

// ---- code generation ---------------------------------------------------------------------------------------------------
// C expression of node `ni` (typed: long long or double); `col(node)` gives the unsigned 64-bit raw value of a column node,
// literals are the run-time variables K_elit<i>.
template <typename F, typename V>
std::string expr_valid(const std::vector<JitExprNode>& ex, int root, F col, V colvalid);
template <typename F, typename V>
std::string expr_value(const std::vector<JitExprNode>& ex, int ni, F col, V colvalid) {
  const JitExprNode& n = ex[(size_t)ni];
  const bool f = n.type == FDB_T_F64;
  if (n.kind == 0) return f ? ("__longlong_as_double((long long)" + col(ni) + ")") : ("(long long)" + col(ni));
  if (n.kind == 1) return f ? ("__longlong_as_double(K_elit" + std::to_string(ni) + ")") : ("K_elit" + std::to_string(ni));
  if (n.kind == 7) return "(long long)" + col(ni);  // a column compared with a literal the way a filter leaf does it (project.go:409-447): col(ni) = the leaf's match bit of this row
  if (n.kind == 4) return "(double)(" + expr_value(ex, n.left, col, colvalid) + ")";  // float64(c.Value(i)): the raw slot (project.go:523-535)
  if (n.kind == 5) return "(" + colvalid(n.left) + " ? 0ll : 1ll)";                     // cols[0].IsNull(i) (project.go:588-590)
  if (n.kind == 6)  // cond.IsValid(i) && cond.Value(i) ? a.Value(i) : b.Value(i) (project.go:685-701); conditions here are never NULL
    return "((" + expr_value(ex, n.op, col, colvalid) + " != 0ll) ? " + expr_value(ex, n.left, col, colvalid) + " : " + expr_value(ex, n.right, col, colvalid) + ")";
  std::string a = expr_value(ex, n.left, col, colvalid), b = expr_value(ex, n.right, col, colvalid);
  if (n.kind == 3) {
    // comparison → 0 / 1, never NULL: a NULL operand compares false like in a filter leaf (project.go:409-470); mixed
    // int64 / float64 operands compare as doubles
    if (n.op == FDB_OP_AND || n.op == FDB_OP_OR) return "((" + a + (n.op == FDB_OP_AND ? " & " : " | ") + b + ") & 1ll)";
    if (ex[(size_t)n.left].type != ex[(size_t)n.right].type) { a = "(double)" + a; b = "(double)" + b; }
    const char* op = n.op == FDB_OP_EQ ? " == " : n.op == FDB_OP_NOT_EQ ? " != " : n.op == FDB_OP_LT ? " < " : n.op == FDB_OP_LT_EQ ? " <= " : n.op == FDB_OP_GT ? " > " : " >= ";
    return "((" + expr_valid(ex, n.left, col, colvalid) + " && " + expr_valid(ex, n.right, col, colvalid) + " && (" + a + op + b + ")) ? 1ll : 0ll)";
  }
  if (f) {
    if (n.op == FDB_OP_DIV) return "f64_div(" + a + ", " + b + ")";
    return "(" + a + (n.op == FDB_OP_ADD ? " + " : n.op == FDB_OP_SUB ? " - " : " * ") + b + ")";
  }
  // (uint64: the same bits for + − × — both wrap modulo 2^64 —, an unsigned quotient)
  return std::string(n.op == FDB_OP_ADD ? "i64_add(" : n.op == FDB_OP_SUB ? "i64_sub(" : n.op == FDB_OP_MUL ? "i64_mul(" : n.type == FDB_T_U64 ? "u64_div(" : "i64_div(") + a + ", " + b + ")";
}
// Validity of the ROOT value: only an outermost division can be NULL (divisor 0); `colvalid(node)` for a bare column.
template <typename F, typename V>
std::string expr_valid(const std::vector<JitExprNode>& ex, int root, F col, V colvalid) {
  const JitExprNode& n = ex[(size_t)root];
  if (n.kind == 0) return colvalid(root);
  if (n.kind == 2 && n.op == FDB_OP_DIV) {
    const std::string d = expr_value(ex, n.right, col, colvalid);
    return n.type == FDB_T_F64 ? ("(" + d + " != 0.0)") : ("(" + d + " != 0)");
  }
  return "true";
}
inline std::string expr_bits(const std::vector<JitExprNode>& ex, int root, const std::string& v) {
  if (ex[(size_t)root].type == FDB_T_BOOL) return "(unsigned long long)(" + v + " + 1ll)";  // a bool key is 1 (false) / 2 (true), like a stored bool column
  return ex[(size_t)root].type == FDB_T_F64 ? ("(unsigned long long)__double_as_longlong(" + v + ")") : ("(unsigned long long)" + v);
}

struct Gen {
  std::ostringstream o;
  const JitShape& s;
  explicit Gen(const JitShape& sh) : s(sh) {}

  // name of the register holding slot `slot` of pool (wide, late)
  static std::string reg(bool wide, bool late, int slot) { return std::string(late ? "l" : "e") + (wide ? "8_" : "4_") + std::to_string(slot); }

  void loads(bool late) {
    const int n4 = late ? s.n_l4 : s.n_c4, n8 = late ? s.n_l8 : s.n_c8;
    for (int i = 0; i < n4; i++) {
      const JitSlot& c = late ? s.l4[i] : s.c4[i];
      const std::string r = reg(false, late, i);
      if (c.has_values) o << "    const u32x4 " << r << " = ld4(P_" << r << "_v + tile_off4, lane_off4);\n";
      mask(c, r);
    }
    for (int i = 0; i < n8; i++) {
      const JitSlot& c = late ? s.l8[i] : s.c8[i];
      const std::string r = reg(true, late, i);
      if (c.has_values) {
        o << "    const u64x2 " << r << "a = ld8(P_" << r << "_v + tile_off8, lane_off8);\n";
        o << "    const u64x2 " << r << "b = ld8(P_" << r << "_v + tile_off8, lane_off8 + 16u);\n";
      }
      mask(c, r);
    }
  }

  void mask(const JitSlot& c, const std::string& r) {
    if (c.has_validity == 1) o << "    const uint32_t " << r << "_m = ldv(P_" << r << "_b + tile_offb, lane_offb, lane_shb);\n";
    else if (c.has_validity == 2) o << "    const uint32_t " << r << "_m = P_" << r << "_b != nullptr ? ldv(P_" << r << "_b + tile_offb, lane_offb, lane_shb) : 0xFu;\n";
    else o << "    const uint32_t " << r << "_m = 0xFu;\n";
  }

  std::string leaf_expr(int l) { return leaf_expr_of(s.leaves[l], l, reg(s.leaves[l].wide, false, s.leaves[l].slot)); }

  // 4-bit match mask of leaf `l` over the registers `r` (values r / r+"a", r+"b"; validity nibble r+"_m")
  static std::string leaf_expr_of(const JitLeaf& L, int l, const std::string& r) {
    const std::string li = std::to_string(l);
    std::ostringstream e;
    switch (L.kind) {
      case FDB_LEAF_CONST: e << "(K_op" << li << " ? 0xFu : 0u)"; break;
      case FDB_LEAF_VALIDITY: e << "(K_op" << li << " ? " << r << "_m : (~" << r << "_m & 0xFu))"; break;
      case FDB_LEAF_DICT_BITS: e << "leaf_bits(" << r << ", " << r << "_m, (unsigned long long)K_lit" << li << ", K_len" << li << " - 1u)"; break;
      case FDB_LEAF_DICT_LUT:
        if (L.lut_in_lds) e << "leaf_lut(" << r << ", " << r << "_m, smem + K_lds" << li << ", K_len" << li << " - 1u)";
        else e << "leaf_lut(" << r << ", " << r << "_m, as_global(K_lut" << li << "), K_len" << li << " - 1u)";
        break;
      case FDB_LEAF_CMP_I64:
        e << "cmp4<" << L.op << ", long long>((long long)" << r << "a.x, (long long)" << r << "a.y, (long long)" << r << "b.x, (long long)" << r << "b.y, K_lit" << li << ", " << r << "_m)";
        break;
      case FDB_LEAF_CMP_U64:
        e << "cmp4<" << L.op << ", unsigned long long>(" << r << "a.x, " << r << "a.y, " << r << "b.x, " << r << "b.y, (unsigned long long)K_lit" << li << ", " << r << "_m)";
        break;
      case FDB_LEAF_CMP_F64:
        e << "cmp4<" << L.op << ", double>(__longlong_as_double((long long)" << r << "a.x), __longlong_as_double((long long)" << r << "a.y), __longlong_as_double((long long)"
          << r << "b.x), __longlong_as_double((long long)" << r << "b.y), __longlong_as_double(K_lit" << li << "), " << r << "_m)";
        break;
      case FDB_LEAF_CMP_I64_F64:
        e << "cmp4<" << L.op << ", double>((double)(long long)" << r << "a.x, (double)(long long)" << r << "a.y, (double)(long long)" << r << "b.x, (double)(long long)" << r
          << "b.y, __longlong_as_double(K_lit" << li << "), " << r << "_m)";
        break;
    }
    return e.str();
  }

  std::string filter_expr() {  // postfix program → infix C expression
    return filter_expr_of(s.code, [&](int l) { return leaf_expr(l); });
  }
  template <typename F>
  static std::string filter_expr_of(const std::vector<uint8_t>& code, F leaf) {
    std::vector<std::string> st;
    for (uint8_t c : code) {
      if (c < 0x80) { st.push_back(leaf(c)); continue; }
      const std::string b = st.back(); st.pop_back();
      const std::string a = st.back(); st.pop_back();
      st.push_back("(" + a + (c == FDB_CODE_AND ? " & " : " | ") + b + ")");
    }
    return st.empty() ? "0xFu" : st.back();
  }

  std::string source() {
    const int BLK = s.block;
    o << "#include \"fdb_kernels.h\"\n" << kPreamble;
    o << "extern \"C\" __global__ __launch_bounds__(" << BLK << ") void fdb_plan_kernel(const FdbScanArgs* __restrict__ parts, const int n_parts, const long long total_tiles, const FdbScanArgs c) {\n";
    o << "  extern __shared__ __align__(16) unsigned char smem[];\n  const uint32_t tid = threadIdx.x;\n  const uint32_t n_slots = c.n_slots;\n";
    o << "  if (c.fill_state != nullptr)  // (a fresh table: its identity fill rides on this launch, see FdbScanArgs)\n";
    o << "    for (uint32_t i = blockIdx.x * " << BLK << "u + tid; i < c.fill_words; i += gridDim.x * " << BLK << "u) c.fill_state[i] = c.fill_idents[i / c.fill_alloc];\n";
    // (wave_tables: the same layout once per wave, TB bytes apart; a wave sees only its own)
    o << "  const size_t TB = (((size_t)n_slots * 4 + 15) & ~(size_t)15) + (size_t)n_slots * 8 * (size_t)c.n_aggs;\n  (void)TB;\n";
    const std::string tbl = s.wave_tables ? "smem + c.lds_lut_bytes + (size_t)(tid >> 6) * TB" : "smem + c.lds_lut_bytes";
    o << "  uint32_t* l_cnt = reinterpret_cast<uint32_t*>(" << tbl << ");\n";
    o << "  unsigned long long* l_acc = reinterpret_cast<unsigned long long*>(" << tbl << " + (((size_t)n_slots * 4 + 15) & ~(size_t)15));\n";
    if (s.lds_acc) {
      const std::string i0 = s.wave_tables ? "(tid & 63u)" : "tid", step = s.wave_tables ? "64" : std::to_string(BLK);
      o << "  for (uint32_t i = " << i0 << "; i < n_slots; i += " << step << ") l_cnt[i] = 0;\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const char* ident = A.func == FDB_AGG_MIN ? "0x7FFFFFFFFFFFFFFFull" : A.func == FDB_AGG_MAX ? "0x8000000000000000ull" : "0ull";
        o << "  for (uint32_t i = " << i0 << "; i < n_slots; i += " << step << ") l_acc[(size_t)" << j << " * n_slots + i] = " << ident << ";\n";
      }
    }
    if (s.cache) {
      o << "  const uint32_t CS = (uint32_t)c.cache_slots;\n";
      o << "  uint32_t* t_tag = reinterpret_cast<uint32_t*>(smem + c.lds_lut_bytes);\n  uint32_t* t_cnt = t_tag + CS;\n";
      o << "  unsigned long long* t_acc = reinterpret_cast<unsigned long long*>(t_cnt + CS);\n";
      o << "  for (uint32_t i = tid; i < CS; i += " << BLK << ") { t_tag[i] = 0u; t_cnt[i] = 0u; }\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const char* ident = A.func == FDB_AGG_MIN ? "0x7FFFFFFFFFFFFFFFull" : A.func == FDB_AGG_MAX ? "0x8000000000000000ull" : "0ull";
        o << "  for (uint32_t i = tid; i < CS; i += " << BLK << ") t_acc[(size_t)" << j << " * CS + i] = " << ident << ";\n";
      }
    }
    // per-record values, decoded when the workgroup enters a record
    auto decl_slots = [&](bool late) {
      const int n4 = late ? s.n_l4 : s.n_c4, n8 = late ? s.n_l8 : s.n_c8;
      for (int i = 0; i < n4; i++) o << "  const char* P_" << reg(false, late, i) << "_v = nullptr; const uint8_t* P_" << reg(false, late, i) << "_b = nullptr;\n";
      for (int i = 0; i < n8; i++) o << "  const char* P_" << reg(true, late, i) << "_v = nullptr; const uint8_t* P_" << reg(true, late, i) << "_b = nullptr;\n";
    };
    decl_slots(false); decl_slots(true);
    for (size_t l = 0; l < s.leaves.size(); l++)
      o << "  long long K_lit" << l << " = 0; uint32_t K_len" << l << " = 1, K_lds" << l << " = 0; int K_op" << l << " = 0; const uint8_t* K_lut" << l << " = nullptr;\n";
    for (size_t g = 0; g < s.gcols.size(); g++) o << "  uint32_t G_lds" << g << " = 0, G_stride" << g << " = 0; const uint32_t* G_lut" << g << " = nullptr;\n";
    for (size_t i = 0; i < s.exprs.size(); i++) if (s.exprs[i].kind == 1) o << "  long long K_elit" << i << " = 0;\n";
    for (int t = 0; t < s.reg_slots; t++) {  // the lane-private table
      o << "  unsigned long long R_cnt" << t << " = 0ull;\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "  double R_a" << j << "_" << t << " = 0.0;\n";
        else if (A.func == FDB_AGG_SUM) o << "  unsigned long long R_a" << j << "_" << t << " = 0ull;\n";
        else o << "  long long R_a" << j << "_" << t << " = " << (A.func == FDB_AGG_MIN ? "0x7FFFFFFFFFFFFFFFLL" : "(-0x7FFFFFFFFFFFFFFFLL - 1)") << ";\n";
      }
    }
    o << "  long long n_rows = 0, tile_begin = 0, tile_end = 0; int part = -1, lut_class = -1;\n";
    o << "  const uint32_t lane_off4 = tid * 16u, lane_off8 = tid * 32u, lane_offb = tid >> 1, lane_shb = (tid & 1u) * 4u;\n";
    o << "  for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {\n";
    o << "    if (part < 0 || tile >= tile_end) {\n      int np = part < 0 ? 0 : part;\n      while (np + 1 < n_parts && tile >= parts[np].tile_end) np++;\n      part = np;\n";
    o << "      const FdbScanArgs& pa = parts[part];\n      n_rows = pa.n_rows; tile_begin = pa.tile_begin; tile_end = pa.tile_end;\n";
    auto set_slots = [&](bool late) {
      const int n4 = late ? s.n_l4 : s.n_c4, n8 = late ? s.n_l8 : s.n_c8;
      const char* p4 = late ? "l4" : "c4"; const char* p8 = late ? "l8" : "c8";
      for (int i = 0; i < n4; i++)
        o << "      P_" << reg(false, late, i) << "_v = (const char*)pa." << p4 << "[" << i << "].values; P_" << reg(false, late, i) << "_b = pa." << p4 << "[" << i << "].validity;\n";
      for (int i = 0; i < n8; i++)
        o << "      P_" << reg(true, late, i) << "_v = (const char*)pa." << p8 << "[" << i << "].values; P_" << reg(true, late, i) << "_b = pa." << p8 << "[" << i << "].validity;\n";
    };
    set_slots(false); set_slots(true);
    for (size_t l = 0; l < s.leaves.size(); l++)
      o << "      K_lit" << l << " = pa.leaves[" << l << "].lit; K_len" << l << " = pa.leaves[" << l << "].lut_len; K_lds" << l << " = pa.leaves[" << l << "].lut_lds; K_op" << l
        << " = pa.leaves[" << l << "].op; K_lut" << l << " = pa.leaves[" << l << "].lut;\n";
    for (size_t g = 0; g < s.gcols.size(); g++)
      o << "      G_lds" << g << " = pa.gcols[" << g << "].lut_lds; G_stride" << g << " = pa.gcols[" << g << "].stride; G_lut" << g << " = pa.gcols[" << g << "].lut;\n";
    for (size_t i = 0; i < s.exprs.size(); i++) if (s.exprs[i].kind == 1) o << "      K_elit" << i << " = pa.expr[" << i << "].lit;\n";
    o << "      if (pa.lut_class != lut_class) {\n        lut_class = pa.lut_class;\n        __syncthreads();\n";
    for (size_t l = 0; l < s.leaves.size(); l++)
      if (s.leaves[l].kind == FDB_LEAF_DICT_LUT && s.leaves[l].lut_in_lds)
        o << "        for (uint32_t i = tid; i < K_len" << l << "; i += " << BLK << ") smem[K_lds" << l << " + i] = as_global(K_lut" << l << ")[i];\n";
    for (size_t g = 0; g < s.gcols.size(); g++)
      if (s.gcols[g].lut_in_lds)
        o << "        for (uint32_t i = tid; i < pa.gcols[" << g << "].lut_len; i += " << BLK << ") reinterpret_cast<uint32_t*>(smem + G_lds" << g << ")[i] = as_global(G_lut" << g << ")[i];\n";
    o << "        __syncthreads();\n      }\n    }\n";
    // tile
    o << "    const long long r0 = (tile - tile_begin) * " << BLK * 4 << "LL;\n";
    o << "    const long long left = n_rows - r0 - (long long)tid * 4;\n    if (left <= 0) continue;\n";
    o << "    const size_t tile_off4 = (size_t)r0 * 4, tile_off8 = (size_t)r0 * 8, tile_offb = (size_t)(r0 >> 3);\n";
    loads(false);
    o << "    uint32_t sel = left >= 4 ? 0xFu : ((1u << (int)left) - 1u);\n";
    if (!s.code.empty()) o << "    sel &= " << filter_expr() << ";\n";
    {  // match bits of the leaves that boolean projections read (expression nodes of kind 7): leaves outside the filter program
      std::vector<int> seen;
      for (const JitExprNode& e : s.exprs)
        if (e.kind == 7 && std::find(seen.begin(), seen.end(), e.slot) == seen.end()) { seen.push_back(e.slot); o << "    const uint32_t PM" << e.slot << " = " << leaf_expr(e.slot) << "; (void)PM" << e.slot << ";\n"; }
    }
    // Sorted input (a table sorted by its label columns is FrostDB's normal case): every selected row of a wave falls into ONE slot, and
    // 256 LDS atomics on one address are 256 serialised updates (cfg 2's query over a table sorted by labels.path: 0.40 ms per 100 M rows
    // against 0.24 unsorted). Single-phase shapes therefore keep the wave together (a lane without selected rows stays, its rows
    // predicated off as they are anyway) and test whether the slots of the wave's selected rows agree: if so every lane folds its rows,
    // a butterfly folds the lanes, ONE lane updates the table. Unsorted input pays the test — a ballot, a readlane, four compares, a
    // ballot per tile. ($FDB_NO_UNIFORM_FOLD: A/B aid)
    const bool uniform_fold = s.uniform_fold && s.lds_acc && !s.cache && s.reg_slots == 0 && !s.wave_tables && !s.two_phase && !s.gcols.empty();
    if (uniform_fold) o << "    const unsigned long long has = __ballot(sel != 0u);\n    if (has == 0ull) continue;\n";
    else o << "    if (sel == 0u) continue;\n";
    loads(true);
    // group slot
    o << "    uint32_t gid0 = 0, gid1 = 0, gid2 = 0, gid3 = 0;\n";
    for (size_t g = 0; g < s.gcols.size(); g++) {
      const JitGroup& G = s.gcols[g];
      const std::string r = reg(false, s.two_phase, G.slot);
      const std::string lut = G.lut_in_lds ? ("reinterpret_cast<const uint32_t*>(smem + G_lds" + std::to_string(g) + ")") : ("as_global(G_lut" + std::to_string(g) + ")");
      const char* comp[4] = {"x", "y", "z", "w"};
      for (int k = 0; k < 4; k++)
        o << "    gid" << k << " += ((" << r << "_m >> " << k << ") & 1u ? " << lut << "[" << r << "." << comp[k] << "] : 0u) * G_stride" << g << ";\n";
    }
    if (uniform_fold) {
      auto raw_of = [&](size_t j, int k) {
        const JitAgg& A = s.aggs[j];
        const std::string comp = std::string(k < 2 ? "a" : "b") + (k % 2 == 0 ? ".x" : ".y");
        if (A.expr != 0) {
          auto col = [&](int ni) -> std::string { if (s.exprs[(size_t)ni].kind == 7) return "((PM" + std::to_string(s.exprs[(size_t)ni].slot) + " >> " + std::to_string(k) + ") & 1u)"; return reg(true, s.two_phase, s.exprs[(size_t)ni].slot) + comp; };
          auto colvalid = [&](int ni) { return "((" + reg(true, s.two_phase, s.exprs[(size_t)ni].slot) + "_m >> " + std::to_string(k) + ") & 1u)"; };
          return "(" + expr_valid(s.exprs, A.expr - 1, col, colvalid) + " ? " + expr_bits(s.exprs, A.expr - 1, expr_value(s.exprs, A.expr - 1, col, colvalid)) + " : 0ull)";
        }
        const std::string r = reg(true, s.two_phase, A.slot);
        return "((" + r + "_m >> " + std::to_string(k) + ") & 1u ? " + r + comp + " : c.aggs[" + std::to_string(j) + "].null_value)";
      };
      // (quick reject first: the lanes' FIRST selected rows against one of them — unsorted input leaves here)
      o << "    {\n      const uint32_t my_first = (sel & 1u) ? gid0 : (sel & 2u) ? gid1 : (sel & 4u) ? gid2 : gid3;\n";
      o << "      const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)my_first, __builtin_ctzll(has));\n";
      o << "      if (__ballot(sel != 0u && my_first != g0) == 0ull && __ballot(1) == ~0ull &&\n";
      o << "          __ballot(((sel & 1u) && gid0 != g0) || ((sel & 2u) && gid1 != g0) || ((sel & 4u) && gid2 != g0) || ((sel & 8u) && gid3 != g0)) == 0ull) {\n";
      o << "        uint32_t u_cnt = (uint32_t)__builtin_popcount(sel);\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const std::string v = "u" + std::to_string(j);
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) {
          o << "        double " << v << " = 0.0;\n";
          for (int k = 0; k < 4; k++) o << "        if ((sel >> " << k << ") & 1u) " << v << " += __longlong_as_double((long long)" << raw_of(j, k) << ");\n";
        } else if (A.func == FDB_AGG_SUM) {
          o << "        unsigned long long " << v << " = 0ull;\n";
          for (int k = 0; k < 4; k++) o << "        if ((sel >> " << k << ") & 1u) " << v << " += " << raw_of(j, k) << ";\n";
        } else {
          o << "        long long " << v << " = " << (A.func == FDB_AGG_MIN ? "0x7FFFFFFFFFFFFFFFLL" : "(-0x7FFFFFFFFFFFFFFFLL - 1)") << ";\n";
          for (int k = 0; k < 4; k++) {
            const std::string key = A.type == FDB_T_F64 ? ("f64_minmax_key(__longlong_as_double((long long)" + raw_of(j, k) + "), " + (A.func == FDB_AGG_MIN ? "true" : "false") + ")") : ("(long long)" + raw_of(j, k));
            o << "        if ((sel >> " << k << ") & 1u) { const long long y = " << key << "; " << v << " = " << (A.func == FDB_AGG_MIN ? "y < " : "y > ") << v << " ? y : " << v << "; }\n";
          }
        }
      }
      o << "#pragma unroll\n        for (int sh = 32; sh > 0; sh >>= 1) {\n          u_cnt += (uint32_t)__shfl_xor((int)u_cnt, sh, 64);\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const std::string v = "u" + std::to_string(j);
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "          " << v << " += __shfl_xor(" << v << ", sh, 64);\n";
        else if (A.func == FDB_AGG_SUM) o << "          " << v << " += (unsigned long long)__shfl_xor((long long)" << v << ", sh, 64);\n";
        else o << "          { const long long y = __shfl_xor(" << v << ", sh, 64); " << v << " = " << (A.func == FDB_AGG_MIN ? "y < " : "y > ") << v << " ? y : " << v << "; }\n";
      }
      o << "        }\n        if ((tid & 63u) == 0u) {\n";
      if (s.need_count) o << "          atomicAdd(&l_cnt[g0], u_cnt);\n";
      else o << "          l_cnt[g0] = 1u;\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const std::string v = "u" + std::to_string(j);
        const std::string acc = "(l_acc + (size_t)" + std::to_string(j) + " * n_slots + g0)";
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "          atomicAdd(reinterpret_cast<double*>" << acc << ", " << v << ");\n";
        else if (A.func == FDB_AGG_SUM) o << "          atomicAdd(" << acc << ", " << v << ");\n";
        else o << "          " << (A.func == FDB_AGG_MIN ? "atomicMin" : "atomicMax") << "(reinterpret_cast<long long*>" << acc << ", " << v << ");\n";
      }
      o << "        }\n        continue;\n      }\n    }\n";
    }
    // accumulate, row by row (one divergent region per row, every aggregate inside it)
    for (int k = 0; k < 4 && s.reg_slots > 0; k++) {  // lane-private table: predicated updates, no memory traffic at all
      const std::string comp = std::string(k < 2 ? "a" : "b") + (k % 2 == 0 ? ".x" : ".y");
      o << "    {\n      const bool on = (sel >> " << k << ") & 1u;\n";
      std::vector<std::string> val(s.aggs.size());
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        std::string raw;
        if (A.expr != 0) {
          auto col = [&](int ni) -> std::string { if (s.exprs[(size_t)ni].kind == 7) return "((PM" + std::to_string(s.exprs[(size_t)ni].slot) + " >> " + std::to_string(k) + ") & 1u)"; return reg(true, s.two_phase, s.exprs[(size_t)ni].slot) + comp; };
          auto colvalid = [&](int ni) { return "((" + reg(true, s.two_phase, s.exprs[(size_t)ni].slot) + "_m >> " + std::to_string(k) + ") & 1u)"; };
          raw = "(" + expr_valid(s.exprs, A.expr - 1, col, colvalid) + " ? " + expr_bits(s.exprs, A.expr - 1, expr_value(s.exprs, A.expr - 1, col, colvalid)) + " : 0ull)";
        } else {
          const std::string r = reg(true, s.two_phase, A.slot);
          raw = "((" + r + "_m >> " + std::to_string(k) + ") & 1u ? " + r + comp + " : c.aggs[" + std::to_string(j) + "].null_value)";
        }
        const std::string v = "v" + std::to_string(j);
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "      const double " << v << " = __longlong_as_double((long long)" << raw << ");\n";
        else if (A.func == FDB_AGG_SUM) o << "      const unsigned long long " << v << " = " << raw << ";\n";
        else o << "      const long long " << v << " = " << (A.type == FDB_T_F64 ? ("f64_minmax_key(__longlong_as_double((long long)" + raw + "), " + (A.func == FDB_AGG_MIN ? "true" : "false") + ")") : ("(long long)" + raw)) << ";\n";
        val[j] = v;
      }
      for (int t = 0; t < s.reg_slots; t++) {
        const std::string m = s.reg_slots == 1 ? std::string("on") : ("(on && gid" + std::to_string(k) + " == " + std::to_string(t) + "u)");
        o << "      R_cnt" << t << " += " << m << " ? 1ull : 0ull;\n";
        for (size_t j = 0; j < s.aggs.size(); j++) {
          const JitAgg& A = s.aggs[j];
          if (A.func == FDB_AGG_COUNT) continue;
          const std::string acc = "R_a" + std::to_string(j) + "_" + std::to_string(t);
          if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "      " << acc << " += " << m << " ? " << val[j] << " : 0.0;\n";
          else if (A.func == FDB_AGG_SUM) o << "      " << acc << " += " << m << " ? " << val[j] << " : 0ull;\n";
          else if (A.func == FDB_AGG_MIN) o << "      " << acc << " = (" << m << " && " << val[j] << " < " << acc << ") ? " << val[j] << " : " << acc << ";\n";
          else o << "      " << acc << " = (" << m << " && " << val[j] << " > " << acc << ") ? " << val[j] << " : " << acc << ";\n";
        }
      }
      o << "    }\n";
    }
    for (int k = 0; k < 4 && s.reg_slots == 0; k++) {
      const std::string gidk = "gid" + std::to_string(k);
      o << "    if ((sel >> " << k << ") & 1u) {\n";
      if (s.cache) {
        // cached = this row's slot owns (or just claimed) its direct-mapped place in the workgroup's combining cache
        o << "      const uint32_t ch = ((gid" << k << " * 0x9E3779B1u) >> 12) & (CS - 1u);\n";
        o << "      uint32_t tg = t_tag[ch];\n      if (tg == 0u) { tg = atomicCAS(&t_tag[ch], 0u, gid" << k << " + 1u); if (tg == 0u) tg = gid" << k << " + 1u; }\n";
        o << "      const bool cached = tg == gid" << k << " + 1u;\n";
        o << "      if (cached) atomicAdd(&t_cnt[ch], 1u);\n";
        if (s.need_count) o << "      else atomicAdd(&c.cnt[gid" << k << "], 1ull);\n";
        else o << "      else c.cnt[gid" << k << "] = 1ull;\n";  // occupancy flag only: a plain store (every writer stores the same value)
      } else if (s.lds_acc) {
        if (s.need_count) o << "      atomicAdd(&l_cnt[" << gidk << "], 1u);\n";
        else o << "      l_cnt[" << gidk << "] = 1u;\n";
      } else {
        o << "      atomicAdd(&c.cnt[gid" << k << "], 1ull);\n";
      }
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const std::string comp = std::string(k < 2 ? "a" : "b") + (k % 2 == 0 ? ".x" : ".y");
        std::string raw;
        if (A.expr != 0) {  // computed input: the expression over this row's raw column values; a NULL (÷ 0) adds the zero slot
          auto col = [&](int ni) -> std::string { if (s.exprs[(size_t)ni].kind == 7) return "((PM" + std::to_string(s.exprs[(size_t)ni].slot) + " >> " + std::to_string(k) + ") & 1u)"; return reg(true, s.two_phase, s.exprs[(size_t)ni].slot) + comp; };
          auto colvalid = [&](int ni) { return "((" + reg(true, s.two_phase, s.exprs[(size_t)ni].slot) + "_m >> " + std::to_string(k) + ") & 1u)"; };
          raw = "(" + expr_valid(s.exprs, A.expr - 1, col, colvalid) + " ? " + expr_bits(s.exprs, A.expr - 1, expr_value(s.exprs, A.expr - 1, col, colvalid)) + " : 0ull)";
        } else {
          const std::string r = reg(true, s.two_phase, A.slot);
          raw = "((" + r + "_m >> " + std::to_string(k) + ") & 1u ? " + r + comp + " : c.aggs[" + std::to_string(j) + "].null_value)";
        }
        const std::string gacc = "(c.aggs[" + std::to_string(j) + "].acc + gid" + std::to_string(k) + ")";
        const std::string acc = s.lds_acc ? ("(l_acc + (size_t)" + std::to_string(j) + " * n_slots + " + gidk + ")") : gacc;
        auto emit = [&](const std::string& where, const char* ind) {
          if (A.func == FDB_AGG_SUM) {
            if (A.type == FDB_T_F64) o << ind << "atomicAdd(reinterpret_cast<double*>" << where << ", __longlong_as_double((long long)" << raw << "));\n";
            else o << ind << "atomicAdd(" << where << ", " << raw << ");\n";
          } else {
            const std::string key = A.type == FDB_T_F64 ? ("f64_minmax_key(__longlong_as_double((long long)" + raw + "), " + (A.func == FDB_AGG_MIN ? "true" : "false") + ")") : ("(long long)" + raw);
            o << ind << (A.func == FDB_AGG_MIN ? "atomicMin" : "atomicMax") << "(reinterpret_cast<long long*>" << where << ", " << key << ");\n";
          }
        };
        if (s.cache) {
          o << "      if (cached) {\n";
          emit("(t_acc + (size_t)" + std::to_string(j) + " * CS + ch)", "        ");
          o << "      } else {\n";
          emit(gacc, "        ");
          o << "      }\n";
        } else {
          emit(acc, "      ");
        }
      }
      o << "    }\n";
    }
    o << "  }\n";
    if (s.reg_slots > 0) {
      // lane-private tables → one value per wave (butterfly over the 64 lanes) → one update per wave and slot
      o << "  for (int sh = 32; sh > 0; sh >>= 1) {\n";
      for (int t = 0; t < s.reg_slots; t++) {
        o << "    R_cnt" << t << " += (unsigned long long)__shfl_xor((long long)R_cnt" << t << ", sh, 64);\n";
        for (size_t j = 0; j < s.aggs.size(); j++) {
          const JitAgg& A = s.aggs[j];
          if (A.func == FDB_AGG_COUNT) continue;
          const std::string acc = "R_a" + std::to_string(j) + "_" + std::to_string(t);
          if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "    " << acc << " += __shfl_xor(" << acc << ", sh, 64);\n";
          else if (A.func == FDB_AGG_SUM) o << "    " << acc << " += (unsigned long long)__shfl_xor((long long)" << acc << ", sh, 64);\n";
          else if (A.func == FDB_AGG_MIN) o << "    { const long long y = __shfl_xor(" << acc << ", sh, 64); " << acc << " = y < " << acc << " ? y : " << acc << "; }\n";
          else o << "    { const long long y = __shfl_xor(" << acc << ", sh, 64); " << acc << " = y > " << acc << " ? y : " << acc << "; }\n";
        }
      }
      o << "  }\n  if ((tid & 63u) == 0u) {\n";
      for (int t = 0; t < s.reg_slots; t++) {
        o << "    if (R_cnt" << t << " != 0ull) {\n";
        if (s.lds_acc) o << "      atomicAdd(&l_cnt[" << t << "], (uint32_t)(R_cnt" << t << " > 0xFFFFFFFFull ? 0xFFFFFFFFull : R_cnt" << t << "));\n";
        else o << "      atomicAdd(&c.cnt[" << t << "], R_cnt" << t << ");\n";
        for (size_t j = 0; j < s.aggs.size(); j++) {
          const JitAgg& A = s.aggs[j];
          if (A.func == FDB_AGG_COUNT) continue;
          const std::string v = "R_a" + std::to_string(j) + "_" + std::to_string(t);
          const std::string acc = s.lds_acc ? ("(l_acc + (size_t)" + std::to_string(j) + " * n_slots + " + std::to_string(t) + ")") : ("(c.aggs[" + std::to_string(j) + "].acc + " + std::to_string(t) + ")");
          if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "      atomicAdd(reinterpret_cast<double*>" << acc << ", " << v << ");\n";
          else if (A.func == FDB_AGG_SUM) o << "      atomicAdd(" << acc << ", " << v << ");\n";
          else o << "      " << (A.func == FDB_AGG_MIN ? "atomicMin" : "atomicMax") << "(reinterpret_cast<long long*>" << acc << ", " << v << ");\n";
        }
        o << "    }\n";
      }
      o << "  }\n";
    }
    if (s.cache) {  // the combining cache goes to the global table: one atomic per entry and aggregate
      o << "  __syncthreads();\n  for (uint32_t i = tid; i < CS; i += " << BLK << ") {\n    const uint32_t tg = t_tag[i];\n    if (tg == 0u) continue;\n    const uint32_t g = tg - 1u;\n";
      if (s.need_count) o << "    atomicAdd(&c.cnt[g], (unsigned long long)t_cnt[i]);\n";
      else o << "    c.cnt[g] = 1ull;\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const std::string v = "t_acc[(size_t)" + std::to_string(j) + " * CS + i]";
        const std::string dst = "c.aggs[" + std::to_string(j) + "].acc + g";
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "    atomicAdd(reinterpret_cast<double*>(" << dst << "), __longlong_as_double((long long)" << v << "));\n";
        else if (A.func == FDB_AGG_SUM) o << "    atomicAdd(" << dst << ", " << v << ");\n";
        else o << "    " << (A.func == FDB_AGG_MIN ? "atomicMin" : "atomicMax") << "(reinterpret_cast<long long*>(" << dst << "), (long long)" << v << ");\n";
      }
      o << "  }\n";
    }
    // flush
    if (s.lds_acc && s.wave_tables) {
      // the waves' tables, added up in wave order: the workgroup's partial table (the fold kernel adds the workgroups' in workgroup order)
      o << "  __syncthreads();\n  {\n    unsigned long long* out = c.partials + (size_t)blockIdx.x * (size_t)(1 + c.n_aggs) * n_slots;\n";
      o << "    const unsigned char* t0 = smem + c.lds_lut_bytes;\n";
      o << "    for (uint32_t i = tid; i < n_slots; i += " << BLK << ") {\n      unsigned long long n = 0;\n";
      o << "      for (int w = 0; w < " << BLK / 64 << "; w++) n += reinterpret_cast<const uint32_t*>(t0 + (size_t)w * TB)[i];\n      out[i] = n;\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const std::string at = "reinterpret_cast<const unsigned long long*>(t0 + (size_t)w * TB + (((size_t)n_slots * 4 + 15) & ~(size_t)15))[(size_t)" + std::to_string(j) + " * n_slots + i]";
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) {
          o << "      { double v = 0.0; for (int w = 0; w < " << BLK / 64 << "; w++) v += __longlong_as_double((long long)" << at << "); out[(size_t)" << (1 + j) << " * n_slots + i] = (unsigned long long)__double_as_longlong(v); }\n";
        } else if (A.func == FDB_AGG_SUM) {
          o << "      { unsigned long long v = 0; for (int w = 0; w < " << BLK / 64 << "; w++) v += " << at << "; out[(size_t)" << (1 + j) << " * n_slots + i] = v; }\n";
        } else {
          o << "      { long long v = " << (A.func == FDB_AGG_MIN ? "0x7FFFFFFFFFFFFFFFLL" : "(-0x7FFFFFFFFFFFFFFFLL - 1)") << "; for (int w = 0; w < " << BLK / 64 << "; w++) { const long long y = (long long)" << at
            << "; v = " << (A.func == FDB_AGG_MIN ? "y < v" : "y > v") << " ? y : v; } out[(size_t)" << (1 + j) << " * n_slots + i] = (unsigned long long)v; }\n";
        }
      }
      o << "    }\n  }\n";
    } else if (s.lds_acc) {
      o << "  __syncthreads();\n";
      o << "  if (c.partials != nullptr) {\n    unsigned long long* out = c.partials + (size_t)blockIdx.x * (size_t)(1 + c.n_aggs) * n_slots;\n";
      o << "    for (uint32_t i = tid; i < n_slots; i += " << BLK << ") out[i] = (unsigned long long)l_cnt[i];\n";
      for (size_t j = 0; j < s.aggs.size(); j++)
        if (s.aggs[j].func != FDB_AGG_COUNT)
          o << "    for (uint32_t i = tid; i < n_slots; i += " << BLK << ") out[(size_t)" << (1 + j) << " * n_slots + i] = l_acc[(size_t)" << j << " * n_slots + i];\n";
      o << "  } else {\n    for (uint32_t i = tid; i < n_slots; i += " << BLK << ") {\n      const uint32_t n_sel = l_cnt[i];\n      if (n_sel == 0) continue;\n";
      o << "      atomicAdd(&c.cnt[i], (unsigned long long)n_sel);\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        const std::string v = "l_acc[(size_t)" + std::to_string(j) + " * n_slots + i]";
        const std::string dst = "c.aggs[" + std::to_string(j) + "].acc + i";
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "      atomicAdd(reinterpret_cast<double*>(" << dst << "), __longlong_as_double((long long)" << v << "));\n";
        else if (A.func == FDB_AGG_SUM) o << "      atomicAdd(" << dst << ", " << v << ");\n";
        else o << "      " << (A.func == FDB_AGG_MIN ? "atomicMin" : "atomicMax") << "(reinterpret_cast<long long*>(" << dst << "), (long long)" << v << ");\n";
      }
      o << "    }\n  }\n";
    }
    o << "}\n";
    return o.str();
  }

  // ---- filter(): the selection bitmap of every record of a scan, one launch (Plan::filter_batches) --------------------------------
  // Record / LUT handling as in fdb_plan_kernel; the geometry is the compaction's: a WAVE owns a tile of 2 048 rows
  // (FDB_COMPACT_TILE; 8 steps of 64 lanes × 4 consecutive rows), a 256-thread workgroup four consecutive tiles of one record
  // (the unit `tile_begin` / `tile_end` of the argument blocks count here). No atomics and nothing to zero: the wave writes all
  // 64 mask words of its tile and the tile's count with plain stores (a first version added every wave's count to one counter
  // per 1 024-tile block: thousands of atomics on one address serialise in the L2 — 1.7 ms per 100 M rows instead of 0.2).
  // No branches in the row loop either — lanes past the end of the record read its last rows again and mask the result — so the
  // compiler issues the loads of several steps before it consumes the first.
  std::string flags_source() {
    const int BLK = 256;
    o << "#include \"fdb_kernels.h\"\n" << kPreamble;
    o << "extern \"C\" __global__ __launch_bounds__(" << BLK << ") void fdb_flags_kernel(const FdbScanArgs* __restrict__ parts, const int n_parts, const long long total_tiles, const FdbScanArgs c, uint32_t* __restrict__ masks, uint32_t* __restrict__ tile_counts) {\n";
    o << "  extern __shared__ __align__(16) unsigned char smem[];\n  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;\n";
    for (int i = 0; i < s.n_c4; i++) o << "  const char* P_" << reg(false, false, i) << "_v = nullptr; const uint8_t* P_" << reg(false, false, i) << "_b = nullptr;\n";
    for (int i = 0; i < s.n_c8; i++) o << "  const char* P_" << reg(true, false, i) << "_v = nullptr; const uint8_t* P_" << reg(true, false, i) << "_b = nullptr;\n";
    for (size_t l = 0; l < s.leaves.size(); l++)
      o << "  long long K_lit" << l << " = 0; uint32_t K_len" << l << " = 1, K_lds" << l << " = 0; int K_op" << l << " = 0; const uint8_t* K_lut" << l << " = nullptr;\n";
    o << "  long long n_rows = 0, tile_begin = 0, tile_end = 0, out_tile = 0; int part = -1, lut_class = -1;\n";
    o << "  for (long long st = blockIdx.x; st < total_tiles; st += gridDim.x) {\n";
    o << "    if (part < 0 || st >= tile_end) {\n      int np = part < 0 ? 0 : part;\n      while (np + 1 < n_parts && st >= parts[np].tile_end) np++;\n      part = np;\n";
    o << "      const FdbScanArgs& pa = parts[part];\n      n_rows = pa.n_rows; tile_begin = pa.tile_begin; tile_end = pa.tile_end; out_tile = pa.out_tile_base;\n";
    for (int i = 0; i < s.n_c4; i++)
      o << "      P_" << reg(false, false, i) << "_v = (const char*)pa.c4[" << i << "].values; P_" << reg(false, false, i) << "_b = pa.c4[" << i << "].validity;\n";
    for (int i = 0; i < s.n_c8; i++)
      o << "      P_" << reg(true, false, i) << "_v = (const char*)pa.c8[" << i << "].values; P_" << reg(true, false, i) << "_b = pa.c8[" << i << "].validity;\n";
    for (size_t l = 0; l < s.leaves.size(); l++)
      o << "      K_lit" << l << " = pa.leaves[" << l << "].lit; K_len" << l << " = pa.leaves[" << l << "].lut_len; K_lds" << l << " = pa.leaves[" << l << "].lut_lds; K_op" << l
        << " = pa.leaves[" << l << "].op; K_lut" << l << " = pa.leaves[" << l << "].lut;\n";
    o << "      if (pa.lut_class != lut_class) {\n        lut_class = pa.lut_class;\n        __syncthreads();\n";
    for (size_t l = 0; l < s.leaves.size(); l++)
      if (s.leaves[l].kind == FDB_LEAF_DICT_LUT && s.leaves[l].lut_in_lds)
        o << "        for (uint32_t i = tid; i < K_len" << l << "; i += " << BLK << ") smem[K_lds" << l << " + i] = as_global(K_lut" << l << ")[i];\n";
    o << "        __syncthreads();\n      }\n    }\n";
    o << "    const long long ltile = (st - tile_begin) * 4 + wave;  // this wave's tile inside the record\n";
    o << "    const long long rbase = ltile * 2048LL;\n    if (rbase >= n_rows) continue;\n";
    o << "    const long long last_group = (n_rows - 1) & ~3LL;\n";
    // the 4-row evaluation, as a function of the (clamped) first row of the lane's group: the same load / leaf code as the scan kernel
    o << "    auto eval = [&](const long long row) -> uint32_t {\n";
    o << "      const size_t tile_off4 = (size_t)row * 4, tile_off8 = (size_t)row * 8, tile_offb = (size_t)(row >> 3);\n";
    o << "      const uint32_t lane_off4 = 0u, lane_off8 = 0u, lane_offb = 0u, lane_shb = (uint32_t)row & 4u;\n";
    o << "      (void)tile_off4; (void)tile_off8; (void)tile_offb; (void)lane_off4; (void)lane_off8; (void)lane_offb; (void)lane_shb;\n";
    loads(false);
    o << "      return " << (s.code.empty() ? std::string("0xFu") : filter_expr()) << ";\n    };\n";
    o << "    uint32_t cnt = 0;\n";
    // four steps at a time: first every load and compare (straight-line code: the loads of all four are in flight together),
    // then the words and the counts
    o << "#pragma unroll 1\n    for (int q0 = 0; q0 < 8; q0 += 4) {\n      uint32_t sel[4];\n";
    o << "#pragma unroll\n      for (int u = 0; u < 4; u++) {\n";
    o << "        const long long row = rbase + (q0 + u) * 256 + (long long)lane * 4;\n        const long long left = n_rows - row;\n";
    o << "        sel[u] = eval(left > 0 ? row : last_group) & (left >= 4 ? 0xFu : left > 0 ? ((1u << (int)left) - 1u) : 0u);\n      }\n";
    // a step's 256 mask bits = 8 words: lane L's nibble sits at bits 4 (L % 8) of word L / 8
    o << "#pragma unroll\n      for (int u = 0; u < 4; u++) {\n        uint32_t w = sel[u];\n";
    o << "        w |= (uint32_t)__shfl_down((int)w, 1, 64) << 4;\n        w |= (uint32_t)__shfl_down((int)w, 2, 64) << 8;\n        w |= (uint32_t)__shfl_down((int)w, 4, 64) << 16;\n";
    o << "        sel[u] = w;\n";
    for (int r = 0; r < 4; r++) o << "        cnt += (uint32_t)__popcll(__ballot((w >> " << r << ") & 1u));\n";
    o << "      }\n";
    // one store per lane group for the four steps: lanes with L % 8 == 0 hold the words of their group
    o << "      if ((lane & 7u) == 0u) {\n        uint32_t* mw = masks + (size_t)(out_tile + ltile) * 64 + q0 * 8 + (lane >> 3);\n";
    o << "        mw[0] = sel[0]; mw[8] = sel[1]; mw[16] = sel[2]; mw[24] = sel[3];\n      }\n";
    o << "    }\n";
    o << "    if (lane == 0u) tile_counts[out_tile + ltile] = cnt;\n";
    o << "  }\n}\n";
    return o.str();
  }

  // ---- filter() in one pass over the filter columns (Plan::filter_batches; FdbSelectArgs in fdb_kernels.h) ----------------------------
  // fdb_flags_kernel's row evaluation with three additions. (1) Work is handed out by a ticket counter, one ticket per workgroup share
  // of four tiles: whatever a workgroup waits for is then held by workgroups that already run, never by one the dispatcher has not
  // placed yet (several filter() scans, or scans of other plans, share the GPU). (2) The selected values of the fused slots are staged
  // in the wave's LDS region at their local positions while the predicate is evaluated — the values are in registers at that point,
  // the filter column is not read again. (3) The workgroup publishes the count of its share and learns the share's place — the
  // exclusive prefix of the counts inside the record — from the launch's SCANNER (the workgroup that arrived first: see below);
  // every wave then writes its staged values to their final place, consecutive lanes consecutive 16 bytes; tile offsets are left
  // behind for compact_multi_kernel. The next share's loads are issued before the wait (`load` and `pred` are generated apart).
  // Geometry: 512-thread workgroups, a wave owns HALF a tile (1 024 rows = 4 steps → 8 KiB of LDS per 8-byte column and wave, 16 waves
  // per CU). How the places are found went through four versions: a decoupled look-back with a status word per
  // wave (every unit evaluated within one look-back's duration is "count only", the nearest inclusive prefix thousands of words away:
  // 0.2 ms of evaluation became 0.5); one status per share and 256 words per poll (512 workgroups polling the same sixteen cache lines:
  // the polls set the pace); a ticket per four shares (a wave that places one unit before it has counted the next chains the whole
  // launch behind its predecessor: 43 ms — a workgroup publishes every count BEFORE it waits for anything); the scanner (0.30 ms).
  struct Fused { bool wide; int slot; size_t off; };
  static constexpr int kSelectBlock = 512, kSelectUnit = FDB_COMPACT_TILE / 2;
  std::vector<Fused> fused(size_t* per_wave) const {
    std::vector<Fused> f;
    size_t at = 0;
    for (int i = 0; i < s.n_c8; i++) if ((s.fuse8 >> i) & 1) { f.push_back({true, i, at}); at += (size_t)kSelectUnit * 8; }
    for (int i = 0; i < s.n_c4; i++) if ((s.fuse4 >> i) & 1) { f.push_back({false, i, at}); at += (size_t)kSelectUnit * 4; }
    if (per_wave != nullptr) *per_wave = at;
    return f;
  }
  std::string select_source() {
    const int BLK = kSelectBlock;
    size_t per_wave = 0;
    const std::vector<Fused> fz = fused(&per_wave);
    // ($FDB_SELECT_ABLATE, tuning aid — results are WRONG: 1 = no look-back (every share at offset 0), 2 = staged values are not written out, 3 = nothing is staged either)
    const int ablate = std::getenv("FDB_SELECT_ABLATE") ? std::atoi(std::getenv("FDB_SELECT_ABLATE")) : 0;
    const int w_sleep = std::getenv("FDB_SELECT_SLEEP") ? std::atoi(std::getenv("FDB_SELECT_SLEEP")) : 4, s_sleep = std::getenv("FDB_SELECT_SCAN_SLEEP") ? std::atoi(std::getenv("FDB_SELECT_SCAN_SLEEP")) : 4;  // (tuning aids)
    o << "#include \"fdb_kernels.h\"\n" << kPreamble;
    // what one step (4 rows per lane) of the filter columns looks like in registers: filled by `load`, consumed by `pred` — apart, so that
    // the NEXT share's loads are in flight while the workgroup waits for the current share's place
    std::vector<std::pair<std::string, std::string>> raw_fields;  // (type, name)
    for (int i = 0; i < s.n_c4; i++) {
      const std::string r = reg(false, false, i);
      if (s.c4[i].has_values) raw_fields.push_back({"u32x4", r});
      raw_fields.push_back({"uint32_t", r + "_m"});
    }
    for (int i = 0; i < s.n_c8; i++) {
      const std::string r = reg(true, false, i);
      if (s.c8[i].has_values) { raw_fields.push_back({"u64x2", r + "a"}); raw_fields.push_back({"u64x2", r + "b"}); }
      raw_fields.push_back({"uint32_t", r + "_m"});
    }
    o << "struct Raw {";
    for (auto& f : raw_fields) o << " " << f.first << " " << f.second << ";";
    o << " int none; };\n";
    o << "__device__ __forceinline__ uint32_t lanes_below(const unsigned long long b) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u)); }\n";
    o << "extern \"C\" __global__ __launch_bounds__(" << BLK << ") void fdb_select_kernel(const FdbScanArgs* __restrict__ parts, const int n_parts, const long long total_tiles, const FdbScanArgs c, uint32_t* __restrict__ masks, uint32_t* __restrict__ offsets, const FdbSelectArgs sa) {\n";
    o << "  extern __shared__ __align__(16) unsigned char smem[];\n  __shared__ long long s_ticket[2];\n  __shared__ unsigned long long s_base;\n  __shared__ uint32_t s_cnt[2][8];\n";
    o << "  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;\n";
    // (FdbSelectArgs::zero: every workgroup — the scanner too — clears its share of the regions the NEXT launch accumulates into)
    o << "  for (int zr = 0; zr < sa.n_zero; zr++) {\n    u32x4* zp = reinterpret_cast<u32x4*>(sa.zero[2 * zr]);\n    const long long zq = (long long)(sa.zero[2 * zr + 1] >> 4);\n";
    o << "    for (long long zi = (long long)blockIdx.x * " << BLK << " + tid; zi < zq; zi += (long long)gridDim.x * " << BLK << ") zp[zi] = u32x4{0u, 0u, 0u, 0u};\n  }\n";
    for (int i = 0; i < s.n_c4; i++) o << "  const char* P_" << reg(false, false, i) << "_v = nullptr; const uint8_t* P_" << reg(false, false, i) << "_b = nullptr;\n";
    for (int i = 0; i < s.n_c8; i++) o << "  const char* P_" << reg(true, false, i) << "_v = nullptr; const uint8_t* P_" << reg(true, false, i) << "_b = nullptr;\n";
    for (size_t l = 0; l < s.leaves.size(); l++)
      o << "  long long K_lit" << l << " = 0; uint32_t K_len" << l << " = 1, K_lds" << l << " = 0; int K_op" << l << " = 0; const uint8_t* K_lut" << l << " = nullptr;\n";
    for (size_t k = 0; k < fz.size(); k++) o << "  char* D_" << k << " = nullptr;\n";
  
    o << "  long long n_rows = 0, tile_begin = 0, tile_end = 0, out_tile = 0; int part = -1, lut_class = -1;\n";
    o << "  unsigned char* const stg = smem + sa.stage_off + (size_t)wave * " << per_wave << "u;\n  (void)stg;\n";
    o << "  unsigned long long* const status = sa.ctl + sa.status_off;\n";
    o << "  const unsigned long long ep = (unsigned long long)sa.epoch << 40, VAL = (1ull << 38) - 1ull;\n";
    o << "  auto load = [&](const long long row, Raw& raw) {\n";
    o << "      const size_t tile_off4 = (size_t)row * 4, tile_off8 = (size_t)row * 8, tile_offb = (size_t)(row >> 3);\n";
    o << "      const uint32_t lane_off4 = 0u, lane_off8 = 0u, lane_offb = 0u, lane_shb = (uint32_t)row & 4u;\n";
    o << "      (void)tile_off4; (void)tile_off8; (void)tile_offb; (void)lane_off4; (void)lane_off8; (void)lane_offb; (void)lane_shb; (void)raw;\n";
    loads(false);
    for (auto& f : raw_fields) o << "      raw." << f.second << " = " << f.second << ";\n";
    o << "  };\n";
    o << "  auto pred = [&](const Raw& raw) -> uint32_t {\n      (void)raw;\n";
    for (auto& f : raw_fields) o << "      const " << f.first << " " << f.second << " = raw." << f.second << "; (void)" << f.second << ";\n";
    o << "      return " << (s.code.empty() ? std::string("0xFu") : filter_expr()) << ";\n  };\n";
    // the argument block of the record that holds workgroup share `st` (shares only ever grow): pointers, literals, LUTs into LDS
    auto enter = [&]() {
      o << "    if (part < 0 || st >= tile_end) {\n      int np = part < 0 ? 0 : part;\n      while (np + 1 < n_parts && st >= parts[np].tile_end) np++;\n      part = np;\n";
      o << "      const FdbScanArgs& pa = parts[part];\n      n_rows = pa.n_rows; tile_begin = pa.tile_begin; tile_end = pa.tile_end; out_tile = pa.out_tile_base;\n";
      for (int i = 0; i < s.n_c4; i++)
        o << "      P_" << reg(false, false, i) << "_v = (const char*)pa.c4[" << i << "].values; P_" << reg(false, false, i) << "_b = pa.c4[" << i << "].validity;\n";
      for (int i = 0; i < s.n_c8; i++)
        o << "      P_" << reg(true, false, i) << "_v = (const char*)pa.c8[" << i << "].values; P_" << reg(true, false, i) << "_b = pa.c8[" << i << "].validity;\n";
      for (size_t l = 0; l < s.leaves.size(); l++)
        o << "      K_lit" << l << " = pa.leaves[" << l << "].lit; K_len" << l << " = pa.leaves[" << l << "].lut_len; K_lds" << l << " = pa.leaves[" << l << "].lut_lds; K_op" << l
          << " = pa.leaves[" << l << "].op; K_lut" << l << " = pa.leaves[" << l << "].lut;\n";
      for (size_t k = 0; k < fz.size(); k++) o << "      D_" << k << " = (char*)sa.sparts[part].dst[" << k << "];\n";
      o << "      if (pa.lut_class != lut_class) {\n        lut_class = pa.lut_class;\n        __syncthreads();\n";
      for (size_t l = 0; l < s.leaves.size(); l++)
        if (s.leaves[l].kind == FDB_LEAF_DICT_LUT && s.leaves[l].lut_in_lds)
          o << "        for (uint32_t i = tid; i < K_len" << l << "; i += " << BLK << ") smem[K_lds" << l << " + i] = as_global(K_lut" << l << ")[i];\n";
      o << "        __syncthreads();\n      }\n    }\n";
      o << "    unit = (st - tile_begin) * 8 + wave;  // this wave's half tile inside the record\n";
      o << "    rbase = unit * " << kSelectUnit << "LL;\n    active = rbase < n_rows;  // (the record's last share may be short of units)\n";
      o << "    if (active) {\n      const long long last_group = (n_rows - 1) & ~3LL;\n";
      o << "#pragma unroll\n      for (int u = 0; u < 4; u++) {\n        const long long row = rbase + u * 256 + (long long)lane * 4;\n";
      o << "        load(row < n_rows ? row : last_group, raw[u]);\n      }\n    }\n";
    };
    // The workgroup that arrives first does not filter: its first wave is the launch's SCANNER. It walks the shares in order, waits
    // for each one's count, and hands every share its place (the exclusive prefix inside its record) in a word of its own, 128 bytes
    // apart. A worker publishes its count and then polls that one word. (The first version let every workgroup sum the counts in
    // front of it itself, 256 status words per poll: 512 workgroups polling the same sixteen cache lines — one memory channel —
    // took 16 µs per round and set the kernel's pace; so did a status word per wave before that.) Nothing waits for a workgroup
    // that is not running: the scanner runs (it arrived first), and the counts it waits for belong to tickets that were drawn by
    // running workgroups, which publish before they wait.
    o << "  __shared__ int s_scanner;\n  if (tid == 0) s_scanner = (atomicAdd(sa.ctl + 2, 1ull) - sa.arrival_base) == 0ull ? 1 : 0;\n  __syncthreads();\n";
    o << "  unsigned long long* const place = sa.ctl + sa.place_off;  // place[16 st]: (epoch << 40) | exclusive prefix of share st\n";
    o << "  if (s_scanner != 0) {\n    if (wave != 0u) return;\n";
    o << "    long long pos = 0, p_end = parts[0].tile_end; int p = 0; unsigned long long running = 0; uint32_t spins = 0;\n";
    o << "    while (pos < total_tiles) {\n      unsigned long long v[4];\n";
    o << "#pragma unroll\n      for (int j = 0; j < 4; j++) {\n        const long long idx = pos + (long long)lane + 64 * j;\n        v[j] = 0ull;\n";
    o << "        if (idx < p_end) v[j] = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n      }\n";
    o << "      bool more = true; const long long pos0 = pos;\n";
    o << "#pragma unroll\n      for (int j = 0; j < 4; j++) {\n        if (!more) continue;\n";
    o << "        const long long idx = pos0 + (long long)lane + 64 * j;  // (== pos + lane: every earlier window was taken whole)\n";
    o << "        const bool in_range = idx < p_end;\n";
    o << "        const bool ready = in_range && (v[j] >> 40) == (unsigned long long)sa.epoch && ((v[j] >> 38) & 3ull) == 1ull;\n";
    o << "        const unsigned long long wait = __ballot(in_range && !ready), have = __ballot(in_range);\n";
    o << "        const int n = wait != 0ull ? __builtin_ctzll(wait) : __popcll(have);  // shares of this window whose counts are in, from the front\n";
    o << "        unsigned long long cnt = (int)lane < n ? (v[j] & VAL) : 0ull, incl = cnt;\n";
    o << "#pragma unroll\n        for (int sh = 1; sh < 64; sh <<= 1) { const unsigned long long t = (unsigned long long)__shfl_up((long long)incl, sh, 64); if ((int)lane >= sh) incl += t; }\n";
    o << "        if ((int)lane < n) __hip_atomic_store(place + idx * 16, ep | (running + incl - cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n";
    o << "        running += (unsigned long long)__shfl((long long)incl, 63, 64);\n        pos += n;\n        if (n < 64) more = false;\n      }\n";
    o << "      if (pos == p_end) {  // the record is complete: its row count, and the next record starts from nothing\n";
    o << "        if (lane == 0u) *sa.sparts[p].total = running;\n        running = 0; p++;\n        if (p < n_parts) p_end = parts[p].tile_end;\n      }\n";
    o << "      if (pos == pos0) {\n        if (++spins > (1u << 22)) { if (lane == 0u) sa.ctl[1] = 1ull; break; }  // (never seen; a stuck launch must not hang the device)\n";
    o << "        __builtin_amdgcn_s_sleep(" << s_sleep << ");\n      } else spins = 0;\n    }\n    return;\n  }\n";
    // thread 0 always holds the ticket after the next one
    o << "  long long next_ticket = 0;\n  if (tid == 0) { s_ticket[0] = (long long)(atomicAdd(sa.ctl, 1ull) - sa.ticket_base); next_ticket = s_ticket[0] < total_tiles ? (long long)(atomicAdd(sa.ctl, 1ull) - sa.ticket_base) : s_ticket[0]; }\n";
    o << "  __syncthreads();\n  long long st = s_ticket[0];\n  if (st >= total_tiles) return;\n";
    o << "  long long unit = 0, rbase = 0; bool active = false;\n  Raw raw[4];\n  {\n";
    enter();
    o << "  }\n";
    o << "  for (uint32_t it = 0;; it++) {\n";
    // 1. the current share: predicate, staging, bitmap
    o << "    uint32_t cnt = 0;  // selected rows of the unit so far (wave-uniform)\n";
    o << "    if (active) {\n      uint32_t sel[4];\n";
    o << "#pragma unroll\n      for (int u = 0; u < 4; u++) {\n";
    o << "        const long long left = n_rows - (rbase + u * 256 + (long long)lane * 4);\n";
    o << "        sel[u] = pred(raw[u]) & (left >= 4 ? 0xFu : left > 0 ? ((1u << (int)left) - 1u) : 0u);\n      }\n";
    o << "#pragma unroll\n      for (int u = 0; u < 4; u++) {\n        uint32_t w = sel[u];\n";
    o << "        const unsigned long long b0 = __ballot(w & 1u), b1 = __ballot(w & 2u), b2 = __ballot(w & 4u), b3 = __ballot(w & 8u);\n";
    if (!fz.empty() && ablate != 3) {
      // rows keep their order: lane L's four rows sit behind every selected row of the lanes below it
      o << "        const uint32_t p0 = cnt + lanes_below(b0) + lanes_below(b1) + lanes_below(b2) + lanes_below(b3);\n";
      o << "        const uint32_t p1 = p0 + (w & 1u), p2 = p1 + ((w >> 1) & 1u), p3 = p2 + ((w >> 2) & 1u);\n";
      for (size_t k = 0; k < fz.size(); k++) {
        const std::string r = "raw[u]." + reg(fz[k].wide, false, fz[k].slot);
        if (fz[k].wide) {
          o << "        { unsigned long long* sk = reinterpret_cast<unsigned long long*>(stg + " << fz[k].off << "u);\n";
          o << "          if (w & 1u) sk[p0] = " << r << "a.x; if (w & 2u) sk[p1] = " << r << "a.y; if (w & 4u) sk[p2] = " << r << "b.x; if (w & 8u) sk[p3] = " << r << "b.y; }\n";
        } else {
          o << "        { uint32_t* sk = reinterpret_cast<uint32_t*>(stg + " << fz[k].off << "u);\n";
          o << "          if (w & 1u) sk[p0] = " << r << ".x; if (w & 2u) sk[p1] = " << r << ".y; if (w & 4u) sk[p2] = " << r << ".z; if (w & 8u) sk[p3] = " << r << ".w; }\n";
        }
      }
    }
    o << "        cnt += (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));\n";
    // a step's 256 mask bits = 8 words: lane L's nibble sits at bits 4 (L % 8) of word L / 8
    o << "        w |= (uint32_t)__shfl_down((int)w, 1, 64) << 4;\n        w |= (uint32_t)__shfl_down((int)w, 2, 64) << 8;\n        w |= (uint32_t)__shfl_down((int)w, 4, 64) << 16;\n";
    o << "        sel[u] = w;\n      }\n";
    o << "      if ((lane & 7u) == 0u) {\n        uint32_t* mw = masks + (size_t)(out_tile + (unit >> 1)) * 64 + (size_t)(unit & 1) * 32 + (lane >> 3);\n";
    o << "        mw[0] = sel[0]; mw[8] = sel[1]; mw[16] = sel[2]; mw[24] = sel[3];\n      }\n";
    o << "    }\n";
    // 2. the waves' counts meet in LDS; the next ticket with them
    o << "    if (lane == 0u) s_cnt[it & 1u][wave] = cnt;\n    if (tid == 0) s_ticket[(it + 1u) & 1u] = next_ticket;\n    __syncthreads();\n";
    o << "    const long long c_st = st, c_out_tile = out_tile, c_unit = unit;\n    const bool c_active = active;\n";
    for (size_t k = 0; k < fz.size(); k++) o << "    char* const c_D_" << k << " = D_" << k << "; (void)c_D_" << k << ";\n";
    // the first wave publishes the share's count — before anything else is loaded or waited for
    o << "    uint32_t share = 0;\n";
    o << "    if (wave == 0u) {\n      share = lane < 8u ? s_cnt[it & 1u][lane] : 0u;\n";
    o << "#pragma unroll\n      for (int sh = 4; sh > 0; sh >>= 1) share += (uint32_t)__shfl_xor((int)share, sh, 64);\n";
    o << "      share = (uint32_t)__builtin_amdgcn_readfirstlane((int)share);\n";
    o << "      if (lane == 0u) __hip_atomic_store(status + c_st, ep | (1ull << 38) | (unsigned long long)share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n    }\n";
    // 3. the next share: its loads go out now and land while the first wave looks back
    o << "    const long long nst = s_ticket[(it + 1u) & 1u];\n";
    o << "    if (tid == 0 && nst < total_tiles) next_ticket = (long long)(atomicAdd(sa.ctl, 1ull) - sa.ticket_base);\n";
    o << "    if (nst < total_tiles) {\n      st = nst;\n";
    enter();
    o << "    }\n";
    // 4. the share's place, from the scanner
    o << "    if (wave == 0u) {\n      unsigned long long excl = 0;\n";
    if (ablate != 1) {
      o << "      uint32_t spins = 0;\n      for (;;) {\n        const unsigned long long v = __hip_atomic_load(place + c_st * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n";
      o << "        if ((v >> 40) == (unsigned long long)sa.epoch) { excl = v & VAL; break; }\n";
      o << "        if (++spins > (1u << 22)) { if (lane == 0u) sa.ctl[1] = 1ull; break; }  // (never seen)\n";
      o << "        __builtin_amdgcn_s_sleep(" << w_sleep << ");\n      }\n";
    }
    o << "      if (lane == 0u) s_base = excl;\n    }\n    __syncthreads();\n";
    // 5. every wave writes its staged values to their place
    o << "    if (c_active) {\n      unsigned long long excl = s_base;\n";
    o << "      for (uint32_t w2 = 0; w2 < wave; w2++) excl += s_cnt[it & 1u][w2];\n";
    o << "      if (lane == 0u && (c_unit & 1) == 0) offsets[c_out_tile + (c_unit >> 1)] = (uint32_t)excl;\n";
    if (!fz.empty() && ablate < 2) {
      for (size_t k = 0; k < fz.size(); k++) {
        if (fz[k].wide) {
          o << "      { unsigned long long* d = reinterpret_cast<unsigned long long*>(c_D_" << k << ") + excl;\n";
          o << "        const u32x4* sv = reinterpret_cast<const u32x4*>(stg + " << fz[k].off << "u);\n";
          o << "        for (uint32_t i = lane * 2u; i < cnt; i += 128u) {\n";
          o << "          if (i + 1u < cnt) __builtin_nontemporal_store(sv[i >> 1], reinterpret_cast<u32x4*>(d + i));\n";
          o << "          else d[i] = reinterpret_cast<const unsigned long long*>(sv)[i];\n        }\n      }\n";
        } else {
          o << "      { uint32_t* d = reinterpret_cast<uint32_t*>(c_D_" << k << ") + excl;\n";
          o << "        const u32x4* sv = reinterpret_cast<const u32x4*>(stg + " << fz[k].off << "u);\n";
          o << "        for (uint32_t i = lane * 4u; i < cnt; i += 256u) {\n";
          o << "          if (i + 3u < cnt) __builtin_nontemporal_store(sv[i >> 2], reinterpret_cast<u32x4*>(d + i));\n";
          o << "          else { const u32x4 t = sv[i >> 2]; d[i] = t.x; if (i + 1u < cnt) d[i + 1] = t.y; if (i + 2u < cnt) d[i + 2] = t.z; }\n        }\n      }\n";
        }
      }
    }
    o << "    }\n    __builtin_amdgcn_wave_barrier();  // (the next unit stages into the same region)\n";
    o << "    if (nst >= total_tiles) break;\n";
    o << "  }\n}\n";
    return o.str();
  }
};


// ---- high-cardinality path: fdb_hash_kernel -----------------------------------------------------------------------------
// Same job as scan_hash_kernel (fdb_kernels.hip) with the plan's shape baked in: 4 consecutive rows per lane (every index
// column is one 16-byte load per lane), key columns folded into the fingerprint 8 at a time with all 8 loads in flight,
// no per-row staging of the key tuple (the rare lane that creates a group re-reads its row's columns instead).
const char* kHashPreamble = "";

struct HashGen {
  std::ostringstream o;
  const JitHashShape& s;
  explicit HashGen(const JitHashShape& sh) : s(sh) {}
  static const char* comp4(int k) { static const char* c[4] = {".x", ".y", ".z", ".w"}; return c[k]; }
  static std::string comp8(const std::string& r, int k) { return r + (k < 2 ? "a" : "b") + (k % 2 == 0 ? ".x" : ".y"); }

  std::string source() {
    const int BLK = 256, TILE = BLK * 4, GROUP = 8;
    // runs: 1 = narrow records (a key id is a byte, packed while the fingerprint is computed), 2 = wide records (the table's own key tuple,
    // written by the lanes that end a run from re-loaded columns — fdb_kernels.h FdbRunsOut)
    // 3 = medium records: like 1 with TWO bytes per key id (≤ 65 534 distinct values per column; 16 more registers per row)
    const bool narrow = s.runs == 1, wide = s.runs == 2, medium = s.runs == 3;
    if (temporal_loads(s.runs)) o << "#define FDB_LD_TEMPORAL 1\n";
    o << "#define FDB_DEVICE_HELPERS 1\n#include \"fdb_kernels.h\"\n" << kPreamble << kHashPreamble;
    // The runs kernel wants ≈149 VGPRs = 3 waves per SIMD, which is also what its LDS stage (4 × 12 KiB per workgroup) lets a CU hold.
    // ($FDB_RUNS_WAVES_PER_EU: tuning aid — caps the registers so that that many waves fit a SIMD; 0 / unset = no cap)
    const int waves_cap = s.runs && std::getenv("FDB_RUNS_WAVES_PER_EU") ? std::atoi(std::getenv("FDB_RUNS_WAVES_PER_EU")) : 0;
    o << "extern \"C\" __global__ __launch_bounds__(" << BLK << ") ";
    if (waves_cap > 0) o << "__attribute__((amdgpu_waves_per_eu(" << waves_cap << ", " << waves_cap << "))) ";
    o << "void fdb_hash_kernel(const FdbHashArgs h) {\n";
    o << "  extern __shared__ __align__(16) unsigned char smem[];\n  __shared__ unsigned int s_new;\n";
    if (s.runs) o << "  __shared__ unsigned int s_runs[16];  // per wave: first run of its open chunk, runs used in it, first of those still waiting in LDS\n  if (threadIdx.x < 16) s_runs[threadIdx.x] = (threadIdx.x & 3u) == 0u ? 0u : " << FDB_RUN_CHUNK << "u;\n";
    o << "  const FdbScanArgs& a = h.base;\n  // descriptors are read through the constant address space: scalar loads, no vector registers\n  const __attribute__((address_space(4))) FdbHashCol* hc = (const __attribute__((address_space(4))) FdbHashCol*)h.hcols;\n  const uint32_t tid = threadIdx.x;\n  if (tid == 0) s_new = 0;\n";
    for (size_t l = 0; l < s.leaves.size(); l++) {
      o << "  const long long K_lit" << l << " = a.leaves[" << l << "].lit; const uint32_t K_len" << l << " = a.leaves[" << l << "].lut_len, K_lds" << l << " = a.leaves[" << l
        << "].lut_lds; const int K_op" << l << " = a.leaves[" << l << "].op; const uint8_t* K_lut" << l << " = a.leaves[" << l << "].lut;\n";
      o << "  (void)K_lit" << l << "; (void)K_len" << l << "; (void)K_lds" << l << "; (void)K_op" << l << "; (void)K_lut" << l << ";\n";
      if (s.leaves[l].kind == FDB_LEAF_DICT_LUT && s.leaves[l].lut_in_lds)
        o << "  for (uint32_t i = tid; i < K_len" << l << "; i += " << BLK << ") smem[K_lds" << l << " + i] = as_global(K_lut" << l << ")[i];\n";
    }
    for (size_t c = 0; c < s.cols.size(); c++)
      if (s.cols[c].kind == 0 && s.cols[c].lut_in_lds && !s.cols[c].lut_identity)
        o << "  { uint32_t* dst = reinterpret_cast<uint32_t*>(smem + hc[" << c << "].lut_lds); const uint32_t n = hc[" << c << "].lut_len; const uint32_t* src = hc[" << c
          << "].lut; for (uint32_t i = tid; i < n; i += " << BLK << ") dst[i] = as_global(src)[i]; }\n";
    o << "  __syncthreads();\n";
    for (size_t i = 0; i < s.exprs.size(); i++) if (s.exprs[i].kind == 1) o << "  const long long K_elit" << i << " = a.expr[" << i << "].lit; (void)K_elit" << i << ";\n";
    o << "  const int ew = h.entry_words;\n  const long long n_tiles = (h.row_end - h.row_begin + " << TILE - 1 << ") / " << TILE << ";\n";
    o << "  const uint32_t lane_shb = (tid & 1u) * 4u;\n";
    o << "  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {\n";
    o << "    const long long r0 = h.row_begin + tile * " << TILE << "LL;\n    const long long left = h.row_end - r0 - (long long)tid * 4;\n    if (left <= 0) continue;\n";
    o << "    const size_t o4 = (size_t)r0 * 4, o8 = (size_t)r0 * 8, ob = (size_t)(r0 >> 3);\n";
    // The lane offsets are made opaque per tile: otherwise the compiler pre-computes one 64-bit (column base + lane offset)
    // address pair per column and validity bitmap outside the loop (≈130 VGPRs for 32 columns) instead of using the
    // scalar-base + 32-bit-vector-offset form of global_load.
    o << "    uint32_t lane_off4 = tid * 16u, lane_off8 = tid * 32u, lane_offb = tid >> 1;\n    asm volatile(\"\" : \"+v\"(lane_off4), \"+v\"(lane_off8), \"+v\"(lane_offb));\n";
    o << "    uint32_t sel = left >= 4 ? 0xFu : ((1u << (int)left) - 1u);\n";
    // filter: one register set per leaf
    for (size_t l = 0; l < s.leaves.size(); l++) {
      const JitLeaf& L = s.leaves[l];
      if (L.kind == FDB_LEAF_CONST) continue;
      const std::string r = "f" + std::to_string(l);
      if (L.kind != FDB_LEAF_VALIDITY) {
        if (L.wide) {
          o << "    const u64x2 " << r << "a = ld8(reinterpret_cast<const char*>(a.leaves[" << l << "].values) + o8, lane_off8);\n";
          o << "    const u64x2 " << r << "b = ld8(reinterpret_cast<const char*>(a.leaves[" << l << "].values) + o8, lane_off8 + 16u);\n";
        } else {
          o << "    const u32x4 " << r << " = ld4(reinterpret_cast<const char*>(a.leaves[" << l << "].values) + o4, lane_off4);\n";
        }
      }
      if (s.leaf_validity[l]) o << "    const uint32_t " << r << "_m = ldv(a.leaves[" << l << "].validity + ob, lane_offb, lane_shb);\n";
      else o << "    const uint32_t " << r << "_m = 0xFu;\n";
    }
    if (!s.code.empty()) {
      o << "    sel &= " << Gen::filter_expr_of(s.code, [&](int l) { return Gen::leaf_expr_of(s.leaves[(size_t)l], l, "f" + std::to_string(l)); }) << ";\n";
      o << "    if (sel == 0u) continue;\n";
    }
    {  // match bits of the leaves that boolean projections read (expression nodes of kind 7): leaves outside the filter program
      std::vector<int> seen;
      for (const JitExprNode& e : s.exprs)
        if (e.kind == 7 && std::find(seen.begin(), seen.end(), e.slot) == seen.end()) {
          seen.push_back(e.slot);
          o << "    const uint32_t PM" << e.slot << " = " << Gen::leaf_expr_of(s.leaves[(size_t)e.slot], e.slot, "f" + std::to_string(e.slot)) << "; (void)PM" << e.slot << ";\n";
        }
    }
    // columns read by computed inputs / keys (base.l8), then the aggregated columns: requested now, consumed after the probe
    for (int x = 0; x < s.n_expr_cols; x++) {
      o << "    const u64x2 x" << x << "a = ld8(reinterpret_cast<const char*>(a.l8[" << x << "].values) + o8, lane_off8);\n";
      o << "    const u64x2 x" << x << "b = ld8(reinterpret_cast<const char*>(a.l8[" << x << "].values) + o8, lane_off8 + 16u);\n";
      o << "    const uint32_t x" << x << "_m = a.l8[" << x << "].validity != nullptr ? ldv(a.l8[" << x << "].validity + ob, lane_offb, lane_shb) : 0xFu; (void)x" << x << "_m;\n";
    }
    for (size_t j = 0; j < s.aggs.size(); j++) {
      if (s.aggs[j].func == FDB_AGG_COUNT || s.aggs[j].expr != 0) continue;
      const std::string r = "g" + std::to_string(j);
      o << "    const u64x2 " << r << "a = ld8(reinterpret_cast<const char*>(a.aggs[" << j << "].values) + o8, lane_off8);\n";
      o << "    const u64x2 " << r << "b = ld8(reinterpret_cast<const char*>(a.aggs[" << j << "].values) + o8, lane_off8 + 16u);\n";
      if (s.agg_validity[j]) o << "    const uint32_t " << r << "_m = ldv(a.aggs[" << j << "].validity + ob, lane_offb, lane_shb);\n";
      else o << "    const uint32_t " << r << "_m = 0xFu;\n";
    }
    o << "    unsigned long long h1_0 = 0, h1_1 = 0, h1_2 = 0, h1_3 = 0, h2_0 = 0, h2_1 = 0, h2_2 = 0, h2_3 = 0, vm_0 = 0, vm_1 = 0, vm_2 = 0, vm_3 = 0;\n";
    // runs mode: the key ids of a row, one byte per group column, packed as they are computed (8 registers per row)
    if (narrow) for (int k = 0; k < 4; k++) o << "    u32x4 ta_" << k << " = {0u, 0u, 0u, 0u}, tb_" << k << " = {0u, 0u, 0u, 0u};\n";
    if (medium) for (int k = 0; k < 4; k++) o << "    u32x4 ta_" << k << " = {0u, 0u, 0u, 0u}, tb_" << k << " = {0u, 0u, 0u, 0u}, tc_" << k << " = {0u, 0u, 0u, 0u}, td_" << k << " = {0u, 0u, 0u, 0u};\n";
    for (size_t c0 = 0; c0 < s.cols.size(); c0 += GROUP) {
      const size_t c1 = std::min(s.cols.size(), c0 + GROUP);
      o << "    {\n";
      for (size_t c = c0; c < c1; c++) {
        const JitHashCol& C = s.cols[c];
        const std::string r = "k" + std::to_string(c);
        if (C.kind == 2) continue;
        if (C.kind == 0) o << "      const u32x4 " << r << " = ld4(reinterpret_cast<const char*>(hc[" << c << "].values) + o4, lane_off4);\n";
        else {
          o << "      const u64x2 " << r << "a = ld8(reinterpret_cast<const char*>(hc[" << c << "].values) + o8, lane_off8);\n";
          o << "      const u64x2 " << r << "b = ld8(reinterpret_cast<const char*>(hc[" << c << "].values) + o8, lane_off8 + 16u);\n";
        }
        if (C.has_validity) o << "      const uint32_t " << r << "_m = ldv(hc[" << c << "].validity + ob, lane_offb, lane_shb);\n";
        else o << "      const uint32_t " << r << "_m = 0xFu;\n";
      }
      for (size_t c = c0; c < c1; c++) {
        const JitHashCol& C = s.cols[c];
        const std::string r = "k" + std::to_string(c);
        // (the multipliers are made opaque per tile for the same reason as the lane offsets: their VGPR copies — 4 per column —
        // would otherwise be hoisted out of the tile loop and stay live across it)
        o << "      {\n        unsigned long long K1 = hc[" << c << "].k1, K2 = hc[" << c << "].k2; asm volatile(\"\" : \"+s\"(K1), \"+s\"(K2));\n        const unsigned long long bit = 1ull << hc[" << c
          << "].gi;\n";
        if (C.kind == 0) {
          // (lut_identity: the record's dictionary IS the plan's value list of this column, in order — key id = index + 1, no table to read)
          if (C.lut_identity) {}
          else if (C.lut_in_lds) o << "        const uint32_t* L = reinterpret_cast<const uint32_t*>(smem + hc[" << c << "].lut_lds);\n";
          else o << "        const uint32_t* L = hc[" << c << "].lut;\n";
          for (int k = 0; k < 4; k++) {
            o << "        { const uint32_t id = ((" << r << "_m >> " << k << ") & 1u) ? " << (C.lut_identity ? "(" + r + comp4(k) + " + 1u)" : std::string(C.lut_in_lds ? "L" : "as_global(L)") + "[" + r + comp4(k) + "]") << " : 0u; fp_add32(h1_" << k << ", h2_"
              << k << ", K1, K2, id);" << (narrow || medium ? "" : " if (id != 0u) vm_" + std::to_string(k) + " |= bit;");
            if (narrow) o << " t" << (c < 16 ? "a" : "b") << "_" << k << comp4((int)((c % 16) / 4)) << " |= id << " << 8 * (c % 4) << ";";
            if (medium) o << " t" << "abcd"[c / 8] << "_" << k << comp4((int)((c % 8) / 2)) << " |= id << " << 16 * (c % 2) << ";";
            o << " }\n";
          }
        } else if (C.kind == 1) {
          // int64 keys: NULL and the value 0 hash alike in the reference (dynparquet/hashed.go:254-272, zero hashes are skipped by
          // aggregate.go:398-409), so neither contributes to the fingerprint; the valid bit records which of the two this row is
          for (int k = 0; k < 4; k++)
            o << "        if ((" << r << "_m >> " << k << ") & 1u) { if (" << comp8(r, k) << " != 0ull) fp_add(h1_" << k << ", h2_" << k << ", K1, K2, " << comp8(r, k) << "); vm_" << k << " |= bit; }\n";
        } else {
          for (int k = 0; k < 4; k++) {
            auto col = [&](int ni) -> std::string { if (s.exprs[(size_t)ni].kind == 7) return "((PM" + std::to_string(s.exprs[(size_t)ni].slot) + " >> " + std::to_string(k) + ") & 1u)"; return comp8("x" + std::to_string(s.exprs[(size_t)ni].slot), k); };
            auto colvalid = [&](int ni) { return "((x" + std::to_string(s.exprs[(size_t)ni].slot) + "_m >> " + std::to_string(k) + ") & 1u)"; };
            o << "        if (" << expr_valid(s.exprs, C.expr_root, col, colvalid) << ") { const unsigned long long y = " << expr_bits(s.exprs, C.expr_root, expr_value(s.exprs, C.expr_root, col, colvalid))
              << "; if (y != 0ull) fp_add(h1_" << k << ", h2_" << k << ", K1, K2, y); vm_" << k << " |= bit; }\n";
          }
        }
        o << "      }\n";
      }
      // keep the next group's loads below this point: hoisting all 32 columns' loads to the top of the tile costs ≈390 VGPRs
      // and force the fingerprint updates to happen HERE: LLVM otherwise sinks all 32 columns' multiply-adds into the per-row
      // `if (selected)` blocks below and keeps 4 × 32 key ids live until then
      if (narrow || medium) o << "      asm volatile(\"\" : \"+v\"(h1_0), \"+v\"(h1_1), \"+v\"(h1_2), \"+v\"(h1_3), \"+v\"(h2_0), \"+v\"(h2_1), \"+v\"(h2_2), \"+v\"(h2_3) :: \"memory\");\n";
      else o << "      asm volatile(\"\" : \"+v\"(h1_0), \"+v\"(h1_1), \"+v\"(h1_2), \"+v\"(h1_3), \"+v\"(h2_0), \"+v\"(h2_1), \"+v\"(h2_2), \"+v\"(h2_3), \"+v\"(vm_0), \"+v\"(vm_1), \"+v\"(vm_2), \"+v\"(vm_3) :: \"memory\");\n";
      // (same for the packed key ids of runs mode: pinned here, or all 4 × 32 ids stay live until the rows' tuples are stored)
      if (narrow) o << "      asm volatile(\"\" : \"+v\"(ta_0), \"+v\"(ta_1), \"+v\"(ta_2), \"+v\"(ta_3), \"+v\"(tb_0), \"+v\"(tb_1), \"+v\"(tb_2), \"+v\"(tb_3));\n";
      if (medium) {  // (only the register quads this column group wrote: the others are still zero and need not be pinned)
        const char q = "abcd"[c0 / 8];
        o << "      asm volatile(\"\" : \"+v\"(t" << q << "_0), \"+v\"(t" << q << "_1), \"+v\"(t" << q << "_2), \"+v\"(t" << q << "_3));\n";
      }
      o << "    }\n";
    }
    for (int k = 0; k < 4; k++) o << "    fp_final(h1_" << k << ", h2_" << k << ");\n";
    if (s.ablate & 1) o << "    if ((h1_0 ^ h2_0 ^ h1_1 ^ h2_1 ^ h1_2 ^ h2_2 ^ h1_3 ^ h2_3) == 0x1234567ull) h.table[0] = vm_0 ^ vm_1 ^ vm_2 ^ vm_3;\n    continue;\n";  // tuning aid
    // Run combining: a lane owns 4 CONSECUTIVE rows, and scans of tables sorted by their label columns (FrostDB's sorting columns)
    // bring rows of one group next to each other. Rows of the lane with the same fingerprint as the row before them are folded
    // into it — count and every aggregate — and only the LAST row of such a run goes to the table: one probe + one set of atomics
    // per run instead of per row. Costs a few compares on unsorted input.
    const bool combine = !(s.ablate & 2) || s.runs != 0;
    const int runs_ablate = s.runs && std::getenv("FDB_RUNS_ABLATE") ? std::atoi(std::getenv("FDB_RUNS_ABLATE")) : 0;  // (tuning aid: 1 no stores, 2 no folding across lanes; results are wrong)
    if (combine) {
      for (int k = 0; k < 4; k++) o << "    unsigned long long cnt_" << k << " = 1ull;\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT) continue;
        for (int k = 0; k < 4; k++) {
          const std::string r = "g" + std::to_string(j);
          std::string raw = "(((" + r + "_m >> " + std::to_string(k) + ") & 1u) ? " + comp8(r, k) + " : a.aggs[" + std::to_string(j) + "].null_value)";
          if (A.expr != 0) {
            auto col = [&](int ni) -> std::string { if (s.exprs[(size_t)ni].kind == 7) return "((PM" + std::to_string(s.exprs[(size_t)ni].slot) + " >> " + std::to_string(k) + ") & 1u)"; return comp8("x" + std::to_string(s.exprs[(size_t)ni].slot), k); };
            auto colvalid = [&](int ni) { return "((x" + std::to_string(s.exprs[(size_t)ni].slot) + "_m >> " + std::to_string(k) + ") & 1u)"; };
            raw = "(" + expr_valid(s.exprs, A.expr - 1, col, colvalid) + " ? " + expr_bits(s.exprs, A.expr - 1, expr_value(s.exprs, A.expr - 1, col, colvalid)) + " : 0ull)";
          }
          const std::string v = "v" + std::to_string(j) + "_" + std::to_string(k);
          if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "    double " << v << " = __longlong_as_double((long long)" << raw << ");\n";
          else if (A.func == FDB_AGG_SUM) o << "    unsigned long long " << v << " = " << raw << ";\n";
          else o << "    long long " << v << " = " << (A.type == FDB_T_F64 ? ("f64_minmax_key(__longlong_as_double((long long)" + raw + "), " + (A.func == FDB_AGG_MIN ? "true" : "false") + ")") : ("(long long)" + raw)) << ";\n";
        }
      }
      for (int k = 1; k < 4; k++) {
        o << "    if (((sel >> " << (k - 1) << ") & 3u) == 3u && h1_" << (k - 1) << " == h1_" << k << " && h2_" << (k - 1) << " == h2_" << k << ") {\n";
        o << "      cnt_" << k << " += cnt_" << (k - 1) << ";\n";
        for (size_t j = 0; j < s.aggs.size(); j++) {
          const JitAgg& A = s.aggs[j];
          if (A.func == FDB_AGG_COUNT) continue;
          const std::string a = "v" + std::to_string(j) + "_" + std::to_string(k - 1), b = "v" + std::to_string(j) + "_" + std::to_string(k);
          if (A.func == FDB_AGG_SUM) o << "      " << b << " += " << a << ";\n";
          else if (A.func == FDB_AGG_MIN) o << "      " << b << " = " << a << " < " << b << " ? " << a << " : " << b << ";\n";
          else o << "      " << b << " = " << a << " > " << b << " ? " << a << " : " << b << ";\n";
        }
        o << "      sel &= ~" << (1u << (k - 1)) << "u;\n    }\n";
      }
      // The same across the lanes of the wave (its 256 rows are consecutive): a run that continues in the next lane hands its
      // folded partial on instead of touching the table, and the lane where the run ends applies the total — one probe + one set of
      // atomics per run and wave. Kogge-Stone scan over lanes of (fingerprint of the lane's trailing run, its partial, "the run
      // covers every row of the span"); a lane without selected rows ends every run (less folding, never a wrong one). Equal keys
      // that are not adjacent are simply two runs. Costs ≈50 cross-lane moves per 256 rows when nothing folds.
      auto pick = [&](const std::string& base, const char* idx) {
        return "(" + std::string(idx) + " == 0 ? " + base + "_0 : " + idx + " == 1 ? " + base + "_1 : " + idx + " == 2 ? " + base + "_2 : " + base + "_3)";
      };
      // (lanes that left the tile early — rows past the end, nothing selected by the filter — do not take part: what a shuffle
      // reads from them is not data, so their bits in `act` gate every value that comes from another lane)
      if (runs_ablate & 2) o << "    if (false)\n";
      o << "    {\n      const int wl = (int)(tid & 63u);\n      const bool has = sel != 0u;\n      const unsigned long long act = __ballot(1);\n";
      o << "      const int kf = has ? __builtin_ctz(sel) : 0, kl = has ? 31 - __builtin_clz(sel) : 0;\n";
      o << "      const unsigned long long fa = " << pick("h1", "kf") << ", fb = " << pick("h2", "kf") << ";\n";
      o << "      const unsigned long long ta = " << pick("h1", "kl") << ", tb = " << pick("h2", "kl") << ";\n";
      o << "      unsigned long long t_cnt = " << pick("cnt", "kl") << ";\n";
      std::vector<size_t> folded;  // aggregates that carry a value
      for (size_t j = 0; j < s.aggs.size(); j++) if (s.aggs[j].func != FDB_AGG_COUNT) folded.push_back(j);
      auto vtype = [&](size_t j) { const JitAgg& A = s.aggs[j]; return A.func == FDB_AGG_SUM ? (A.type == FDB_T_F64 ? "double" : "unsigned long long") : "long long"; };
      for (size_t j : folded) o << "      " << vtype(j) << " t_v" << j << " = " << pick("v" + std::to_string(j), "kl") << ";\n";
      o << "      int t_flags = has ? (2 | ((sel & (sel - 1u)) == 0u ? 1 : 0)) : 0;  // bit 1: the span ends in a run; bit 0: that run covers the whole span\n";
      for (int off = 1; off < 64; off <<= 1) {
        o << "      {\n        const unsigned long long pa = __shfl_up(ta, " << off << ", 64), pb = __shfl_up(tb, " << off << ", 64), pc = __shfl_up(t_cnt, " << off << ", 64);\n";
        for (size_t j : folded) o << "        const " << vtype(j) << " pv" << j << " = __shfl_up(t_v" << j << ", " << off << ", 64);\n";
        o << "        const int pf_raw = __shfl_up(t_flags, " << off << ", 64);\n        const int pf = (wl >= " << off << " && ((act >> (wl - " << off << ")) & 1ull)) ? pf_raw : 0;\n";
        o << "        if (wl >= " << off << " && (t_flags & 3) == 3) {\n";
        o << "          if ((pf & 2) && pa == ta && pb == tb) {\n            t_cnt += pc;\n";
        for (size_t j : folded) {
          const JitAgg& A = s.aggs[j];
          const std::string t = "t_v" + std::to_string(j), pv = "pv" + std::to_string(j);
          if (A.func == FDB_AGG_SUM) o << "            " << t << " += " << pv << ";\n";
          else if (A.func == FDB_AGG_MIN) o << "            " << t << " = " << pv << " < " << t << " ? " << pv << " : " << t << ";\n";
          else o << "            " << t << " = " << pv << " > " << t << " ? " << pv << " : " << t << ";\n";
        }
        o << "            t_flags = 2 | (pf & 1);\n          } else {\n            t_flags = 2;\n          }\n        }\n      }\n";
      }
      // what the lanes before this one hand over (the inclusive result of lane - 1), and whether the next lane takes over
      o << "      const unsigned long long xa = __shfl_up(ta, 1, 64), xb = __shfl_up(tb, 1, 64), xc = __shfl_up(t_cnt, 1, 64);\n";
      for (size_t j : folded) o << "      const " << vtype(j) << " xv" << j << " = __shfl_up(t_v" << j << ", 1, 64);\n";
      o << "      const int xf_raw = __shfl_up(t_flags, 1, 64);\n      const int xf = (wl > 0 && ((act >> (wl - 1)) & 1ull)) ? xf_raw : 0;\n";
      o << "      const unsigned long long na = __shfl_down(fa, 1, 64), nb = __shfl_down(fb, 1, 64);\n      const int nh_raw = __shfl_down((int)has, 1, 64);\n      const int nh = (wl < 63 && ((act >> (wl + 1)) & 1ull)) ? nh_raw : 0;\n";
      o << "      if (has && wl > 0 && (xf & 2) && xa == fa && xb == fb) {\n";
      for (int k = 0; k < 4; k++) {
        o << "        if (kf == " << k << ") {\n          cnt_" << k << " += xc;\n";
        for (size_t j : folded) {
          const JitAgg& A = s.aggs[j];
          const std::string v = "v" + std::to_string(j) + "_" + std::to_string(k), xv = "xv" + std::to_string(j);
          if (A.func == FDB_AGG_SUM) o << "          " << v << " += " << xv << ";\n";
          else if (A.func == FDB_AGG_MIN) o << "          " << v << " = " << xv << " < " << v << " ? " << xv << " : " << v << ";\n";
          else o << "          " << v << " = " << xv << " > " << v << " ? " << xv << " : " << v << ";\n";
        }
        o << "        }\n";
      }
      o << "      }\n";
      o << "      if (has && wl < 63 && nh && na == ta && nb == tb) sel &= ~(1u << kl);\n    }\n";
    }
    std::vector<int> cword(s.cols.size());
    { int w = 4; for (size_t c = 0; c < s.cols.size(); c++) { cword[c] = w; w += s.cols[c].kind == 0 ? 1 : 2; } }
    // (`for_runs`: the wide run records of an ordered plan take the same road — the "inserting" rows are the ones that end a run, the
    // destinations d_0 … d_3 are the runs' places in the wave's LDS stage or in the chunk, and count + aggregate follow the tuple)
    auto emit_tuple_stores = [&](bool for_runs) {
    if (!for_runs) {
      o << "    if (ins_mask != 0u) {\n";
      o << "      atomicAdd(&s_new, (unsigned int)__builtin_popcount(ins_mask));\n";
    } else {
      o << "    {\n";
    }
    // (fresh opaque lane offsets: with the tile's own the compiler recognises these loads as the fingerprint phase's and keeps
    // all 32 columns' values — 128 VGPRs — alive from there to here instead of re-loading)
    o << "      uint32_t ioff4 = lane_off4, ioff8 = lane_off8, ioffb = lane_offb;\n      asm volatile(\"\" : \"+v\"(ioff4), \"+v\"(ioff8), \"+v\"(ioffb));\n";
    if (!for_runs) for (int k = 0; k < 4; k++) o << "      uint32_t* d_" << k << " = h.keys + slot_" << k << " * (uint64_t)h.key_words;\n";
    o << "      const bool canon = h.canonical != 0;\n";
    o << "      if (!canon) {  // columns this record does not carry are NULL\n";
    for (int k = 0; k < 4; k++) o << "        if (ins_mask & " << (1 << k) << "u) for (int w = 2; w < h.key_words; w++) d_" << k << "[w] = 0u;\n";
    o << "      }\n";
    for (int k = 0; k < 4; k++)
      o << "      if (ins_mask & " << (1 << k) << "u) { if (canon) *reinterpret_cast<u32x4*>(d_" << k << ") = u32x4{(uint32_t)vm_" << k << ", (uint32_t)(vm_" << k << " >> 32), 0u, 0u}; else { d_" << k
        << "[0] = (uint32_t)vm_" << k << "; d_" << k << "[1] = (uint32_t)(vm_" << k << " >> 32); } }\n";
    for (size_t c0 = 0; c0 < s.cols.size(); c0 += GROUP) {
      const size_t c1 = std::min(s.cols.size(), c0 + GROUP);
      o << "      {\n";
      for (size_t c = c0; c < c1; c++) {
        const JitHashCol& C = s.cols[c];
        const std::string r = "w" + std::to_string(c);
        if (C.kind == 2) continue;
        if (C.kind == 0) o << "        const u32x4 " << r << " = ld4(reinterpret_cast<const char*>(hc[" << c << "].values) + o4, ioff4);\n";
        else {
          o << "        const u64x2 " << r << "a = ld8(reinterpret_cast<const char*>(hc[" << c << "].values) + o8, ioff8);\n";
          o << "        const u64x2 " << r << "b = ld8(reinterpret_cast<const char*>(hc[" << c << "].values) + o8, ioff8 + 16u);\n";
        }
        if (C.has_validity) o << "        const uint32_t " << r << "_m = ldv(hc[" << c << "].validity + ob, ioffb, lane_shb);\n";
        else o << "        const uint32_t " << r << "_m = 0xFu;\n";
      }
      // key ids of the dictionary columns (only for rows that inserted: other rows of the tile's tail may hold anything)
      for (size_t c = c0; c < c1; c++) {
        const JitHashCol& C = s.cols[c];
        if (C.kind != 0) continue;
        const std::string r = "w" + std::to_string(c);
        if (C.lut_identity) {}
        else if (C.lut_in_lds) o << "        const uint32_t* L" << c << " = reinterpret_cast<const uint32_t*>(smem + hc[" << c << "].lut_lds);\n";
        else o << "        const uint32_t* L" << c << " = hc[" << c << "].lut;\n";
        for (int k = 0; k < 4; k++)
          o << "        const uint32_t i" << c << "_" << k << " = ((ins_mask & " << (1 << k) << "u) && ((" << r << "_m >> " << k << ") & 1u)) ? "
            << (C.lut_identity ? "(" + r + comp4(k) + " + 1u)" : std::string(C.lut_in_lds ? "L" : "as_global(L") + std::to_string(c) + (C.lut_in_lds ? "" : ")") + "[" + r + comp4(k) + "]") << " : 0u;\n";
      }
      // stores: aligned quads of dictionary columns as one 16-byte store when the layout is canonical
      std::vector<bool> in_quad(s.cols.size(), false);
      o << "        if (canon) {\n";
      for (size_t c = c0; c + 4 <= c1; ) {
        const bool quad = cword[c] % 4 == 0 && s.cols[c].kind == 0 && s.cols[c + 1].kind == 0 && s.cols[c + 2].kind == 0 && s.cols[c + 3].kind == 0;
        if (!quad) { c++; continue; }
        for (int k = 0; k < 4; k++)
          o << "          if (ins_mask & " << (1 << k) << "u) *reinterpret_cast<u32x4*>(d_" << k << " + " << cword[c] << ") = u32x4{i" << c << "_" << k << ", i" << c + 1 << "_" << k << ", i" << c + 2 << "_" << k
            << ", i" << c + 3 << "_" << k << "};\n";
        in_quad[c] = in_quad[c + 1] = in_quad[c + 2] = in_quad[c + 3] = true;
        c += 4;
      }
      o << "        }\n";
      for (size_t c = c0; c < c1; c++) {
        const JitHashCol& C = s.cols[c];
        const std::string r = "w" + std::to_string(c);
        o << "        " << (in_quad[c] ? "if (!canon) " : "") << "{\n          const int W = hc[" << c << "].word;\n";
        if (C.kind == 0) {
          for (int k = 0; k < 4; k++) o << "          if (ins_mask & " << (1 << k) << "u) d_" << k << "[W] = i" << c << "_" << k << ";\n";
        } else if (C.kind == 1) {
          for (int k = 0; k < 4; k++)
            o << "          if (ins_mask & " << (1 << k) << "u) { const unsigned long long y = ((" << r << "_m >> " << k << ") & 1u) ? " << comp8(r, k) << " : 0ull; d_" << k << "[W] = (uint32_t)y; d_" << k
              << "[W + 1] = (uint32_t)(y >> 32); }\n";
        } else {
          for (int k = 0; k < 4; k++) {
            auto col = [&](int ni) -> std::string { if (s.exprs[(size_t)ni].kind == 7) return "((PM" + std::to_string(s.exprs[(size_t)ni].slot) + " >> " + std::to_string(k) + ") & 1u)"; return comp8("x" + std::to_string(s.exprs[(size_t)ni].slot), k); };
            auto colvalid = [&](int ni) { return "((x" + std::to_string(s.exprs[(size_t)ni].slot) + "_m >> " + std::to_string(k) + ") & 1u)"; };
            o << "          if (ins_mask & " << (1 << k) << "u) { const unsigned long long y = " << expr_valid(s.exprs, C.expr_root, col, colvalid) << " ? "
              << expr_bits(s.exprs, C.expr_root, expr_value(s.exprs, C.expr_root, col, colvalid)) << " : 0ull; d_" << k << "[W] = (uint32_t)y; d_" << k << "[W + 1] = (uint32_t)(y >> 32); }\n";
          }
        }
        o << "        }\n";
      }
      // (keep the next group's loads below this point: all 32 columns' loads hoisted to the top cost ≈100 more VGPRs)
      o << "        asm volatile(\"\" ::: \"memory\");\n      }\n";
    }
    if (for_runs) {
      // A canonical tuple's padding words [used, key_words) are part of the run record: a segment only remembers its record width, so a group
      // column the plan gains LATER lands on them and must read 0 = NULL there (the hash table tracks its used prefix instead, h_key_used_).
      int used = 4;
      for (size_t c = 0; c < s.cols.size(); c++) used += s.cols[c].kind == 0 ? 1 : 2;
      if (used % 4 != 0) {
        o << "      if (canon) {\n";
        for (int k = 0; k < 4; k++) {
          o << "        if (ins_mask & " << (1 << k) << "u) {";
          for (int w = used; w < ((used + 3) & ~3); w++) o << " d_" << k << "[" << w << "] = 0u;";
          o << " }\n";
        }
        o << "      }\n";
      }
      const bool has_val = !s.aggs.empty() && s.aggs[0].func != FDB_AGG_COUNT;
      const bool f64v = has_val && s.aggs[0].func == FDB_AGG_SUM && s.aggs[0].type == FDB_T_F64;
      for (int k = 0; k < 4; k++)
        o << "      if (ins_mask & " << (1 << k) << "u) *reinterpret_cast<u64x2*>(d_" << k << " + h.key_words) = u64x2{cnt_" << k << ", "
          << (has_val ? (f64v ? "(unsigned long long)__double_as_longlong(v0_" + std::to_string(k) + ")" : "(unsigned long long)v0_" + std::to_string(k)) : std::string("0ull")) << "};\n";
    }
    o << "    }\n";
    };
    if (wide || medium) {
      // Wide records (and medium ones: the same bookkeeping, the tuple comes out of registers). As in the narrow case every row still selected ends a run of its wave and holds the run's count and folded
      // aggregate; its key tuple is written from re-loaded columns by emit_tuple_stores. The wave's runs wait in its LDS stage
      // (h.runs.stage_cap of them) and leave as one contiguous copy; a tile with more runs than the stage holds (input that is not
      // really ordered) writes them straight into the chunk.
      o << "    {\n      const unsigned long long act2 = __ballot(1);\n      const int wl2 = (int)(tid & 63u), wv = (int)(tid >> 6), first2 = __builtin_ctzll(act2);\n";
      o << "      const unsigned long long b0 = __ballot((sel & 1u) != 0u), b1 = __ballot((sel & 2u) != 0u), b2 = __ballot((sel & 4u) != 0u), b3 = __ballot((sel & 8u) != 0u);\n";
      o << "      const uint32_t n_w = (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));\n";
      o << "      if (n_w != 0u) {\n        const unsigned long long lt = (1ull << wl2) - 1ull;\n";
      o << "        const uint32_t before = (uint32_t)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));\n";
      o << "        uint32_t rb = s_runs[wv * 4], rp = s_runs[wv * 4 + 1], ps = s_runs[wv * 4 + 2];\n";
      o << "        const uint32_t RW = " << (medium ? std::to_string(FDB_RUN_MEDIUM_BYTES / 4) + "u" : std::string("(uint32_t)h.runs.run_words")) << ", RQ = RW >> 2, CAP = (uint32_t)h.runs.stage_cap;\n";
      o << "        u32x4* stage = reinterpret_cast<u32x4*>(smem + a.lds_lut_bytes + (size_t)wv * " << FDB_RUN_WAVE_LDS << ");\n";
      o << "        const uint32_t n_act = (uint32_t)__popcll(act2), my_act = (uint32_t)__popcll(act2 & lt);\n";
      o << "        const bool sw = rp + n_w > " << FDB_RUN_CHUNK << "u, direct = n_w > CAP;\n";
      o << "        if (sw || direct || (rp - ps) + n_w > CAP) {\n";
      o << "          u32x4* out = reinterpret_cast<u32x4*>(h.runs.tuples) + (size_t)(rb + ps) * RQ;\n";
      o << "          for (uint32_t q = my_act; q < (rp - ps) * RQ; q += n_act) out[q] = stage[q];\n";
      o << "          __builtin_amdgcn_wave_barrier();\n          ps = rp;\n        }\n";
      o << "        if (sw) {\n          uint32_t nc = 0u;\n          if (wl2 == first2) nc = atomicAdd(h.runs.chunk_cursor, 1u);\n          rb = (uint32_t)__shfl((int)nc, first2, 64) * " << FDB_RUN_CHUNK
        << "u; rp = 0u; ps = 0u;\n        }\n";
      o << "        const uint32_t base = rb + rp, sbase = rp - ps;\n        __builtin_amdgcn_wave_barrier();\n";
      o << "        if (wl2 == first2) { s_runs[wv * 4] = rb; s_runs[wv * 4 + 1] = rp + n_w; s_runs[wv * 4 + 2] = direct ? rp + n_w : ps; *reinterpret_cast<uint2*>(h.runs.dir + ((size_t)tile * 4 + wv) * 2) = make_uint2(base, n_w); }\n";
      for (int k = 0; k < 4; k++)
        o << "        uint32_t* d_" << k << " = direct ? reinterpret_cast<uint32_t*>(h.runs.tuples) + (size_t)(base + before + (uint32_t)__builtin_popcount(sel & " << ((1 << k) - 1) << "u)) * RW"
          << " : reinterpret_cast<uint32_t*>(stage) + (size_t)(sbase + before + (uint32_t)__builtin_popcount(sel & " << ((1 << k) - 1) << "u)) * RW;\n";
      if (wide) {
        o << "        const uint32_t ins_mask = sel;\n";
        emit_tuple_stores(true);
      } else {
        const bool has_val = !s.aggs.empty() && s.aggs[0].func != FDB_AGG_COUNT;
        const bool f64v = has_val && s.aggs[0].func == FDB_AGG_SUM && s.aggs[0].type == FDB_T_F64;
        for (int k = 0; k < 4; k++) {
          o << "        if (sel & " << (1 << k) << "u) {\n          u32x4* d = reinterpret_cast<u32x4*>(d_" << k << ");\n";
          o << "          d[0] = ta_" << k << "; d[1] = tb_" << k << "; d[2] = tc_" << k << "; d[3] = td_" << k << ";\n";
          o << "          *reinterpret_cast<u64x2*>(d + 4) = u64x2{cnt_" << k << ", "
            << (has_val ? (f64v ? "(unsigned long long)__double_as_longlong(v0_" + std::to_string(k) + ")" : "(unsigned long long)v0_" + std::to_string(k)) : std::string("0ull")) << "};\n        }\n";
        }
      }
      o << "        __builtin_amdgcn_wave_barrier();\n      }\n    }\n    continue;\n";
    }
    if (narrow) {
      // Every row still selected is the LAST row of a run of equal keys inside this wave and holds the run's count and folded
      // aggregate. Position of a run among the wave's runs: rows are ordered lane-major (a lane's 4 rows are consecutive).
      const bool has_val = !s.aggs.empty() && s.aggs[0].func != FDB_AGG_COUNT;
      const bool f64v = has_val && s.aggs[0].func == FDB_AGG_SUM && s.aggs[0].type == FDB_T_F64;
      o << "    {\n      const unsigned long long act2 = __ballot(1);\n      const int wl2 = (int)(tid & 63u), wv = (int)(tid >> 6), first2 = __builtin_ctzll(act2);\n";
      o << "      const unsigned long long b0 = __ballot((sel & 1u) != 0u), b1 = __ballot((sel & 2u) != 0u), b2 = __ballot((sel & 4u) != 0u), b3 = __ballot((sel & 8u) != 0u);\n";
      o << "      const uint32_t n_w = (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));\n";
      o << "      if (n_w != 0u" << ((runs_ablate & 1) ? " && h.row_begin == 12345" : "") << ") {\n        const unsigned long long lt = (1ull << wl2) - 1ull;\n";
      o << "        const uint32_t before = (uint32_t)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));\n";
      // The wave's runs wait in its LDS stage (FDB_RUN_STAGE of them) and leave as one contiguous copy when the stage is full or the
      // chunk changes: a store per tile made the next tile's loads wait for its acknowledgement.
      o << "        uint32_t rb = s_runs[wv * 4], rp = s_runs[wv * 4 + 1], ps = s_runs[wv * 4 + 2];\n";
      o << "        u32x4* stage = reinterpret_cast<u32x4*>(smem + a.lds_lut_bytes + (size_t)wv * " << FDB_RUN_WAVE_LDS << ");\n";
      o << "        const uint32_t n_act = (uint32_t)__popcll(act2), my_act = (uint32_t)__popcll(act2 & lt);\n";
      o << "        const bool sw = rp + n_w > " << FDB_RUN_CHUNK << "u;\n";
      o << "        if (sw || (rp - ps) + n_w > " << FDB_RUN_STAGE << "u) {\n";
      o << "          u32x4* out = reinterpret_cast<u32x4*>(h.runs.tuples) + (size_t)(rb + ps) * 3;\n";
      o << "          for (uint32_t q = my_act; q < (rp - ps) * 3u; q += n_act) out[q] = stage[q];\n";
      o << "          __builtin_amdgcn_wave_barrier();\n          ps = rp;\n        }\n";
      o << "        if (sw) {\n          uint32_t nc = 0u;\n          if (wl2 == first2) nc = atomicAdd(h.runs.chunk_cursor, 1u);\n          rb = (uint32_t)__shfl((int)nc, first2, 64) * " << FDB_RUN_CHUNK
        << "u; rp = 0u; ps = 0u;\n        }\n";
      o << "        const uint32_t base = rb + rp, sbase = rp - ps;\n        __builtin_amdgcn_wave_barrier();\n";
      o << "        if (wl2 == first2) { s_runs[wv * 4] = rb; s_runs[wv * 4 + 1] = rp + n_w; s_runs[wv * 4 + 2] = ps; *reinterpret_cast<uint2*>(h.runs.dir + ((size_t)tile * 4 + wv) * 2) = make_uint2(base, n_w); }\n";
      for (int k = 0; k < 4; k++) {
        o << "        if (sel & " << (1 << k) << "u) {\n          const uint32_t at = sbase + before + (uint32_t)__builtin_popcount(sel & " << ((1 << k) - 1) << "u);\n";
        o << "          stage[at * 3] = ta_" << k << "; stage[at * 3 + 1] = tb_" << k << ";\n";
        o << "          *reinterpret_cast<u64x2*>(stage + at * 3 + 2) = u64x2{cnt_" << k << ", "
          << (has_val ? (f64v ? "(unsigned long long)__double_as_longlong(v0_" + std::to_string(k) + ")" : "(unsigned long long)v0_" + std::to_string(k)) : std::string("0ull")) << "};\n        }\n";
      }
      o << "        __builtin_amdgcn_wave_barrier();\n      }\n    }\n    continue;\n";
    }
    // Probe: the home entries of the 4 rows are requested together (one memory round trip for the common case "group exists
    // and sits in its home slot"); rows that miss there take the general find-or-insert path.
    for (int k = 0; k < 4; k++) o << "    uint64_t slot_" << k << " = h1_" << k << " & h.mask; unsigned long long p_" << k << " = 0, q_" << k << " = 0;\n";
    // (one plain 16-byte load per row: a fingerprint never changes once written, so a stale cached copy can only make the row
    // miss here and take the general path, whose loads are coherent — never match wrongly)
    for (int k = 0; k < 4; k++)
      o << "    if ((sel >> " << k << ") & 1u) { const u64x2 pq = *as_global(reinterpret_cast<const u64x2*>(h.table + slot_" << k << " * (uint64_t)ew)); p_" << k << " = pq.x; q_" << k
        << " = pq.y; }\n";
    // Rows whose home entry looked EMPTY claim it right away — the (up to) four compare-and-swaps of a lane are in flight together
    // instead of one dependent load + CAS per row — and every winner publishes the high half of its fingerprint before any lane
    // of the wave enters the general path below (a lane that lost to a lane of its own wave spins there until it sees the
    // published half: the publication must not sit behind that loop in a sibling branch).
    o << "    uint32_t ins_mask = 0;\n";
    for (int k = 0; k < 4; k++) o << "    unsigned long long c_" << k << " = ~0ull;\n";
    for (int k = 0; k < 4; k++)
      o << "    if (((sel >> " << k << ") & 1u) && p_" << k << " == 0ull) c_" << k << " = atomicCAS(h.table + slot_" << k << " * (uint64_t)ew, 0ull, h1_" << k << ");\n";
    for (int k = 0; k < 4; k++)
      o << "    if (c_" << k << " == 0ull) { __hip_atomic_store(h.table + slot_" << k << " * (uint64_t)ew + 1, h2_" << k << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ins_mask |= " << (1 << k) << "u; }\n";
    for (int k = 0; k < 4; k++) {
      o << "    if ((sel >> " << k << ") & 1u) {\n";
      o << "      if (c_" << k << " != 0ull && !(p_" << k << " == h1_" << k << " && q_" << k << " == h2_" << k << ")) { bool ins; slot_" << k << " = hash_find_or_insert(h.table, h.mask, ew, h1_" << k << ", h2_" << k
        << ", ins); if (ins) ins_mask |= " << (1 << k) << "u; }\n";
      o << "      unsigned long long* e = h.table + slot_" << k << " * (uint64_t)ew;\n";
      if (!(s.ablate & 2) && s.need_count) o << "      atomicAdd(e + 2, cnt_" << k << ");\n";
      for (size_t j = 0; j < s.aggs.size(); j++) {
        const JitAgg& A = s.aggs[j];
        if (A.func == FDB_AGG_COUNT || (s.ablate & 2)) continue;
        const std::string v = "v" + std::to_string(j) + "_" + std::to_string(k);
        const std::string acc = "(e + " + std::to_string(3 + j) + ")";
        if (A.func == FDB_AGG_SUM && A.type == FDB_T_F64) o << "      atomicAdd(reinterpret_cast<double*>" << acc << ", " << v << ");\n";
        else if (A.func == FDB_AGG_SUM) o << "      atomicAdd(" << acc << ", " << v << ");\n";
        else o << "      " << (A.func == FDB_AGG_MIN ? "atomicMin" : "atomicMax") << "(reinterpret_cast<long long*>" << acc << ", " << v << ");\n";
      }
      o << "    }\n";
    }
    // Key tuples of the groups this lane created. The lane's 4 rows are consecutive, so one 16-byte load per column (two for
    // 8-byte keys) brings the values of all of them; 8 columns' loads are in flight together, then their key-id lookups, then
    // the stores — 4 memory round trips for 32 columns however many of the lane's rows inserted (one row at a time, re-reading
    // column by column, cost 2 round trips per 16 columns and inserted row and made insert-heavy launches 5 × slower per row).
    // Canonical layout (h.canonical: the record carries every group column of the table, in order): column c's word is the
    // constant 4 + Σ widths before it, tuples are 16-byte aligned, and an aligned quad of dictionary columns is ONE 16-byte store —
    // every store of a scattered tuple is its own write request, and with 34 four-byte stores those were most of an insert's cost.
    emit_tuple_stores(false);
    o << "  }\n";
    if (wide || medium) {
      o << "  {  // the runs that still wait in the waves' stages\n    const int wv = (int)(tid >> 6), wl2 = (int)(tid & 63u);\n    __builtin_amdgcn_wave_barrier();\n";
      o << "    const uint32_t rb = s_runs[wv * 4], rp = s_runs[wv * 4 + 1], ps = s_runs[wv * 4 + 2], RQ = " << (medium ? std::to_string(FDB_RUN_MEDIUM_BYTES / 16) + "u" : std::string("(uint32_t)h.runs.run_words >> 2")) << ";\n";
      o << "    if (rp != " << FDB_RUN_CHUNK << "u || ps != " << FDB_RUN_CHUNK << "u) {\n";
      o << "      const u32x4* stage = reinterpret_cast<const u32x4*>(smem + a.lds_lut_bytes + (size_t)wv * " << FDB_RUN_WAVE_LDS << ");\n";
      o << "      u32x4* out = reinterpret_cast<u32x4*>(h.runs.tuples) + (size_t)(rb + ps) * RQ;\n";
      o << "      for (uint32_t q = (uint32_t)wl2; q < (rp - ps) * RQ; q += 64u) out[q] = stage[q];\n    }\n  }\n";
    }
    if (narrow) {
      o << "  {  // the runs that still wait in the waves' stages\n    const int wv = (int)(tid >> 6), wl2 = (int)(tid & 63u);\n    __builtin_amdgcn_wave_barrier();\n";
      o << "    const uint32_t rb = s_runs[wv * 4], rp = s_runs[wv * 4 + 1], ps = s_runs[wv * 4 + 2];\n";
      o << "    if (rp != " << FDB_RUN_CHUNK << "u || ps != " << FDB_RUN_CHUNK << "u) {\n";
      o << "      const u32x4* stage = reinterpret_cast<const u32x4*>(smem + a.lds_lut_bytes + (size_t)wv * " << FDB_RUN_WAVE_LDS << ");\n";
      o << "      u32x4* out = reinterpret_cast<u32x4*>(h.runs.tuples) + (size_t)(rb + ps) * 3;\n";
      o << "      for (uint32_t q = (uint32_t)wl2; q < (rp - ps) * 3u; q += 64u) out[q] = stage[q];\n    }\n  }\n";
    }
    o << "  __syncthreads();\n  if (tid == 0 && s_new != 0) atomicAdd(h.n_groups, (unsigned long long)s_new);\n}\n";
    return o.str();
  }
};

}  // namespace

std::string JitShape::key(bool with_validity) const {
  std::ostringstream k;
  k << "b" << block << "l" << lds_acc << "c" << need_count << "t" << two_phase << "r" << reg_slots << "h" << cache << (wave_tables ? "w" : "") << (uniform_fold ? "u" : "") << "|";
  auto slots = [&](const JitSlot* p, int n) { for (int i = 0; i < n; i++) k << (p[i].has_values ? 'v' : '-') << (with_validity ? p[i].has_validity : 0); k << '|'; };
  slots(c4, n_c4); slots(c8, n_c8); slots(l4, n_l4); slots(l8, n_l8);
  for (const JitLeaf& L : leaves) k << L.kind << ',' << L.slot << ',' << L.wide << ',' << (L.kind >= FDB_LEAF_CMP_I64 && L.kind <= FDB_LEAF_CMP_I64_F64 ? L.op : 0) << ',' << L.lut_in_lds << ';';
  k << '|';
  for (uint8_t c : code) k << (int)c << ',';
  k << '|';
  for (const JitGroup& G : gcols) k << G.slot << ',' << G.lut_in_lds << ';';
  k << '|';
  for (const JitAgg& A : aggs) k << A.func << ',' << A.type << ',' << A.slot << ',' << A.expr << ';';
  k << '|';
  for (const JitExprNode& e : exprs) k << e.kind << ',' << e.op << ',' << e.left << ',' << e.right << ',' << e.slot << ',' << e.type << ';';
  return k.str();
}

std::string jit_source(const JitShape& shape) { return Gen(shape).source(); }

JitShape jit_shape(const FdbScanArgs& a, bool two_phase, int block) {
  JitShape s;
  s.block = block;
  s.lds_acc = a.lds_acc != 0;
  s.need_count = a.need_count != 0;
  s.two_phase = two_phase;
  s.n_c4 = a.n_c4; s.n_c8 = a.n_c8;
  s.n_l4 = two_phase ? a.n_l4 : 0; s.n_l8 = two_phase ? a.n_l8 : 0;
  auto slots = [](JitSlot* dst, const FdbColSlot* src, int n) {
    for (int i = 0; i < n; i++) { dst[i].has_values = src[i].values != nullptr; dst[i].has_validity = src[i].validity != nullptr ? 1 : 0; }
  };
  slots(s.c4, a.c4, s.n_c4); slots(s.c8, a.c8, s.n_c8); slots(s.l4, a.l4, s.n_l4); slots(s.l8, a.l8, s.n_l8);
  for (int l = 0; l < a.n_leaves; l++) {
    const FdbLeaf& L = a.leaves[l];
    s.leaves.push_back({L.kind, L.slot, L.wide, L.op, L.kind == FDB_LEAF_DICT_LUT && L.lut_lds != FDB_NO_LDS});
  }
  s.code.assign(a.code, a.code + a.n_code);
  for (int g = 0; g < a.n_gcols; g++) s.gcols.push_back({a.gcols[g].slot, a.gcols[g].lut_lds != FDB_NO_LDS});
  for (int j = 0; j < a.n_aggs; j++) s.aggs.push_back({a.aggs[j].func, a.aggs[j].type, a.aggs[j].slot, a.aggs[j].expr});
  for (int i = 0; i < a.n_expr; i++) s.exprs.push_back({a.expr[i].kind, a.expr[i].op, a.expr[i].left, a.expr[i].right, a.expr[i].slot, a.expr[i].type});
  s.cache = !s.lds_acc && a.cache_slots > 0;
  return s;
}

bool jit_shape_merge(JitShape* into, const JitShape& other) {
  if (into->key(false) != other.key(false)) return false;
  auto slots = [](JitSlot* a, const JitSlot* b, int n) { for (int i = 0; i < n; i++) if (a[i].has_validity != b[i].has_validity) a[i].has_validity = 2; };
  slots(into->c4, other.c4, into->n_c4); slots(into->c8, other.c8, into->n_c8); slots(into->l4, other.l4, into->n_l4); slots(into->l8, other.l8, into->n_l8);
  return true;
}

// jit_shape(b) would merge into `into` (= the shape of `a`, possibly merged with others already): the comparison key(false) makes,
// field by field on the argument blocks themselves — a scan of 40 records built 80 key strings for this (≈ 1 µs per record, on the
// critical path in front of the launch) — and the merge of the validity flags.
bool jit_shape_merge_args(JitShape* into, const FdbScanArgs& a, const FdbScanArgs& b, bool two_phase) {
  if (a.lds_acc != b.lds_acc || a.need_count != b.need_count || a.n_c4 != b.n_c4 || a.n_c8 != b.n_c8) return false;
  if (two_phase && (a.n_l4 != b.n_l4 || a.n_l8 != b.n_l8)) return false;
  if ((a.lds_acc == 0 && a.cache_slots > 0) != (b.lds_acc == 0 && b.cache_slots > 0)) return false;
  auto values_same = [](const FdbColSlot* x, const FdbColSlot* y, int n) { for (int i = 0; i < n; i++) if ((x[i].values != nullptr) != (y[i].values != nullptr)) return false; return true; };
  if (!values_same(a.c4, b.c4, a.n_c4) || !values_same(a.c8, b.c8, a.n_c8)) return false;
  if (two_phase && (!values_same(a.l4, b.l4, a.n_l4) || !values_same(a.l8, b.l8, a.n_l8))) return false;
  if (a.n_leaves != b.n_leaves || a.n_code != b.n_code || a.n_gcols != b.n_gcols || a.n_aggs != b.n_aggs || a.n_expr != b.n_expr) return false;
  for (int l = 0; l < a.n_leaves; l++) {
    const FdbLeaf& x = a.leaves[l]; const FdbLeaf& y = b.leaves[l];
    if (x.kind != y.kind || x.slot != y.slot || x.wide != y.wide) return false;
    if (x.kind >= FDB_LEAF_CMP_I64 && x.kind <= FDB_LEAF_CMP_I64_F64 && x.op != y.op) return false;
    if ((x.kind == FDB_LEAF_DICT_LUT && x.lut_lds != FDB_NO_LDS) != (y.kind == FDB_LEAF_DICT_LUT && y.lut_lds != FDB_NO_LDS)) return false;
  }
  if (std::memcmp(a.code, b.code, (size_t)a.n_code) != 0) return false;
  for (int g = 0; g < a.n_gcols; g++)
    if (a.gcols[g].slot != b.gcols[g].slot || (a.gcols[g].lut_lds != FDB_NO_LDS) != (b.gcols[g].lut_lds != FDB_NO_LDS)) return false;
  for (int j = 0; j < a.n_aggs; j++)
    if (a.aggs[j].func != b.aggs[j].func || a.aggs[j].type != b.aggs[j].type || a.aggs[j].slot != b.aggs[j].slot || a.aggs[j].expr != b.aggs[j].expr) return false;
  for (int i = 0; i < a.n_expr; i++) {
    const auto& x = a.expr[i]; const auto& y = b.expr[i];
    if (x.kind != y.kind || x.op != y.op || x.left != y.left || x.right != y.right || x.slot != y.slot || x.type != y.type) return false;
  }
  auto merge = [](JitSlot* s, const FdbColSlot* y, int n) { for (int i = 0; i < n; i++) if (s[i].has_validity != (y[i].validity != nullptr ? 1 : 0)) s[i].has_validity = 2; };
  merge(into->c4, b.c4, into->n_c4); merge(into->c8, b.c8, into->n_c8); merge(into->l4, b.l4, into->n_l4); merge(into->l8, b.l8, into->n_l8);
  return true;
}

int jit_blocks_per_cu(hipFunction_t fn, int block, size_t lds_bytes) {
  int occ = 0;
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, block, lds_bytes) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 1; }
  if (occ > 2048 / block) occ = 2048 / block;
  return occ;
}

namespace {
// Process cache → disk cache → hiprtc; `key` identifies the shape, `make_source` is only called on a process-cache miss.
std::atomic<int64_t> g_stat_compiled{0}, g_stat_compile_us{0}, g_stat_disk_loads{0};

// Disk cache → hiprtc → module, for the entry its caller owns; whatever happens the entry is completed (a waiter must not wait for
// ever — a failed build publishes nullptr = "interpret").
hipFunction_t build_kernel(KernelEntry* entry, const std::string& kernel_name_s, const std::string& src) {
  const char* kernel_name = kernel_name_s.c_str();
  struct Publish {
    KernelEntry* e; hipFunction_t fn = nullptr;
    ~Publish() { { std::lock_guard<std::mutex> lk(e->m); e->fn = fn; e->ready = true; } e->cv.notify_all(); }
  } publish{entry};
  std::vector<char> code;
  char name[64];
  std::snprintf(name, sizeof name, "/k_%016llx_%zu.hsaco", (unsigned long long)(fnv(src) ^ (fnv(kKernelsHeader) * 31)), src.size());
  const std::string dir = cache_dir();
  const std::string path = dir.empty() ? std::string() : dir + name;
  bool from_disk = false;
  if (!path.empty()) {
    std::ifstream f(path, std::ios::binary);
    if (f) code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    from_disk = !code.empty();
  }
  auto load = [&](hipFunction_t* fn) {
    hipModule_t mod = nullptr;
    *fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(fn, mod, kernel_name) == hipSuccess) return true;
    (void)hipGetLastError();
    *fn = nullptr;
    return false;
  };
  hipFunction_t fn = nullptr;
  if (from_disk && !load(&fn)) {  // truncated / stale / foreign file: drop it and compile afresh instead of living on the fallback forever
    (void)::unlink(path.c_str());
    code.clear();
    from_disk = false;
  }
  if (from_disk) g_stat_disk_loads++;
  if (code.empty()) {
    std::string log;
    bool compiled;
    {
      // (one compilation at a time: the compiler is not known to tolerate concurrent programs, and lookups no longer wait behind it)
      static std::mutex compile_mu;
      std::lock_guard<std::mutex> lk(compile_mu);
      const auto t0 = std::chrono::steady_clock::now();
      compiled = compile(src, &code, &log);
      g_stat_compile_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    g_stat_compiled++;
    if (!compiled) {
      std::fprintf(stderr, "[frostdb_amd] %s specialisation failed, using the interpreting kernel: %s\n", kernel_name, log.c_str());
      if (std::getenv("FDB_JIT_DEBUG")) std::fprintf(stderr, "%s\n", src.c_str());
      return nullptr;
    }
    if (!path.empty()) {  // publish atomically, and only a file that was written completely
      const std::string tmp = path + "." + std::to_string((long)getpid());
      bool ok = false;
      {
        std::ofstream f(tmp, std::ios::binary);
        if (f) { f.write(code.data(), (std::streamsize)code.size()); f.close(); ok = !f.fail(); }
      }
      if (!ok || ::rename(tmp.c_str(), path.c_str()) != 0) (void)::unlink(tmp.c_str());
    }
  }
  if (fn == nullptr && !load(&fn)) {
    std::fprintf(stderr, "[frostdb_amd] could not load a specialised %s, using the interpreting kernel\n", kernel_name);
    fn = nullptr;
  }
  publish.fn = fn;
  return fn;
}

// $FDB_JIT_ASYNC=1 (opt-in): a caller that CAN do without the specialised kernel (jit_may_defer: the interpreting kernels serve its
// launch) does not wait for a kernel that has to be built first — the build runs on a thread of this pool, the call returns nullptr
// ("interpret"), and the next query of the shape finds the kernel. The first query of a new shape then costs an interpreted scan
// instead of 130–500 ms of compiler. The threads are joined before the cache they publish into is destroyed.
thread_local bool t_may_defer = false;
struct Builders {
  struct Slot { std::thread t; std::shared_ptr<std::atomic<bool>> done; };
  std::mutex m;
  std::vector<Slot> slots;
  void spawn(std::function<void()> job) {
    std::lock_guard<std::mutex> lk(m);
    // (threads that have finished are joined here: an unjoined thread keeps its stack, and a long-lived process meets many shapes)
    for (size_t k = 0; k < slots.size();) {
      if (slots[k].done->load(std::memory_order_acquire)) { slots[k].t.join(); slots[k] = std::move(slots.back()); slots.pop_back(); }
      else k++;
    }
    auto done = std::make_shared<std::atomic<bool>>(false);
    slots.push_back(Slot{std::thread([job = std::move(job), done] { job(); done->store(true, std::memory_order_release); }), done});
  }
  ~Builders() { for (Slot& s : slots) if (s.t.joinable()) s.t.join(); }
} g_builders;

template <typename F>
hipFunction_t get_kernel(const std::string& key, const char* kernel_name, F make_source) {
  if (g_disabled || std::getenv("FDB_NO_JIT") != nullptr) return nullptr;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::string ckey = std::to_string(dev) + "|" + kernel_name + "|" + key;  // a loaded module belongs to one device
  std::shared_ptr<KernelEntry> entry;
  bool builder = false;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(ckey);
    if (it == g_cache.end()) { entry = std::make_shared<KernelEntry>(); g_cache.emplace(ckey, entry); builder = true; }
    else entry = it->second;
  }
  const char* as = t_may_defer ? std::getenv("FDB_JIT_ASYNC") : nullptr;
  const bool defer = as != nullptr && std::atoi(as) != 0;
  if (!builder) {
    std::unique_lock<std::mutex> lk(entry->m);
    if (defer && !entry->ready) return nullptr;  // still being built: this launch is interpreted
    entry->cv.wait(lk, [&] { return entry->ready; });
    return entry->fn;
  }
  std::string src;
  try { src = make_source(); }
  catch (...) { { std::lock_guard<std::mutex> lk(entry->m); entry->ready = true; } entry->cv.notify_all(); throw; }
  if (defer) {
    const std::string kn = kernel_name;
    g_builders.spawn([entry, kn, src, dev] { (void)hipSetDevice(dev); (void)build_kernel(entry.get(), kn, src); });
    return nullptr;
  }
  return build_kernel(entry.get(), kernel_name, src);
}
}  // namespace

std::string jit_select_source(const JitShape& shape) { return Gen(shape).select_source(); }
size_t jit_select_stage_bytes(const JitShape& shape) { size_t n = 0; (void)Gen(shape).fused(&n); return n; }
hipFunction_t jit_select_kernel_get(const JitShape& shape) {
  return get_kernel("select|" + shape.key() + "|f" + std::to_string(shape.fuse4) + "," + std::to_string(shape.fuse8) + (std::getenv("FDB_SELECT_ABLATE") ? std::string("|a") + std::getenv("FDB_SELECT_ABLATE") : std::string()) + (std::getenv("FDB_SELECT_SLEEP") ? std::string("|s") + std::getenv("FDB_SELECT_SLEEP") : std::string()) + (std::getenv("FDB_SELECT_SCAN_SLEEP") ? std::string("|S") + std::getenv("FDB_SELECT_SCAN_SLEEP") : std::string()), "fdb_select_kernel", [&] { return jit_select_source(shape); });
}
hipError_t jit_select_launch(hipFunction_t fn, const FdbScanArgs* d_parts, int n_parts, int64_t total_super_tiles, const FdbScanArgs& common, int grid, size_t lds_bytes,
                             uint32_t* masks, uint32_t* offsets, const FdbSelectArgs& sel, hipStream_t stream) {
  long long tt = total_super_tiles;
  void* args[] = {(void*)&d_parts, (void*)&n_parts, (void*)&tt, (void*)&common, (void*)&masks, (void*)&offsets, (void*)&sel};
  return hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)Gen::kSelectBlock, 1, 1, (unsigned)lds_bytes, stream, args, nullptr);
}
int jit_select_block() { return Gen::kSelectBlock; }

std::string jit_flags_source(const JitShape& shape) { return Gen(shape).flags_source(); }
hipFunction_t jit_flags_get(const JitShape& shape) {
  return get_kernel("flags|" + shape.key(), "fdb_flags_kernel", [&] { return jit_flags_source(shape); });
}

hipError_t jit_flags_launch(hipFunction_t fn, const FdbScanArgs* d_parts, int n_parts, int64_t total_super_tiles, const FdbScanArgs& common, int grid, size_t lds_bytes,
                            uint32_t* masks, uint32_t* tile_counts, hipStream_t stream) {
  long long tt = total_super_tiles;
  void* args[] = {(void*)&d_parts, (void*)&n_parts, (void*)&tt, (void*)&common, (void*)&masks, (void*)&tile_counts};
  return hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256u, 1, 1, (unsigned)lds_bytes, stream, args, nullptr);
}

void jit_stats(int64_t* n_compiled, double* compile_ms, int64_t* n_disk_loads) {
  if (n_compiled) *n_compiled = g_stat_compiled.load();
  if (compile_ms) *compile_ms = (double)g_stat_compile_us.load() / 1000.0;
  if (n_disk_loads) *n_disk_loads = g_stat_disk_loads.load();
}

JitDeferScope::JitDeferScope(bool may_defer) : prev_(t_may_defer) { t_may_defer = may_defer; }
JitDeferScope::~JitDeferScope() { t_may_defer = prev_; }

hipFunction_t jit_get(const JitShape& shape) {
  return get_kernel(shape.key(), "fdb_plan_kernel", [&] { return jit_source(shape); });
}

std::string JitHashShape::key() const {
  std::ostringstream k;
  k << "a" << ablate << "c" << need_count << (runs == 1 ? "R" : runs == 2 ? "W" : runs == 3 ? "M" : "") << (runs && std::getenv("FDB_RUNS_WAVES_PER_EU") ? std::getenv("FDB_RUNS_WAVES_PER_EU") : "")
    << (runs && std::getenv("FDB_RUNS_ABLATE") ? std::string("x") + std::getenv("FDB_RUNS_ABLATE") : std::string())
    << (temporal_loads(runs) ? "t" : "") << "|";
  for (const JitHashCol& C : cols) k << C.kind << (C.has_validity ? 'n' : '-') << (C.lut_identity ? 'i' : C.lut_in_lds ? 'l' : 'g') << (C.kind == 2 ? std::to_string(C.expr_root) : std::string());
  k << '|';
  for (size_t l = 0; l < leaves.size(); l++) {
    const JitLeaf& L = leaves[l];
    k << L.kind << ',' << L.wide << ',' << (L.kind >= FDB_LEAF_CMP_I64 && L.kind <= FDB_LEAF_CMP_I64_F64 ? L.op : 0) << ',' << L.lut_in_lds << ',' << leaf_validity[l] << ';';
  }
  k << '|';
  for (uint8_t c : code) k << (int)c << ',';
  k << '|';
  for (size_t j = 0; j < aggs.size(); j++) k << aggs[j].func << ',' << aggs[j].type << ',' << agg_validity[j] << ',' << aggs[j].expr << ';';
  k << '|' << n_expr_cols << '|';
  for (const JitExprNode& e : exprs) k << e.kind << ',' << e.op << ',' << e.left << ',' << e.right << ',' << e.slot << ',' << e.type << ';';
  return k.str();
}

JitHashShape jit_hash_shape(const FdbHashArgs& h, const FdbHashCol* hcols) {
  JitHashShape s;
  const FdbScanArgs& a = h.base;
  for (int c = 0; c < h.n_hcols; c++)
    s.cols.push_back({hcols[c].kind, hcols[c].validity != nullptr, hcols[c].kind == 0 && hcols[c].lut_lds != FDB_NO_LDS, hcols[c].kind == 2 ? hcols[c].src_word : -1,
                      hcols[c].kind == 0 && hcols[c].lut == nullptr});
  for (int i = 0; i < a.n_expr; i++) s.exprs.push_back({a.expr[i].kind, a.expr[i].op, a.expr[i].left, a.expr[i].right, a.expr[i].slot, a.expr[i].type});
  s.n_expr_cols = a.n_l8;
  s.need_count = a.need_count != 0;
  s.runs = h.runs.tuples == nullptr ? 0 : h.runs.run_words == 0 ? 1 : h.runs.run_words == FDB_RUN_MEDIUM_WORDS ? 3 : 2;
  for (int l = 0; l < a.n_leaves; l++) {
    const FdbLeaf& L = a.leaves[l];
    const bool wide = L.kind >= FDB_LEAF_CMP_I64 && L.kind <= FDB_LEAF_CMP_I64_F64;
    s.leaves.push_back({L.kind, 0, wide ? 1 : 0, L.op, L.kind == FDB_LEAF_DICT_LUT && L.lut_lds != FDB_NO_LDS});
    s.leaf_validity.push_back(L.validity != nullptr);
  }
  s.code.assign(a.code, a.code + a.n_code);
  for (int j = 0; j < a.n_aggs; j++) {
    s.aggs.push_back({a.aggs[j].func, a.aggs[j].type, -1, a.aggs[j].expr});
    s.agg_validity.push_back(a.aggs[j].validity != nullptr);
  }
  return s;
}

std::string jit_hash_source(const JitHashShape& shape) { return HashGen(shape).source(); }

hipFunction_t jit_hash_get(const JitHashShape& shape) {
  return get_kernel(shape.key(), "fdb_hash_kernel", [&] { return jit_hash_source(shape); });
}

hipError_t jit_hash_launch(hipFunction_t fn, const FdbHashArgs& args, int grid, size_t lds_bytes, hipStream_t stream) {
  void* kargs[] = {(void*)&args};
  return hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, (unsigned)lds_bytes, stream, kargs, nullptr);
}

// Geometry for `shape`. Measured on MI355X (tools/sweep_geometry.sh): a streaming scan is fastest when ≈64 KB of loads
// are in flight per CU (Little's law for ≈8 TB/s × ≈2 µs over 256 CUs) — 16 waves for cfg 2 (64 B per lane and tile),
// 8 for cfg 3 (128 B) — and gets SLOWER with more resident waves (cfg 2: 6.75 TB/s at 16 waves/CU, 6.0 at 32), so the
// persistent grid is sized from the bytes one lane requests per tile, not from the occupancy limit.
hipFunction_t jit_select(JitShape shape, size_t lds_bytes, int row_bytes, int* block_out, int* blocks_per_cu_out) {
  int waves = row_bytes > 0 ? (65536 / (64 * 4)) / row_bytes : 16;  // 64 lanes × 4 rows per lane and tile
  waves = std::max(8, std::min(32, waves));
  const int block = waves % 8 == 0 ? 512 : 256;
  shape.block = block;
  hipFunction_t fn = jit_get(shape);
  if (fn == nullptr) return nullptr;
  const int fit = jit_blocks_per_cu(fn, block, lds_bytes);
  *block_out = block;
  *blocks_per_cu_out = std::max(1, std::min(fit, (waves + block / 128) / (block / 64)));
  if (std::getenv("FDB_JIT_DEBUG"))
    std::fprintf(stderr, "[frostdb_amd] plan kernel %s: %d B/row, %zu B LDS -> %d-thread workgroups, %d per CU (%d fit)\n", shape.key().c_str(), row_bytes,
                 lds_bytes, block, *blocks_per_cu_out, fit);
  return fn;
}

hipError_t jit_launch(hipFunction_t fn, const FdbScanArgs* d_parts, int n_parts, int64_t total_tiles, const FdbScanArgs& common, int grid, int block,
                      size_t lds_bytes, hipStream_t stream) {
  long long tt = total_tiles;
  void* args[] = {(void*)&d_parts, (void*)&n_parts, (void*)&tt, (void*)&common};
  return hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds_bytes, stream, args, nullptr);
}

}  // namespace fdb
