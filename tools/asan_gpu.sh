#!/bin/bash
# The library's HOST code under AddressSanitizer + UBSan, run against a GPU: the paths tools/asan_full.sh cannot reach without one
# (record slab and staging rings of fdb_plan_push, the run store's Finish, filter()'s host side, the communicator). Built with g++
# and gcc's sanitizer runtime — ROCm clang's runtime intercepts hsa_amd_memory_pool_allocate and aborts on a GPU box ("out of
# memory" in the first HIP call); the host sources contain no device code, fdb_kernels.o (hipcc, not instrumented) is linked as is.
#   here (no GPU):  tools/asan_gpu.sh build        → tools/_asan/libfdb_fullasan.so (git-ignored, travels with gpurun)
#   on the GPU box: tools/asan_gpu.sh run [pytest args, default: a selection of -m gpu tests]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT="$ROOT/tools/_asan"
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"
  SRC="$ROOT/frostdb_amd/csrc"
  python -c "import sys; sys.path.insert(0, '$ROOT'); from frostdb_amd import build; build.build()" > /dev/null   # (fdb_kernels.o, fdb_sort.o, fdb_kernels_h.inc)
  pids=()
  for f in fdb_arrow fdb_context fdb_plan fdb_hash fdb_jit fdb_dynamic fdb_comm fdb_parquet fdb_regex fdb_capi; do
    g++ -std=c++17 -O1 -g1 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$SRC/$f.cpp" -o "$OUT/$f.o" & pids+=($!)
  done
  g++ -std=c++17 -O1 -g1 -fPIC -fsanitize=address,undefined -c "$SRC/fdb_widen.cc" -o "$OUT/fdb_widen.o" & pids+=($!)
  for p in "${pids[@]}"; do wait $p; done
  g++ -shared -fPIC -fsanitize=address,undefined -o "$OUT/libfdb_fullasan.so" "$OUT"/*.o "$SRC/fdb_kernels.o" "$SRC/fdb_merge.o" "$SRC/fdb_sort.o" -L/opt/rocm/lib -lamdhip64 -lhiprtc -ldl -lpthread -lz -Wl,-rpath,/opt/rocm/lib
  rm -f "$OUT"/*.o
  ls -la "$OUT"
  exit 0
fi
shift || true
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
export FDB_LIB_PATH="$OUT/libfdb_fullasan.so"
cd "$ROOT"
if [ $# -gt 0 ]; then python -m pytest "$@"; else
# (tests that initialise torch.cuda are left out: under the preloaded runtime torch's dlopen of libcaffe2_nvrtc.so loses its RPATH)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ordered.py -m gpu -q -p no:cacheprovider \
  -k "(chains or host_record or push or one_pass or filter_of_many or run_path or table_free or ordered or exchange or allreduce or lazy or nan) and not rccl"
fi
