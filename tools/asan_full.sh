#!/bin/bash
# The WHOLE library with its host code under AddressSanitizer + UBSan (hipcc -fsanitize=address,undefined -fno-gpu-sanitize; device
# code is compiled as usual), built outside the tree, and the three CPU fuzzers run against it: plan descriptors through
# fdb_plan_explain, Arrow records with detectable defects through fdb_arrow_roundtrip, mutated Parquet column chunks through
# fdb_batch_from_parquet. No GPU needed (every path stops before, or at, the first device call).
#   tools/asan_full.sh [descriptors, default 20000] [arrow records, default 5000] [parquet mutations per variant, default 100] [regex patterns, default 20000]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/fdb_asan_full
mkdir -p "$OUT"
SRC="$ROOT/frostdb_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
python -c "import sys; sys.path.insert(0, '$ROOT'); from frostdb_amd import build; build.build()" > /dev/null   # (writes fdb_kernels_h.inc)
for f in fdb_arrow fdb_context fdb_plan fdb_hash fdb_jit fdb_dynamic fdb_comm fdb_parquet fdb_regex fdb_capi; do
  $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -munsafe-fp-atomics -fsanitize=address,undefined -fno-gpu-sanitize -x hip -c "$SRC/$f.cpp" -o "$OUT/$f.o"
done
$HIPCC --offload-arch=gfx950 -O1 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-sanitize -c "$SRC/fdb_kernels.hip" -o "$OUT/fdb_kernels.o"
$HIPCC --offload-arch=gfx950 -O1 -std=c++17 -fPIC -fno-gpu-sanitize -c "$SRC/fdb_sort.hip" -o "$OUT/fdb_sort.o"
$HIPCC --offload-arch=gfx950 -O1 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-sanitize -c "$SRC/fdb_merge.hip" -o "$OUT/fdb_merge.o"
$HIPCC -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -c "$SRC/fdb_widen.cc" -o "$OUT/fdb_widen.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$OUT/libfdb_fullasan.so" "$OUT"/*.o -lhiprtc -ldl -lpthread -lz
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0
FDB_FUZZ_LIB="$OUT/libfdb_fullasan.so" python "$ROOT/tools/desc_fuzz.py" "${1:-20000}" 1
FDB_FUZZ_LIB="$OUT/libfdb_fullasan.so" python "$ROOT/tools/arrow_fuzz.py" "${2:-5000}" 1
FDB_ASAN_LIB="$OUT/libfdb_fullasan.so" python "$ROOT/tools/asan_parquet_run.py" "${3:-100}" 1
FDB_ASAN_LIB="$OUT/libfdb_fullasan.so" python "$ROOT/tools/regex_fuzz.py" "${4:-20000}" 1
