#!/bin/bash
# Host-only memory-safety + termination check of the Parquet chunk parser (fdb_parquet.cpp) under AddressSanitizer + UBSan:
# the file is compiled with g++ next to a shim that stubs the kernel launchers (parsing happens before any of them) and driven with
# thousands of mutated column chunks (every codec, page version, DELTA and PLAIN byte-array pages). No GPU needed.
#   tools/asan_parquet.sh [mutations per variant, default 150] [seed, default 7]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/fdb_asan_parquet
mkdir -p "$OUT"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    -I"$ROOT/include" -I"$ROOT/frostdb_amd/csrc" "$ROOT/tools/asan_parquet_shim.cpp" "$ROOT/frostdb_amd/csrc/fdb_parquet.cpp" \
    "$ROOT/frostdb_amd/csrc/fdb_arrow.cpp" "$ROOT/frostdb_amd/csrc/fdb_context.cpp" -L/opt/rocm/lib -lamdhip64 -lz -ldl -lpthread \
    -o "$OUT/libpqasan.so"
FDB_ASAN_LIB="$OUT/libpqasan.so" LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    ASAN_OPTIONS=detect_leaks=0 timeout 3000 python "$ROOT/tools/asan_parquet_run.py" "${1:-150}" "${2:-7}"
