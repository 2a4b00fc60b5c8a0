#!/bin/bash
# Host-only memory-safety check of the Arrow C data import / export code (fdb_arrow.cpp: view_record, read_dictionary,
# encode_plain, set_dictionary, set_plain_strings, export_record) under AddressSanitizer + UBSan. No GPU, no HIP: the file is
# compiled with g++ next to a small shim that exports fdb_arrow_roundtrip, and driven from pyarrow with every supported column
# type, sliced records and bit-unaligned bitmaps. Prints "asan roundtrip ok".
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/fdb_asan
mkdir -p "$OUT"
sed "s#/tmp/asan/libfdb_asan.so#$OUT/libfdb_asan.so#; s#'/root/repo'#'$ROOT'#" "$ROOT/tools/asan_arrow_run.py" > "$OUT/run.py"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -I"$ROOT/include" -I"$ROOT/frostdb_amd/csrc" \
    "$ROOT/tools/asan_arrow_shim.cpp" "$ROOT/frostdb_amd/csrc/fdb_arrow.cpp" -o "$OUT/libfdb_asan.so"
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python "$OUT/run.py"
# … and the same instrumented library against records with detectable defects (tools/arrow_fuzz.py): error codes, no reports
FDB_FUZZ_LIB="$OUT/libfdb_asan.so" LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python "$ROOT/tools/arrow_fuzz.py" "${1:-2000}" "${2:-1}"
