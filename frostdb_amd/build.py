"""Builds libfrostdb_amd.so (HIP kernels + host C++ + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs on the CPU-only build container too. The .so is git-ignored
but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfrostdb_amd.so")
SOURCES = ["fdb_kernels.hip", "fdb_merge.hip", "fdb_sort.hip", "fdb_arrow.cpp", "fdb_context.cpp", "fdb_plan.cpp", "fdb_hash.cpp", "fdb_jit.cpp", "fdb_dynamic.cpp", "fdb_comm.cpp", "fdb_parquet.cpp", "fdb_widen.cc", "fdb_regex.cpp", "fdb_capi.cpp"]
HEADERS = ["fdb_kernels.h", "fdb_arrow.h", "fdb_context.h", "fdb_plan.h", "fdb_plan_internal.h", "fdb_jit.h", "fdb_comm.h", "fdb_dynamic.h", "fdb_regex.h", "fdb_hostpool.h", "fdb_unicode_tables.inc", "exports.map", "../../include/frostdb_amd.h", "../../include/arrow_c_data.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-result",
         "-fvisibility=hidden", "-fvisibility-inlines-hidden"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the argument-block header, embedded as a string for the run-time (hiprtc) compiled plan kernels
    with open(os.path.join(CSRC, "fdb_kernels.h")) as f:
        text = f.read()
    assert ')FDBH"' not in text
    inc = os.path.join(CSRC, "fdb_kernels_h.inc")
    new = 'R"FDBH(' + text + ')FDBH"\n'
    if not os.path.exists(inc) or open(inc).read() != new:
        with open(inc, "w") as f:
            f.write(new)
    # per-object staleness (a header change rebuilds everything), translation units compiled side by side
    from concurrent.futures import ThreadPoolExecutor
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + ["fdb_kernels_h.inc"])
    flags_sig = " ".join(FLAGS)
    sig_path = os.path.join(CSRC, ".build_flags")
    same_flags = os.path.exists(sig_path) and open(sig_path).read() == flags_sig
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        fresh = (not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) >= hdr_t
                 and os.path.getmtime(obj) >= os.path.getmtime(os.path.join(CSRC, src)))
        if not fresh:
            jobs.append([hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs) or 1, os.cpu_count() or 4))) as ex:
        list(ex.map(run, jobs))
    with open(sig_path, "w") as f:
        f.write(flags_sig)
    # The dynamic symbol surface is include/frostdb_amd.h's FDB_API prototypes and nothing else: hidden visibility keeps the library's own
    # C++ out, the version script also drops the weak template instantiations libstdc++'s headers force to default visibility.
    vs = os.path.join(CSRC, "exports.map")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", LIB] + objs + ["-lhiprtc", "-ldl", "-lpthread", "-lz"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
