"""The run-time generated kernels compile for gfx950 — checked here, without a GPU: on the GPU box hiprtc compiles them at the first query
of a shape, and a source that does not compile would silently put that shape on the interpreting kernels (or fail the one-pass filter,
the run kernel, the deterministic mode, which have no interpreting twin). tools/jit_dump.cpp prints the source the generators in
fdb_jit.cpp produce for representative shapes; hipcc compiles each for the device only (the same options hiprtc gets)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SHAPES = [
    ("plan_lds", ["plan", "lds"], "fdb_plan_kernel"),
    ("plan_wave_tables", ["plan", "wave"], "fdb_plan_kernel"),       # fdb_plan_set_deterministic
    ("plan_reg_slots", ["plan", "reg"], "fdb_plan_kernel"),
    ("plan_cache", ["plan", "cache"], "fdb_plan_kernel"),
    ("plan_two_phase", ["plan", "two"], "fdb_plan_kernel"),
    ("flags", ["flags"], "fdb_flags_kernel"),
    ("select_value", ["select", "1"], "fdb_select_kernel"),          # filter() in one pass: one 8-byte column fused
    ("select_value_and_dict", ["select"], "fdb_select_kernel"),      # … next to an unfused dictionary leaf with a LUT in LDS
    ("hash_32_columns", ["32"], "fdb_hash_kernel"),                  # cfg 5
    ("hash_int64_key", ["8", "1"], "fdb_hash_kernel"),
    ("runs_32_columns", ["32", "2"], "fdb_hash_kernel"),             # the table-free OrderedAggregate's run kernel
    ("runs_wide_32_columns", ["32", "3"], "fdb_hash_kernel"),        # … writing wide records (any cardinality: tuples from re-loaded columns)
    ("runs_wide_int64_key", ["3", "4"], "fdb_hash_kernel"),          # … with an int64 key column
    ("runs_medium_32_columns", ["32", "5"], "fdb_hash_kernel"),      # … writing medium records (two bytes per key id, ids kept in registers)
]


@pytest.fixture(scope="module")
def jit_dump(tmp_path_factory):
    from frostdb_amd import build
    lib = build.build()
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this host")
    out = tmp_path_factory.mktemp("jit")
    exe = str(out / "jit_dump")
    # the generators are library internals (hidden symbols: the .so exports the C ABI only), so the tool links the library's OBJECTS
    import glob
    objs = sorted(glob.glob(os.path.join(os.path.dirname(lib), "csrc", "*.o")))
    assert objs, "frostdb_amd/csrc/*.o missing: build.build() keeps them next to the sources"
    subprocess.check_call(["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "frostdb_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "tools", "jit_dump.cpp")] + objs +
                          ["-L/opt/rocm/lib", "-lamdhip64", "-lhiprtc", "-ldl", "-lpthread", "-lz", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe, out


def test_generated_kernels_compile_for_gfx950(jit_dump):
    exe, out = jit_dump

    def one(shape):
        name, args, kernel = shape
        src = subprocess.run([exe] + args, check=True, capture_output=True, text=True).stdout
        assert ("void " + kernel + "(") in src, name
        path = str(out / (name + ".hip"))
        with open(path, "w") as f:
            f.write(src)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-DFDB_DEVICE_ONLY=1", "-include", "hip/hip_runtime.h",
                            "-I", os.path.join(ROOT, "frostdb_amd", "csrc"), "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", path + ".o"],
                           capture_output=True, text=True)
        scratch = [ln for ln in r.stderr.splitlines() if "ScratchSize" in ln]
        return name, r.returncode, r.stderr if r.returncode else "", scratch

    with ThreadPoolExecutor(max_workers=min(len(SHAPES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(one, SHAPES))
    failed = [(n, err[-2000:]) for n, rc, err, _ in results if rc != 0]
    assert not failed, failed
    # none of these shapes may spill registers to scratch memory (a spill in a streaming kernel is a performance bug, not a detail)
    spilled = [(n, s) for n, _, _, s in results if s and not all("ScratchSize [bytes/lane]: 0" in ln for ln in s)]
    assert not spilled, spilled
