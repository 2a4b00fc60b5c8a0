"""Where cfg 5's hash scan spends its time (tuning aid): kernel time per launch with parts of the kernel compiled out,
and the phases of Finish. usage: cfg5_ablate.py [rows] [groups]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import DynCol, Col, Sum

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
chunk = 8_000_000
synth.cfg5_chunk(0, 0, 8, n_groups=groups)
with ThreadPoolExecutor(8) as ex:
    batches = list(ex.map(lambda i: synth.cfg5_chunk(0, i, min(chunk, rows - i * chunk), n_groups=groups), range((rows + chunk - 1) // chunk)))
resident = [pp.ResidentBatch(b) for b in batches]
aggs, G = [Sum(Col("value"))], [DynCol("labels")]
modes = [(0, "specialised kernel (fdb_hash_kernel)"), (2, "specialised: probe only (no count/sum atomics)"), (1, "specialised: stream + fingerprint only"),
         (4 << 5, "interpreting kernel, full"), ((4 << 5) | 2, "interpreting: probe only (no count/sum atomics)"), ((4 << 5) | 1, "interpreting: stream + fingerprint only")]
for g in (512, 1024, 2048, 4096):
    modes.append(((g << 8) | 1, f"specialised: stream + fingerprint only, grid {g}"))
for ablate, what in modes:
    for it in range(2):
        plan = pp.HashAggregatePlan(None, aggs, G)
        plan.set_timing(True)
        # low 4 bits: ablate; bits 5-7: variant mode (4 = interpreting kernels only); bits 8+: grid override
        plan.set_tuning(0, ((ablate & 15) << 20) | (((ablate >> 5) & 7) << 25) | (ablate >> 8))
        t0 = time.perf_counter()
        plan.CallbackResident(resident)
        t1 = time.perf_counter()
        n = plan.num_groups()
        t2 = time.perf_counter()
        st = plan.stats()
        kern = plan.last_kernel()
        if ablate == 0:
            out = plan.Finish()
        t3 = time.perf_counter()
        plan.Close()
    print(f"{kern:18s} {what:50s} kernel {st['kernel_ms'] / st['launches']:.3f} ms/launch x {st['launches']}  {st['algorithmic_bytes'] / st['kernel_ms'] / 1e6:8.1f} GB/s  "
          f"push {1e3 * (t1 - t0):.1f} ms  finish {1e3 * (t3 - t2):.1f} ms  groups {n}", flush=True)
# steady state: a second pass over the same rows finds every group (no inserts, no key-store writes)
for mode, what in [(0, "specialised"), (2 << 20, "specialised, probe only"), (1 << 20, "specialised, stream only"), (4 << 25, "interpreting")]:
    plan = pp.HashAggregatePlan(None, aggs, G)
    plan.set_timing(True)
    plan.set_tuning(0, mode)
    plan.set_tuning(0, mode & ~(15 << 20))  # first pass always builds the table
    plan.CallbackResident(resident)
    plan.num_groups()
    st0 = plan.stats()
    plan.set_tuning(0, mode)
    plan.CallbackResident(resident)
    plan.num_groups()
    st1 = plan.stats()
    n = st1["launches"] - st0["launches"]
    ms = st1["kernel_ms"] - st0["kernel_ms"]
    print(f"{plan.last_kernel():18s} {what:14s} second pass (no inserts): kernel {ms / n:.3f} ms/launch  {(st1['algorithmic_bytes'] - st0['algorithmic_bytes']) / ms / 1e6:8.1f} GB/s", flush=True)
    plan.Close()
os.environ["FDB_PROFILE"] = "1"
plan = pp.HashAggregatePlan(None, aggs, G)
plan.CallbackResident(resident)
out = plan.Finish()
print("rows out", out.num_rows, "cols", out.num_columns)
plan.Close()
