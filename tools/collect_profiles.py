"""Copies what a round's profile run (tools/profile_round.sh <tag>, under gpurun_out/<tag>/) produced into profiles/ — the per-kernel
statistics, the condensed summaries, the bench lines — and turns the FETCH_SIZE / WRITE_SIZE counter passes into the
<tag>_<name>_traffic.json files bench.py reads (traffic_for): mean per launch of the named kernel, FETCH_SIZE KiB × 1024 × 2 (gfx950
tallies 128-byte requests as 64 bytes, MI355X_MICROARCH.md HBM section), WRITE_SIZE KiB × 1024.   python tools/collect_profiles.py round4"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "round6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
# name → (kernel as rocprof names it, kernel as bench.py names it, rows per launch group, the command)
PASSES = {
    "cfg1B": ("fdb_plan_kernel", "fdb_plan_kernel", 1_000_000_000, "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-oracle-parity"),
    "cfg3": ("fdb_plan_kernel", "fdb_plan_kernel", 100_000_000, "python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-oracle-parity"),
    "cfg5": ("fdb_hash_kernel", "fdb_hash_kernel", 100_000_000, "python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity"),
    "cfg5_sorted": ("fdb_hash_kernel", "fdb_hash_kernel(runs)", 100_000_000, "python bench.py --config 5 --cfg5-sorted --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity"),
    "cfg5_sorted_wide": ("fdb_hash_kernel", "fdb_hash_kernel(runs, medium)", 100_000_000, "python bench.py --config 5 --cfg5-sorted --cfg5-wide-dicts --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity"),
    "cfg5_sorted_widerec": ("fdb_hash_kernel", "fdb_hash_kernel(runs, wide)", 100_000_000, "env FDB_RUNS_WIDE=1 python bench.py --config 5 --cfg5-sorted --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity"),
    "cfg2_sorted": ("fdb_plan_kernel", "fdb_plan_kernel", 100_000_000, "python bench.py --config 2 --cfg2-sorted --steps 10 --warmup 2 --no-cpu-baseline --no-oracle-parity"),
    "cfg5_1B": ("fdb_hash_kernel", "fdb_hash_kernel", 1_000_000_000, "python bench.py --rows 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-oracle-parity --only-other cfg5_1B"),
    # filter(): a step is several kernels — the traffic of a step is the sum of their per-launch means (each runs once per step)
    "select": (["fdb_select_kernel", "compact_multi_kernel", "zero_regions_kernel"], "fdb_select_kernel + compact_multi_kernel", 100_000_000,
               "python bench.py --rows 100000000 --steps 5 --warmup 1 --no-cpu-baseline --only-other select --no-oracle-parity"),
}


def counter_mean(name, which, kernel):
    vals = []
    for f in glob.glob(os.path.join(src, name, which, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name", "").startswith(which.upper()):
                    vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


for f in glob.glob(os.path.join(src, "*_line.json")) + glob.glob(os.path.join(src, "*.summary.txt")) + glob.glob(os.path.join(src, "step_probe_*.txt")):
    shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
for d in glob.glob(os.path.join(src, "*", "kt")):
    name = os.path.basename(os.path.dirname(d))
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
for name, (rk, bk, rows, cmd) in PASSES.items():
    if isinstance(rk, list):
        parts = [(counter_mean(name, "fetch", k), counter_mean(name, "write", k)) for k in rk]
        parts = [(f, w) for f, w in parts if f[0] is not None and w[0] is not None]
        if not parts:
            continue
        fetch, write = sum(f[0] for f, _ in parts), sum(w[0] for _, w in parts)
        nf = min(f[1] for f, _ in parts)
        rk = " + ".join(rk)
    else:
        fetch, nf = counter_mean(name, "fetch", rk)
        write, nw = counter_mean(name, "write", rk)
        if fetch is None or write is None:
            continue
    out = {"kernel": bk, "rows": rows, "fetch_bytes_per_launch": fetch * 1024 * 2, "write_bytes_per_launch": write * 1024,
           "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate, no tracing; tools/profile_round.sh {tag}) of `{cmd}`, mean of {nf} launches of {rk}; "
                     "FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE KiB x 1024",
           "per_average_launch": True, "fetch_bytes_per_launch_uncorrected": fetch * 1024}
    with open(os.path.join(dst, f"{tag}_{name}_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(name, out["fetch_bytes_per_launch"] / 1e9, "GB fetched,", out["write_bytes_per_launch"] / 1e9, "GB written per launch")
print("copied into", dst)
