"""Condenses the rocprofv3 CSVs produced by tools/profile.sh into a short per-kernel text summary."""
import csv, glob, os, sys, collections

root = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r


# kernel stats
print("# kernel trace --stats (durations in ns)")
for f, r in rows("kt/**/*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
# counters: average per dispatch per kernel
print("# PMC counters: mean per dispatch, by kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f, r in rows("**/*counter_collection.csv"):
    name = r.get("Kernel_Name", "?")
    agg[name][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
for k, d in agg.items():
    if "scan" not in k and "reduce" not in k:
        continue
    print(k[:90])
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
