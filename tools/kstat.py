"""Per-kernel durations out of a rocprofv3 --kernel-trace run's *_results.db: python tools/kstat.py <dir> [name filter]"""
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"
    for r in db.execute(q):
        if len(sys.argv) < 3 or sys.argv[2] in r[0]:
            print(f"  {r[0][:70]:70s} n={r[1]:4d} avg {r[2] / 1e3:9.1f} us  min {r[3] / 1e3:9.1f} us")
