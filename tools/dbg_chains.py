"""Debug aid: N chains pushing host records concurrently; every chain's own result and the merged one against numpy."""
import math, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import to_desc
import bench
chains, rec_rows, per_chain = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
threads = len(sys.argv) > 4 and sys.argv[4] == "threads"
filt, aggs, groups, _ = bench.query(2)
desc = to_desc(filt, aggs, groups)
src = synth.prometheus_chunk(0, 0, chains * per_chain)
names = synth.PATHS + [None]
def expect(b):
    s, c = bench.expected_cfg2(b)
    return {names[i]: s[i] for i in range(len(names)) if c[i]}
recs = [[src.slice(c * per_chain + o, min(rec_rows, per_chain - o)) for o in range(0, per_chain, rec_rows)] for c in range(chains)]
exported = [[pp.ExportedBatch(r) for r in rs] for rs in recs]
import time
for rep in range(int(os.environ.get("REPEAT", "1"))):
    plans = [pp.HashAggregatePlan(filt, aggs, groups, desc=desc) for _ in range(chains)]
    def work(c):
        plans[c].CallbackExportedMany(exported[c])
        plans[c].last_kernel()
    t0 = time.perf_counter()
    if threads:
        ts = [threading.Thread(target=work, args=(c,)) for c in range(chains)]
        [t.start() for t in ts]; [t.join() for t in ts]
    else:
        for c in range(chains): work(c)
    for p in plans: p.num_groups()
    dt = time.perf_counter() - t0
    print(f"pass {rep}: {dt * 1e3:.2f} ms = {chains * per_chain / dt / 1e9:.3f} G rows/s, {dt * 1e6 / len(exported[0]):.1f} us per record per chain")
    if rep + 1 < int(os.environ.get("REPEAT", "1")):
        for p in plans: p.Close()
bad = 0
for c in range(chains):
    st = plans[c].partial_keys()
    keys = st.column(0).to_pylist()
    vals = np.zeros(len(keys), dtype=np.float64)
    plans[c].partial_state_into(0, vals.ctypes.data, vals.nbytes)
    want = expect(src.slice(c * per_chain, per_chain))
    got = dict(zip(keys, vals.tolist()))
    nb = sum(1 for k in want if not math.isclose(got.get(k, float("nan")), want[k], rel_tol=1e-9))
    if nb: print("chain", c, "bad groups", nb, "of", len(want), "got groups", len(got))
    bad += nb
for p in plans[1:]:
    plans[0].Merge(p)
res = plans[0].Finish()
got = dict(zip(res.column(0).to_pylist(), res.column(1).to_pylist()))
want = expect(src)
nb = sum(1 for k in want if not math.isclose(got.get(k, float("nan")), want[k], rel_tol=1e-9))
print(f"chains {chains} rec_rows {rec_rows} threads {threads}: per-chain bad {bad}, merged bad {nb} of {len(want)}")
