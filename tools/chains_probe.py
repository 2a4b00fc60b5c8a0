"""Where do concurrent chains pushing small HOST records serialise? N threads, one plan each, `records` records of
`rows` rows per chain through fdb_plan_push_many; wall time of the push phase alone (barrier → last chain's settle), then the merges and Finish
apart. FDB_PROFILE_PUSH=1 adds the per-record phases of every plan (printed when it closes).
usage: chains_probe.py [rows per record] [records per chain]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col, Sum

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
per_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
big = synth.prometheus_chunk(0, 0, rows * per_chain)
recs = [big.slice(i * rows, rows) for i in range(per_chain)]
filt, aggs, groups = Col("labels.code") == "200", [Sum(Col("value"))], [Col("labels.path")]
for chains in [int(x) for x in os.environ.get("CHAINS", "1,2,4,8,16,32,64").split(",")]:
    exported = [pp.PreparedRun([pp.ExportedBatch(r) for r in recs]) for _ in range(chains)]
    best = None
    for rep in range(3):
        plans = [pp.HashAggregatePlan(filt, aggs, groups) for _ in range(chains)]
        bar = threading.Barrier(chains + 1)
        def work(c):
            bar.wait()
            plans[c].CallbackPrepared(exported[c])
            plans[c].last_kernel()
        ts = [threading.Thread(target=work, args=(c,)) for c in range(chains)]
        for t in ts: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in ts: t.join()
        t1 = time.perf_counter()
        for p in plans[1:]: plans[0].Merge(p)
        t2 = time.perf_counter()
        out = plans[0].Finish()
        t3 = time.perf_counter()
        if rep < 2:
            os.environ.pop("FDB_PROFILE_PUSH_KEEP", None)
        for p in plans: p.Close()
        cur = (t1 - t0, t2 - t1, t3 - t2)
        if best is None or cur[0] < best[0]: best = cur
    n = chains * per_chain
    print(f"{chains:3d} chains x {per_chain} records of {rows} rows: push {best[0] * 1e3:8.2f} ms = {best[0] / per_chain * 1e6:7.2f} us per record and chain, "
          f"{n * rows / best[0] / 1e9:6.3f} G rows/s; {chains - 1} merges {best[1] * 1e3:7.2f} ms; finish {best[2] * 1e3:6.2f} ms", flush=True)
