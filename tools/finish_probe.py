"""Tuning aid: cfg 5's table built a few times, Finish phases printed by FDB_PROFILE (run on the GPU box):
    python tools/finish_probe.py <records> <rows per record> [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FDB_PROFILE"] = "1"
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col, DynCol, Sum, Count
n_rec, per = int(sys.argv[1]), int(sys.argv[2])
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 2
keep = [pp.ResidentBatch(synth.cfg5_chunk(0, i, per)) for i in range(n_rec)]
for rep in range(repeats):
    plan = pp.HashAggregatePlan(None, [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")])
    plan.CallbackResident(keep)
    sys.stderr.write("== pass %d\n" % rep); sys.stderr.flush()
    t0 = time.perf_counter()
    out = plan.Finish()
    t1 = time.perf_counter()
    sys.stderr.write("== pass %d finish %.2f ms rows %d\n" % (rep, (t1 - t0) * 1e3, out.num_rows)); sys.stderr.flush()
    plan.Close(); del out
