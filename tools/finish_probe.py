"""Tuning aid: cfg 5's table built once per variant, Finish phases printed by FDB_PROFILE (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FDB_PROFILE"] = "1"
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col, DynCol, Sum, Count
n_rec, per = int(sys.argv[1]), int(sys.argv[2])
variants = sys.argv[3].split(",")
keep = [pp.ResidentBatch(synth.cfg5_chunk(0, i, per)) for i in range(n_rec)]
for v in variants + variants:
    os.environ["FDB_FINISH_VARIANT"] = v
    plan = pp.HashAggregatePlan(None, [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")])
    plan.CallbackResident(keep)
    sys.stderr.write("== variant %s\n" % v); sys.stderr.flush()
    t0 = time.perf_counter()
    out = plan.Finish()
    t1 = time.perf_counter()
    sys.stderr.write("== variant %s finish %.2f ms rows %d\n" % (v, (t1 - t0) * 1e3, out.num_rows)); sys.stderr.flush()
    plan.Close(); del out
