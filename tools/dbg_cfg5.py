"""Debug aid: cfg 5-shaped scan at a reduced size, count / sum checks (run on the GPU box)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col, DynCol, Sum, Count
n_groups, per, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
recs = [synth.cfg5_chunk(0, i, per, n_groups=n_groups) for i in range(n)]
gids = np.concatenate([synth.cfg5_group_ids(0, i, per, n_groups=n_groups) for i in range(n)])
keep = [pp.ResidentBatch(r) for r in recs]
plan = pp.HashAggregatePlan(None, [Sum(Col("value")), Count(Col("value"))], [DynCol("labels")])
plan.CallbackResident(keep)
out = plan.Finish()
plan.Close()
exp = len(np.unique(gids))
print("groups out", out.num_rows, "expected", exp, "count sum", int(np.sum(out.column("count(value)").to_numpy())), "rows", per * n)
g = synth.cfg5_decode_group_ids(out)
print("distinct decoded ids", len(np.unique(g)))
