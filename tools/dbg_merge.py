import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from frostdb_amd import physicalplan as pp
from frostdb_amd.logicalplan import Col, Count, DynCol, Max, Min, Sum
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from tests.test_gpu_parity import many_label_batch
rng = np.random.default_rng(406)
aggs = [Sum(Col("value")), Count(Col("value")), Min(Col("floatvalue")), Max(Col("value"))]
b1 = many_label_batch(rng, 20_000, 2, 5)
b2 = many_label_batch(rng, 20_000, 12, 3, n_groups=2500)
b3 = many_label_batch(rng, 15_000, 12, 3, n_groups=2500)
p1 = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
p2 = pp.HashAggregatePlan(None, aggs, [DynCol("labels")])
p1.Callback(b1); p1.Callback(b2); p2.Callback(b3)
print("merging", flush=True)
p1.Merge(p2)
print("merged", p1.num_groups())
