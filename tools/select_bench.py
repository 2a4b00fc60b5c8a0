#!/usr/bin/env python
"""Measures the device-resident filter() path (fdb_plan_filter_batch: predicate + one-pass stream compaction of every column)
and the selection-vector path (fdb_plan_select_batch) on the cfg 2 schema: 100 M rows resident in HBM (4 records of 25 M),
`value > T` with T chosen for the requested selectivity. Prints one JSON line: rows/s, the compaction kernel's achieved GB/s
against its ALGORITHMIC bytes (filter columns once + every selected value and validity bit read once and written once — the
ideal gather; at 50 % selectivity a sector-granular memory has to read every input sector, so real traffic is higher: see
`min_traffic_frac`) and the whole step (kernel + pack of validity bitmaps + the one host round trip per record).

Run on the GPU box: python tools/select_bench.py [--selectivity 0.5] [--rows 100000000] [--steps 10]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--selectivity", type=float, default=0.5)
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--batch-rows", type=int, default=25_000_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--mode", default="filter", choices=["filter", "select"])
args = ap.parse_args()

import numpy as np
import torch
from frostdb_amd import build as fb
fb.build()
from frostdb_amd import physicalplan as pp, synth
from frostdb_amd.logicalplan import Col

thr = (1.0 - args.selectivity) * 1000.0
filt = Col("value") > float(thr)
n_chunks = (args.rows + args.batch_rows - 1) // args.batch_rows
recs, exp_sel, exp_sum = [], 0, 0.0
for i in range(n_chunks):
    b = synth.prometheus_chunk(0, i, min(args.batch_rows, args.rows - i * args.batch_rows), row_base=i * args.batch_rows)
    v = b.column(b.schema.get_field_index("value")).to_numpy()
    m = v > thr
    exp_sel += int(m.sum()); exp_sum += float(v[m].sum())
    recs.append(pp.ResidentBatch(b))
    del b
idx_buf = torch.empty(args.batch_rows, dtype=torch.int32, device="cuda:0") if args.mode == "select" else None


def step(timing=False):
    plan = pp.HashAggregatePlan(filt)
    if timing:
        plan.set_timing(True)
    n, outs = 0, []
    for rb in recs:
        if args.mode == "select":
            n += plan.SelectResident(rb, idx_buf.data_ptr(), args.batch_rows)
        else:
            o = plan.FilterResident(rb)
            n += o.num_rows
            outs.append(o)
    st = plan.stats() if timing else None
    plan.Close()
    return n, outs, st


n, outs, _ = step()
assert n == exp_sel, (n, exp_sel)
if outs:
    got = sum(float(o.to_arrow().column("value").to_numpy().sum()) for o in outs[:1])
    first = recs[0].to_arrow().column("value").to_numpy()
    assert abs(got - float(first[first > thr].sum())) <= 1e-9 * abs(got)
for o in outs:
    o.close()
for _ in range(args.warmup):
    _, outs, _ = step()
    for o in outs:
        o.close()
torch.cuda.synchronize()
t0 = time.perf_counter()
k_ms = k_bytes = k_launches = 0
for _ in range(args.steps):
    _, outs, st = step(timing=True)
    k_ms += st["kernel_ms"]; k_bytes += st["algorithmic_bytes"]; k_launches += st["launches"]
    for o in outs:
        o.close()
torch.cuda.synchronize()
el = time.perf_counter() - t0
in_bytes = sum(r.device_bytes for r in recs)
sel = exp_sel / args.rows
row_in = 24.25  # code 4 + path 4 + timestamp 8 + value 8 + two validity bits
min_traffic = args.rows * (row_in + sel * row_in) if args.mode == "filter" else args.rows * (8 + 4 * sel)
ach = k_bytes / (k_ms * 1e-3) / 1e9
print(json.dumps({
    "metric": f"rows/sec {args.mode} on resident Prometheus records (cfg 2 schema), value > {thr:g}", "value": args.rows * args.steps / el, "unit": "rows/s",
    "steps": args.steps, "ms_per_step": el / args.steps * 1e3, "rows": args.rows, "records": n_chunks, "selectivity": sel,
    "roofline": {"bound": "hbm", "kernel": "filter_flags_kernel + scan_counts_kernel + compact_col_kernel<4|8> per column (event-timed together)", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                 "avg_launch_ms": k_ms / max(k_launches, 1), "algorithmic_bytes_per_launch": k_bytes / max(k_launches, 1),
                 "algorithmic_bytes_per_row": k_bytes / (args.rows * args.steps),
                 "min_traffic_bytes_per_row": min_traffic / args.rows,
                 "min_traffic_frac": (min_traffic * args.steps / (k_ms * 1e-3) / 1e9) / 8000.0,
                 "whole_step_frac": (k_bytes / args.steps) / (el / args.steps) / 1e9 / 8000.0},
    "hbm_resident_bytes": in_bytes}))
