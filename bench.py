#!/usr/bin/env python
"""bench.py — rows/sec of the fused filter + group-by aggregate over synthetic Prometheus Arrow data.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1: one rank per GPU under
``python -m torch.distributed.run``; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env). A *step* is one full
pass of the hot path over this rank's resident shard: create the operator chain (fdb_plan_create), push every
HBM-resident record through the fused HIP kernel (fdb_plan_push_batch), and produce the final record
(fdb_plan_finish at N = 1; at N > 1 the per-GPU partial tables are merged with RCCL all-reduces and rank 0
materialises the record). Inputs are resident in HBM before the timed region; results are checked.

Rank 0 prints ONE JSON line with the metric, the HBM roofline of the scan kernel (hipEvent-timed on the
plan's own stream over the timed steps) and the CPU baseline (the oracle's restatement of the reference's
algorithm, timed on the host cores of this box on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ≈6300 GB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5], help="BASELINE.json config (2: bench line; 3: multi-predicate)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: 100M at N=1, 125M per GPU at N>1)")
    ap.add_argument("--batch-rows", type=int, default=25_000_000, help="rows per resident record (part)")
    ap.add_argument("--groups", type=int, default=10_000_000, help="cfg 5: distinct groups")
    ap.add_argument("--rows-per-thread", type=int, default=0, help="0: slot kernel (default); 4/8: sequential kernel")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--host-records", action="store_true",
                    help="secondary measurement: push HOST Arrow records (fdb_plan_push: PCIe copy + scan per record) instead of HBM-resident parts; never the headline value")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant: 0 default (run-time specialised, 512 threads), 2: 256 threads, 3: 1024 threads, 4: interpreting kernel only")
    ap.add_argument("--per-record-launch", action="store_true", help="one kernel launch per resident record instead of one per scan")
    ap.add_argument("--force-merge", action="store_true", help="run the RCCL merge path even with one rank (functional check on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-seconds", type=float, default=3.0)
    ap.add_argument("--sweep", action="store_true", help="kernel geometry sweep first (tuning aid; table on stderr)")
    return ap.parse_args()


def query(config):
    from frostdb_amd.logicalplan import And, Col, Count, Max, Min, Or, Sum
    if config == 2:
        return (Col("labels.code") == "200", [Sum(Col("value"))], [Col("labels.path")],
                "labels.code=='200' + SUM(value) GROUP BY labels.path")
    if config == 5:
        from frostdb_amd.logicalplan import DynCol
        return (None, [Sum(Col("value"))], [DynCol("labels")], "SUM(value) GROUP BY all 32 labels.* columns (10 M distinct groups)")
    f = And(Or(Col("labels.code") == "200", Col("labels.code") == "500"), Col("labels.method") == "GET",
            Col("labels.instance") != None)  # noqa: E711
    return (f, [Count(Col("value")), Min(Col("timestamp")), Max(Col("timestamp")), Sum(Col("value"))], [Col("labels.path")],
            "(code=='200' OR code=='500') AND method=='GET' AND instance!=NULL + COUNT/MIN/MAX/SUM GROUP BY labels.path")


def expected_cfg2(batch):
    """numpy restatement of cfg 2 on one record (property check of the timed path; not the oracle)."""
    import numpy as np
    code, path = batch.column(0), batch.column(1)
    value = batch.column(batch.schema.get_field_index("value")).to_numpy()
    cidx = code.indices.fill_null(len(code.dictionary)).to_numpy(zero_copy_only=False)
    code200 = [i for i, v in enumerate(code.dictionary.to_pylist()) if v == b"200"][0]
    sel = cidx == code200
    pidx = path.indices.fill_null(len(path.dictionary)).to_numpy(zero_copy_only=False).astype(np.int64)
    sums = np.bincount(pidx[sel], weights=value[sel], minlength=len(path.dictionary) + 1)
    cnts = np.bincount(pidx[sel], minlength=len(path.dictionary) + 1)
    return sums, cnts


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_merge:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from frostdb_amd import build as fb
    if rank == 0:
        fb.build()
    if world > 1:
        dist.barrier()
    from frostdb_amd import physicalplan as pp
    from frostdb_amd import synth
    from frostdb_amd.distributed import layout_probe, merge_plan, merge_plan_alltoall

    rows = args.rows or (100_000_000 if world == 1 else 125_000_000)
    cfg3 = args.config == 3
    filt, aggs, groups, qdesc = query(args.config)

    # ---- synthetic shard, generated on the host and made resident in HBM (outside the timed region) -----------
    t_gen = time.time()
    n_chunks = (rows + args.batch_rows - 1) // args.batch_rows
    sizes = [min(args.batch_rows, rows - i * args.batch_rows) for i in range(n_chunks)]

    def gen(i):
        if args.config == 5:
            return synth.cfg5_chunk(rank, i, sizes[i], n_groups=args.groups)
        return synth.prometheus_chunk(rank, i, sizes[i], row_base=i * args.batch_rows, cfg3=cfg3)

    if args.config == 5:
        synth.cfg5_chunk(rank, 0, 8, n_groups=args.groups)  # builds the per-group digit tables once, before the thread pool
    resident = []
    host_batches = []
    exp_sum = exp_cnt = None
    sample_for_cpu = []
    with ThreadPoolExecutor(max_workers=min(8, n_chunks)) as ex:
        for i, b in enumerate(ex.map(gen, range(n_chunks))):
            if args.config == 2:
                s, c = expected_cfg2(b)
                exp_sum = s if exp_sum is None else exp_sum + s
                exp_cnt = c if exp_cnt is None else exp_cnt + c
            if args.host_records:
                host_batches.append(b)
            else:
                resident.append(pp.ResidentBatch(b, device=local_rank))
            if rank == 0 and i == 0:
                sample_for_cpu.append(b)
    t_gen = time.time() - t_gen
    hbm_bytes = sum(r.device_bytes for r in resident)

    from frostdb_amd.logicalplan import to_desc
    desc = to_desc(filt, aggs, groups)  # the query is planned once; every step instantiates and runs a fresh operator chain

    def step(timing=False, tuning=None):
        plan = pp.HashAggregatePlan(filt, aggs, groups, device=local_rank, desc=desc)
        if timing:
            plan.set_timing(True)
        if tuning:
            plan.set_tuning(*tuning)
        else:
            plan.set_tuning(args.rows_per_thread, args.grid | (args.variant << 25))
        if args.host_records:
            for hb in host_batches:
                plan.Callback(hb)
        elif args.per_record_launch:
            for rb in resident:
                plan.Callback(rb)
        else:
            plan.CallbackResident(resident)
        if (world > 1 or args.force_merge) and args.config == 5:
            # high cardinality: hash-partitioned all-to-all; every rank finishes its own shard of the groups
            shard = merge_plan_alltoall(plan, device=torch.device("cuda", local_rank))
            try:
                out = shard.Finish()
            finally:
                shard.Close()
        elif world > 1 or args.force_merge:
            probe = layout_probe(plan, torch.device("cuda", local_rank))  # overlaps with the scan kernel
            out = merge_plan(plan, probe=probe)
        else:
            out = plan.Finish()
        st = plan.stats() if timing else None
        if st is not None:
            st["kernel"] = plan.last_kernel()
        plan.Close()
        return out, st

    # ---- correctness of what is being timed -------------------------------------------------------------------
    out, _ = step()
    if args.config == 2 and world == 1:
        col0 = out.column(0).dictionary_decode() if hasattr(out.column(0), "dictionary_decode") else out.column(0)
        got = {k: v for k, v in zip(col0.to_pylist(), out.column(1).to_pylist())}
        paths = synth.PATHS + [None]
        for i, p in enumerate(paths):
            if exp_cnt[i] == 0:
                assert p not in got, p
            else:
                assert math.isclose(got[p], exp_sum[i], rel_tol=1e-9), (p, got[p], exp_sum[i])
        assert len(got) == int((exp_cnt > 0).sum())

    if args.config == 5:
        # every group of the synthetic table shows up (rows ≫ groups): the scan / merge must find exactly that many
        n_out = out.num_rows
        if world > 1:
            t = torch.tensor([n_out], dtype=torch.int64, device="cuda")
            dist.all_reduce(t)  # shards of the all-to-all merge are disjoint
            n_out = int(t.item())
        if rank == 0:
            print(f"# cfg5: {n_out} groups in the result ({args.groups} distinct label tuples generated, {rows * world} rows)", file=sys.stderr)
        assert n_out <= args.groups and (rows * world < 5 * args.groups or n_out > 0.99 * args.groups), n_out

    if args.sweep:
        if rank == 0:
            print(f"# sweep: rows={rows} batch_rows={args.batch_rows} cfg={args.config}", file=sys.stderr)
        for rpt, grid in [(4, 512), (0, 0), (0, 1024), (0, 1024 | (1 << 24)), (0, 2048)]:
            if True:
                for _ in range(2):
                    step(tuning=(rpt, grid))
                tot_ms, tot_b, n = 0.0, 0, 0
                for _ in range(5):
                    _, st = step(timing=True, tuning=(rpt, grid))
                    tot_ms += st["kernel_ms"]; tot_b += st["algorithmic_bytes"]; n += st["launches"]
                if rank == 0:
                    print(f"rpt={rpt} grid={grid & 0xFFFFF:5d} ablate={(grid >> 20) & 15:2d} atomic_flush={grid >> 24}  kernel {tot_ms / n:8.4f} ms/launch  {tot_b / tot_ms / 1e6:8.1f} GB/s  ({tot_b / n / 1e6:.1f} MB/launch)",
                          file=sys.stderr)

    # ---- timed region ----------------------------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_ms, k_bytes, k_launches = 0.0, 0, 0
    for _ in range(args.steps):
        _, st = step(timing=True)
        k_ms += st["kernel_ms"]; k_bytes += st["algorithmic_bytes"]; k_launches += st["launches"]
        kernel_name = st["kernel"]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_rows = rows * world * args.steps
    value = total_rows / elapsed

    # ---- CPU baseline (rank 0, N = 1 only): the oracle's restatement on this box's host cores, bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sample_for_cpu[0], filt, aggs, groups, args.cpu_sample_seconds)

    ceiling = None
    if rank == 0 and world == 1:
        ceiling = pp.read_ceiling(local_rank, 2 << 30, 5)  # plain read kernel on this box, this run

    # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside the timed process)
    traffic, traffic_src = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"round1_cfg{args.config}_traffic.json")
    if os.path.exists(tpath) and not args.per_record_launch:
        with open(tpath) as fh:
            tj = json.load(fh)
        if tj.get("rows") == rows and tj.get("kernel") == kernel_name:
            traffic = tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"]
            traffic_src = os.path.relpath(tpath, os.path.dirname(os.path.abspath(__file__)))

    if rank == 0:
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        line = {
            "metric": "rows/sec filter+group-by on Prometheus Arrow (HBM-resident)" if not args.host_records else "rows/sec filter+group-by on Prometheus Arrow (HOST records, PCIe-inclusive; secondary)",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"cfg{args.config if world == 1 else 4}: Prometheus schema, {rows} rows/GPU × {world} GPU, {qdesc}",
                       "rows_per_gpu": rows, "records_per_gpu": n_chunks, "groups": args.groups if args.config == 5 else 1025,
                       "parallelism": (f"parts sharded over {world} GPU(s); " + ("RCCL all-to-all of hash-partitioned partial tables, result sharded" if args.config == 5 else "RCCL all-reduce of partial tables")) if world > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "avg_launch_ms": k_ms / max(k_launches, 1),
                         "algorithmic_bytes_per_launch": k_bytes / max(k_launches, 1),
                         "bytes_per_row": k_bytes / max(rows * args.steps, 1),
                         "measured_read_ceiling": ceiling, "frac_of_measured_ceiling": achieved / ceiling if ceiling else None},
            "cpu_baseline": cpu,
            "setup": {"gen_and_upload_s": t_gen, "hbm_resident_bytes": hbm_bytes},
        }
        print(json.dumps(line))
    if world > 1 or args.force_merge:
        dist.destroy_process_group()


def cpu_baseline(sample, filt, aggs, groups, target_seconds):
    """Times oracle.OraclePlan.execute (T chains → Synchronizer → final stage, the reference's algorithm restated
    in C++) on a bounded sample of the same workload; the Go reference itself cannot run here (no Go toolchain)."""
    import oracle
    from oracle import OracleBatch, OraclePlan
    threads = os.cpu_count() or 1
    bs = 65536
    # import the sample once; time `passes` passes over it so that the timed region is ≈ target_seconds of wall
    nrows = sample.num_rows
    batches = [OracleBatch.from_arrow(sample.slice(o, min(bs, nrows - o))) for o in range(0, nrows, bs)]

    def run(passes):
        plan = OraclePlan(filt, aggs, groups, nchains=threads)
        t = time.perf_counter()
        res = plan.execute(batches * passes, threads)
        dt = time.perf_counter() - t
        res.close(); plan.close()
        return dt

    dt = run(1)
    passes = int(max(1, min(16, target_seconds / max(dt, 1e-3))))
    if passes > 1:
        dt = run(passes)
    rate = nrows * passes / dt
    for b in batches:
        b.close()
    nrows = nrows * passes
    return {"value": rate, "unit": "rows/s", "cores": threads, "kind": "port",
            "sample": f"{nrows} rows ({passes} passes over the first resident record) in {bs}-row records, {threads} chains, {dt:.2f} s wall"}


if __name__ == "__main__":
    main()
