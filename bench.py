#!/usr/bin/env python
"""bench.py — rows/sec of the fused filter + group-by aggregate over synthetic Prometheus Arrow data.

Contract: ``python bench.py --gpus N --steps K --warmup W``.

N = 1 (default): the headline of BASELINE.json's ``metric`` — cfg 2's query ``labels.code=='200' + SUM(value) GROUP BY
labels.path`` (16.25 algorithmic B/row) over **1 B Prometheus rows** resident in the HBM of one GPU.

N > 1: **cfg 4 as BASELINE.json states it** — the same 1 B rows *in total*, sharded over the N GPUs (``shard_rows``: 125 M
rows per GPU at N = 8; strong scaling), every GPU scanning its own resident parts with no data-path collective, then ONE merge
of the per-GPU partial tables over RCCL through the C ABI (``fdb_plan_allreduce``; ≙ Synchronizer + HashAggregate(final=true),
physicalplan.go:438-471). ``--weak`` keeps ``--rows`` (default 1 B) rows on EVERY GPU instead and says ``"scaling": "weak"``.
Two ways to be N ranks:
  * one process per GPU (the driver's form): under ``python -m torch.distributed.run`` the ranks read RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the env; started plainly (``python bench.py --gpus 8``) the script launches those ranks itself
    and refuses loudly if the box has fewer GPUs;
  * ``--one-process``: ONE process, one thread per GPU, ``fdb_comm_init_all`` (ncclCommInitAll) — the reference's own model of
    N chains in one process (physicalplan.go:22, :337-347). ``--force-local`` swaps RCCL for the library's in-process
    peer-to-peer transport, whose ranks may share a device: the functional check of the N-rank path on a 1-GPU box.

A *step* is one full pass of the hot path over this rank's shard: create the operator chain (fdb_plan_create), scan every
resident record with the fused HIP kernel (fdb_plan_push_batches), merge (N > 1) and produce the final Arrow record
(fdb_plan_finish). Inputs are resident in HBM before the timed region; the result of the timed path is checked against a numpy
restatement of the query (summed over the ranks at N > 1).

Rank 0 prints ONE JSON line: the metric, the HBM roofline of the scan kernel (hipEvent-timed on the plan's own stream over the
timed steps), the CPU baseline (the oracle's restatement of the reference's algorithm on this box's host cores, bounded
sample) and — at N = 1 with the default workload — ``other_configs``: the same measurement for BASELINE.json's cfg 3 and
cfg 5, the device-side ``filter()`` compaction (``select``), the PCIe-inclusive ``fdb_plan_push`` of host records
(``host_records``), Parquet row groups decoded on the device (``parquet``) and what run-time specialisation cost (``jit``).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HEADLINE_ROWS = 1_000_000_000
PROFILE_ROUNDS = ("round6", "round5", "round4", "round3", "round2", "round1")  # newest committed PMC pass of the same command first


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 5],
                    help="0 (default): the headline — cfg 2's query over 1 B rows (one GPU: all of them; N GPUs: sharded = cfg 4), plus the other lines at N=1; 2 / 3 / 5: that BASELINE.json config alone (100 M rows)")
    ap.add_argument("--rows", type=int, default=0, help="TOTAL rows of the workload (default: 1 B for the headline, 100 M for --config 2/3/5); with --weak: rows per GPU")
    ap.add_argument("--weak", action="store_true", help="N > 1: every GPU holds --rows rows (weak scaling) instead of a 1/N shard of them")
    ap.add_argument("--one-process", action="store_true", help="N > 1: one process, one thread per GPU, fdb_comm_init_all (ncclCommInitAll)")
    ap.add_argument("--force-local", action="store_true",
                    help="N > 1 in one process over the library's in-process transport (fdb_comm_init_local); ranks share devices when the box has fewer than N")
    ap.add_argument("--batch-rows", type=int, default=25_000_000, help="rows per resident record (part)")
    ap.add_argument("--groups", type=int, default=10_000_000, help="cfg 5: distinct groups")
    ap.add_argument("--cfg5-sorted", action="store_true", help="cfg 5: every record's rows ordered by group (a scan of a table sorted by its label columns)")
    ap.add_argument("--push-order", default="", help="resident records are pushed in this order (comma-separated indices), e.g. 2,0,3,1: a sorted table then arrives as several ordered sets")
    ap.add_argument("--cfg2-sorted", action="store_true", help="cfg 2: the table sorted by labels.path, the plan an OrderedAggregate (the run kernel's wide records: 1 024 path values)")
    ap.add_argument("--cfg5-wide-dicts", action="store_true", help="cfg 5: label dictionaries of 512 … 65 532 entries (with --cfg5-sorted: the run kernel's medium records)")
    ap.add_argument("--rows-per-thread", type=int, default=0, help="0: slot kernel (default); 4/8: sequential kernel")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--host-records", action="store_true",
                    help="secondary measurement: push HOST Arrow records (fdb_plan_push: PCIe copy + scan per record) instead of HBM-resident parts; never the headline value")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant: 0 default (run-time specialised), 2: 256 threads, 3: 1024 threads, 4: interpreting kernel only")
    ap.add_argument("--per-record-launch", action="store_true", help="one kernel launch per resident record instead of one per scan")
    ap.add_argument("--force-merge", action="store_true", help="run the RCCL merge path even with one rank (functional check on a 1-GPU box)")
    ap.add_argument("--torch-merge", action="store_true", help="merge through torch.distributed (frostdb_amd.distributed) instead of the C ABI's fdb_comm_*")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-parity", action="store_true", help="skip the oracle-vs-GPU comparison on the first resident record (checked.oracle)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg 3 / cfg 5 / select / host_records / parquet lines of the default run")
    ap.add_argument("--only-other", default="", help="comma-separated subset of other_configs to run (cfg3,cfg5,cfg5_1B,cfg5_sorted,select,host_records,parquet,…)")
    ap.add_argument("--cpu-sample-seconds", type=float, default=3.0)
    ap.add_argument("--sweep", action="store_true", help="kernel geometry sweep first (tuning aid; table on stderr)")
    return ap.parse_args(argv)


def shard_rows(total: int, world: int, rank: int) -> int:
    """Rows of rank `rank`'s shard when `total` rows are sharded over `world` GPUs (cfg 4: 1 B over 8 → 125 M each): equal
    shares, the remainder one row each to the first ranks — Σ over ranks = total, always."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank / world out of range")
    return total // world + (1 if rank < total % world else 0)


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


def expected_cfg3(batch):
    """numpy restatement of cfg 3 on one record: per path (count, min ts, max ts, sum) of the selected rows."""
    import numpy as np
    col = {n: batch.column(i) for i, n in enumerate(batch.schema.names)}

    def idx(name):
        c = col[name]
        return c.indices.fill_null(len(c.dictionary)).to_numpy(zero_copy_only=False), c.dictionary.to_pylist()

    ci, cd = idx("labels.code")
    mi, md = idx("labels.method")
    ii, idd = idx("labels.instance")
    pi, pd = idx("labels.path")
    sel = ((ci == cd.index(b"200")) | (ci == cd.index(b"500"))) & (mi == md.index(b"GET")) & (ii != len(idd))
    p = pi[sel].astype(np.int64)
    n = len(pd) + 1
    ts = col["timestamp"].to_numpy()[sel]
    val = col["value"].to_numpy()[sel]
    cnt = np.bincount(p, minlength=n)
    s = np.bincount(p, weights=val, minlength=n)
    mn = np.full(n, np.iinfo(np.int64).max)
    mx = np.full(n, np.iinfo(np.int64).min)
    np.minimum.at(mn, p, ts)
    np.maximum.at(mx, p, ts)
    return cnt, mn, mx, s


def expected_cfg5(batch):
    """numpy statistics of one cfg 5 record that pin the grouped result without a host-side group-by: row count and Σ value."""
    import numpy as np
    value = batch.column(batch.schema.get_field_index("value")).to_numpy()
    return np.array([batch.num_rows], dtype=np.int64), np.array([float(value.sum())])


EXPECTED_OPS = {2: ("sum", "sum"), 3: ("sum", "min", "max", "sum"), 5: ("sum", "sum")}  # how the per-record / per-rank answers fold
SELECT_THRESHOLD = 500.0  # `value > 500`: 50 % selectivity on U[0, 1000)


def fold_expected(config, a, b):
    import numpy as np
    f = {"sum": lambda x, y: x + y, "min": np.minimum, "max": np.maximum}
    return [f[op](x, y) for op, x, y in zip(EXPECTED_OPS[config], a, b)]


# ---- the N ranks: processes (torch.distributed) or threads of this process -------------------------------------------------------
class ProcGroup:
    """Host-side control plane of the one-process-per-GPU form: barrier, max over ranks, sums of the numpy expectations. The
    data-path merge never goes through here (it is fdb_comm_* inside the library) unless --torch-merge asks for it."""

    def __init__(self, rank, world, device, dist):
        self.rank, self.world, self.device, self.dist = rank, world, device, dist

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def device_sync(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize(self.device)

    def reduce(self, arr, op):
        import numpy as np
        import torch
        if self.world == 1:
            return arr
        t = torch.from_numpy(np.ascontiguousarray(arr)).clone()  # (all_reduce works in place: never on the caller's array)
        if self.dist.get_backend() == "nccl":  # (gloo — the CPU tests of this control plane — reduces host tensors)
            t = t.to(f"cuda:{self.device}")
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}[op])
        return t.cpu().numpy()

    def gather(self, value):
        if self.world == 1:
            return [value]
        out = [None] * self.world
        self.dist.all_gather_object(out, value)
        return out


class ThreadShared:
    def __init__(self, world):
        self.bar = threading.Barrier(world)
        self.slots = [None] * world


class ThreadGroup:
    """The same control plane for N ranks that are threads of this process (--one-process / --force-local)."""

    def __init__(self, rank, world, device, shared):
        self.rank, self.world, self.device, self.shared = rank, world, device, shared

    def barrier(self):
        self.shared.bar.wait()

    def device_sync(self):
        import torch
        torch.cuda.synchronize(self.device)

    def gather(self, value):
        self.shared.slots[self.rank] = value
        self.shared.bar.wait()
        out = list(self.shared.slots)
        self.shared.bar.wait()
        return out

    def reduce(self, arr, op):
        import numpy as np
        parts = self.gather(arr)
        f = {"sum": np.add, "min": np.minimum, "max": np.maximum}[op]
        out = parts[0].copy()
        for p in parts[1:]:
            out = f(out, p)
        return out


def self_launch(args):
    """`python bench.py --gpus N` started plainly: become N ranks (one per GPU) under torch.distributed.run."""
    import socket
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    n_dev = int(out.stdout.strip() or 0) if out.returncode == 0 else 0
    if n_dev < args.gpus and not (os.environ.get("FDB_BENCH_TEST_SHARE_DEVICE") == "1" and n_dev >= 1):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {n_dev} GPU(s); refusing to report a {args.gpus}-GPU number from fewer devices "
                         "(--force-local runs the N-rank path on the devices there are, as a functional check)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


class Workload:
    """One configuration made resident on this rank's GPU + the numpy expectation of its query."""

    def __init__(self, args, config, rows, rank, device, keep_host=0, gen_threads=None, cfg5_sorted=False, cfg5_wide=False, cfg2_sorted=False):
        from frostdb_amd import physicalplan as pp
        from frostdb_amd import synth
        from frostdb_amd.logicalplan import to_desc
        self.args, self.config, self.rows, self.rank, self.device = args, config, rows, rank, device
        self.filt, self.aggs, self.groups, self.qdesc = query(config)
        # cfg 5 over a table SORTED by its label columns: the reference plans an OrderedAggregate for it (physicalplan.go:433-449,
        # :525-560) — one aggregation, input ordered by the group columns — and so does this workload (fdb_plan_desc.ordered)
        self.ordered = bool((config == 5 and (cfg5_sorted or args.cfg5_sorted)) or (config == 2 and (cfg2_sorted or args.cfg2_sorted)))
        self.wide_dicts = bool(config == 5 and (cfg5_wide or args.cfg5_wide_dicts))
        if self.ordered:
            self.qdesc += "; table sorted by its label columns → OrderedAggregate" if config == 5 else "; table sorted by labels.path → OrderedAggregate (1 024 path values: wide run records)"
        if self.wide_dicts:
            self.qdesc += "; label dictionaries of 512 / 1 024 / 4 096 / 65 532 entries (key ids of two bytes: medium run records)"
        self.desc = to_desc(self.filt, self.aggs, self.groups, ordered=self.ordered)  # planned once; every step instantiates a fresh operator chain
        t0 = time.time()
        br = args.batch_rows
        n_chunks = (rows + br - 1) // br
        sizes = [min(br, rows - i * br) for i in range(n_chunks)]
        self.n_chunks = n_chunks
        self.select_expected = []  # rows of the first records with value > SELECT_THRESHOLD (the `select` line)

        def gen(i):
            if config == 5:
                b = synth.cfg5_chunk(rank, i, sizes[i], n_groups=args.groups, sorted_rows=self.ordered, of_chunks=n_chunks if self.ordered else 0, wide_dicts=self.wide_dicts)
                return b, expected_cfg5(b), None
            b = synth.prometheus_chunk(rank, i, sizes[i], row_base=i * br, cfg3=(config == 3), sorted_by_path=n_chunks if self.ordered else 0)
            sel = None
            if i < keep_host:
                v = b.column(b.schema.get_field_index("value")).to_numpy()
                sel = int((v > SELECT_THRESHOLD).sum())
            return b, (expected_cfg2(b) if config == 2 else expected_cfg3(b)), sel

        if config == 5:
            synth.cfg5_chunk(rank, 0, 8, n_groups=args.groups)  # builds the per-group digit tables once, before the thread pool
        self.resident, self.host_batches, self.sample = [], [], None
        self.expected = None
        workers = gen_threads or max(1, min(16, n_chunks, (os.cpu_count() or 8) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
        with ThreadPoolExecutor(max_workers=workers) as ex:
            for i, (b, e, sel) in enumerate(ex.map(gen, range(n_chunks))):
                self.expected = list(e) if self.expected is None else fold_expected(config, self.expected, e)
                if args.host_records or i < keep_host:
                    self.host_batches.append(b)
                    self.select_expected.append(sel)
                if not args.host_records:
                    self.resident.append(pp.ResidentBatch(b, device=device))
                if i == 0:
                    self.sample = b
        self.t_gen = time.time() - t0
        self.hbm_bytes = sum(r.device_bytes for r in self.resident)
        self.oracle_checked = None  # what bench.oracle_parity found for this workload (run once)

    def release(self):
        for r in self.resident:
            r.close()
        self.resident, self.host_batches = [], []

    # ---- the result of the timed path against the numpy restatement (`expected`: this rank's, or the sum over all ranks) ------
    def check(self, out, expected, total_rows):
        from frostdb_amd import synth
        names = out.schema.names
        col = lambda n: out.column(names.index(n))  # noqa: E731
        if self.config == 5:
            n_rows, total = int(expected[0][0]), float(expected[1][0])
            n_out = out.num_rows
            s = col("value" if self.ordered else "sum(value)").to_numpy()  # (a partial-stage OrderedAggregate names its result after the column, ordered_aggregate.go:551-557)
            assert n_rows == total_rows, (n_rows, total_rows)
            assert n_out <= self.args.groups and (total_rows < 5 * self.args.groups or n_out > 0.99 * self.args.groups), n_out
            assert math.isclose(float(s.sum()), total, rel_tol=1e-9), (float(s.sum()), total)
            res = {"groups_out": n_out, "sum_check": "Σ sum(value) == Σ value (1e-9 rel)"}
            if self.ordered:  # the OrderedAggregate's record is in key order and holds every group once: ranks of the decoded group ids strictly increase
                import numpy as np
                ranks = synth._rev4_12(synth.cfg5_decode_group_ids(out))
                assert bool(np.all(ranks[1:] > ranks[:-1])), "ordered result is not strictly increasing in key order"
                res["key_order"] = "strictly increasing (every group once, sorted by the label columns)"
            return res
        paths = synth.PATHS + [None]
        key = col("labels.path")
        key = key.dictionary_decode() if hasattr(key, "dictionary_decode") else key
        got_keys = key.to_pylist()
        if self.config == 2:
            exp_sum, exp_cnt = expected
            got = dict(zip(got_keys, col("value" if self.ordered else "sum(value)").to_pylist()))  # (partial-stage OrderedAggregate naming, ordered_aggregate.go:551-557)
            for i, p in enumerate(paths):
                if exp_cnt[i] == 0:
                    assert p not in got, p
                else:
                    assert math.isclose(got[p], exp_sum[i], rel_tol=1e-9), (p, got[p], exp_sum[i])
            assert len(got) == int((exp_cnt > 0).sum())
            res = {"groups_out": len(got), "selected_rows": int(exp_cnt.sum())}
            if self.ordered:  # every group once, in key order (NULL last)
                ks = [(k is None, k or b"") for k in got_keys]
                assert all(a < b for a, b in zip(ks[:-1], ks[1:])), "ordered result is not strictly increasing in key order"
                res["key_order"] = "strictly increasing (every group once, sorted by labels.path, NULL last)"
            return res
        cnt, mn, mx, s = expected
        rows = dict(zip(got_keys, zip(col("count(value)").to_pylist(), col("min(timestamp)").to_pylist(), col("max(timestamp)").to_pylist(),
                                      col("sum(value)").to_pylist())))
        for i, p in enumerate(paths):
            if cnt[i] == 0:
                assert p not in rows, p
                continue
            g = rows[p]
            assert g[0] == int(cnt[i]) and g[1] == int(mn[i]) and g[2] == int(mx[i]), (p, g, cnt[i], mn[i], mx[i])
            assert math.isclose(g[3], s[i], rel_tol=1e-9), (p, g[3], s[i])
        assert len(rows) == int((cnt > 0).sum())
        return {"groups_out": len(rows), "selected_rows": int(cnt.sum())}


def run_workload(args, wl, steps, warmup, group, comm, total_rows, resident_finish=False, push_order=None):
    """Correctness step, warm-up, then the timed region (barrier + device sync on both sides, max over ranks).
    `push_order`: the resident records are pushed in this order instead of the table's (an ordered workload then arrives as several ordered sets).
    `resident_finish`: the step ends with fdb_plan_finish_batch (the result record stays in HBM, for a consumer on the device)
    instead of fdb_plan_finish (Arrow record on the host)."""
    import numpy as np
    from frostdb_amd import physicalplan as pp
    rank, device, world = wl.rank, wl.device, group.world
    merging = world > 1 or args.force_merge
    records = [wl.resident[i] for i in push_order] if push_order else wl.resident

    def step(timing=False, tuning=None):
        plan = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=device, desc=wl.desc)
        if timing:
            plan.set_timing(True)
        if tuning:
            plan.set_tuning(*tuning)
        else:
            plan.set_tuning(args.rows_per_thread, args.grid | (args.variant << 25))
        if args.host_records:
            for hb in wl.host_batches:
                plan.Callback(hb)
        elif args.per_record_launch:
            for rb in records:
                plan.Callback(rb)
        else:
            plan.CallbackResident(records)
        scan_kernel = plan.last_kernel()
        out = None
        if merging and args.torch_merge:
            import torch
            from frostdb_amd.distributed import layout_probe, merge_plan, merge_plan_alltoall
            if wl.config == 5:  # high cardinality: hash-partitioned all-to-all; every rank finishes its own shard of the groups
                shard = merge_plan_alltoall(plan, device=torch.device("cuda", device))
                try:
                    out = shard.Finish()
                finally:
                    shard.Close()
            else:
                probe = layout_probe(plan, torch.device("cuda", device))  # overlaps with the scan kernel
                out = merge_plan(plan, probe=probe)
        elif merging:
            if wl.config == 5:
                shard = comm.merge_alltoall(plan)
                try:
                    out = shard.Finish()
                finally:
                    shard.Close()
            else:
                out = comm.merge(plan, dst=0)  # rank 0 gets the record, the others None
        elif resident_finish:
            rb = plan.FinishResident()
            out = rb.to_arrow() if not timing and tuning is None and not step.checked else None  # (exported once, for the check)
            step.checked = True
            rb.close()
        else:
            out = plan.Finish()
        st = plan.stats() if timing else None
        if st is not None:
            st["kernel"] = scan_kernel
        step.after_finish = plan.last_kernel()  # (an ordered Finish that had to sort its runs names the kernels it ran)
        plan.Close()
        return out, st

    step.checked = False
    # ---- the first step of this shape in this process: pays hiprtc (or the disk cache) for its kernels ----
    jit0 = pp.jit_stats()
    t_first = time.perf_counter()
    out, _ = step()
    first_step_ms = (time.perf_counter() - t_first) * 1e3
    jit1 = pp.jit_stats()
    checked = None
    if not args.host_records:
        expected = [group.reduce(np.asarray(e), op) for e, op in zip(wl.expected, EXPECTED_OPS[wl.config])]
        if wl.config == 5 and world > 1:  # the all-to-all merge leaves every rank with its own disjoint shard of the groups
            sums = group.reduce(np.array([float(out.column(out.schema.names.index("sum(value)")).to_numpy().sum())]), "sum")
            n_out = int(group.reduce(np.array([out.num_rows], dtype=np.int64), "sum")[0])
            assert int(expected[0][0]) == total_rows and 0 < n_out <= args.groups, (int(expected[0][0]), total_rows, n_out)
            assert math.isclose(float(sums[0]), float(expected[1][0]), rel_tol=1e-9), (float(sums[0]), float(expected[1][0]))
            checked = {"groups_out": n_out, "sum_check": "Σ over ranks of Σ sum(value) == Σ value (1e-9 rel)",
                       "against": "numpy expectation (row count and Σ value of every generated record); not the oracle"}
        elif rank == 0:
            checked = wl.check(out, expected, total_rows)
            # what the timed path's result was compared with: at these sizes a numpy restatement of the query computed per generated
            # record (bench.py expected_cfg2 / _cfg3 / _cfg5) — the ORACLE checks the same path at sizes it finishes in seconds (tests/)
            checked["against"] = ("numpy expectation: every group's aggregates (bench.py expected_cfg%d), summed over records%s; not the oracle" % (wl.config, " and ranks" if world > 1 else "")
                                  if wl.config in (2, 3) else "numpy expectation (row count and Σ value of every generated record) + group count bound; not the oracle")
    del out
    # ---- the oracle on the same rows (world 1, rank 0): the first resident record (cfg 5: its first 2^19 rows) ----
    if checked is not None and world == 1 and not args.host_records and not args.no_oracle_parity and wl.resident and wl.sample is not None and not wl.oracle_checked:
        checked["oracle"] = oracle_parity(wl, max_rows=(1 << 19) if wl.config == 5 else None)
        wl.oracle_checked = checked["oracle"]
        checked["against"] = ("oracle (first resident record: %d rows, %d groups) + " % (checked["oracle"]["rows"], checked["oracle"]["groups"])) + checked["against"].replace("; not the oracle", "")
    elif checked is not None and getattr(wl, "oracle_checked", None):
        checked["oracle"] = wl.oracle_checked
        checked["against"] = ("oracle (first resident record: %d rows, %d groups) + " % (checked["oracle"]["rows"], checked["oracle"]["groups"])) + checked["against"].replace("; not the oracle", "")

    if args.sweep:
        for rpt, grid in [(4, 512), (0, 0), (0, 1024), (0, 1024 | (1 << 24)), (0, 2048)]:
            for _ in range(2):
                step(tuning=(rpt, grid))
            tot_ms, tot_b, n = 0.0, 0, 0
            for _ in range(5):
                _, st = step(timing=True, tuning=(rpt, grid))
                tot_ms += st["kernel_ms"]; tot_b += st["algorithmic_bytes"]; n += st["launches"]
            if rank == 0:
                print(f"rpt={rpt} grid={grid & 0xFFFFF:5d} atomic_flush={grid >> 24}  kernel {tot_ms / n:8.4f} ms/launch  {tot_b / tot_ms / 1e6:8.1f} GB/s",
                      file=sys.stderr)

    for _ in range(warmup):
        step()
    group.barrier()
    group.device_sync()
    t0 = time.perf_counter()
    k_ms, k_bytes, k_launches, kernel_name, merge_ms = 0.0, 0, 0, "", 0.0
    for _ in range(steps):
        _, st = step(timing=True)
        k_ms += st["kernel_ms"]; k_bytes += st["algorithmic_bytes"]; k_launches += st["launches"]; merge_ms += st["merge_ms"]
        kernel_name = st["kernel"]
    group.barrier()
    group.device_sync()
    elapsed = time.perf_counter() - t0
    elapsed = float(group.reduce(np.array([elapsed]), "max")[0])
    # every rank's own kernel figures (rank order): the scan is rank-local, so the roofline fraction is a per-GPU quantity
    # (rccl_ranks_seen: what the transport itself reports as the communicator's size — ncclCommCount for RCCL — so that a line
    # claiming N GPUs shows N ranks INSIDE the communicator of every rank, not just N processes)
    per_rank = group.gather({"rank": rank, "rows": wl.rows, "kernel_ms_per_step": k_ms / max(steps, 1),
                             "kernel_frac": (k_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else 0.0,
                             "merge_ms_per_step": merge_ms / max(steps, 1),
                             "rccl_ranks_seen": comm.transport_ranks if comm is not None and hasattr(comm, "transport_ranks") else None})
    return {"elapsed": elapsed, "k_ms": k_ms, "k_bytes": k_bytes, "k_launches": k_launches, "kernel": kernel_name, "checked": checked, "after_finish": getattr(step, "after_finish", None),
            "per_rank": per_rank, "merge_ms": max(p["merge_ms_per_step"] for p in per_rank),
            "first_step_ms": first_step_ms, "jit_compiled": jit1["compiled"] - jit0["compiled"], "jit_compile_ms": jit1["compile_ms"] - jit0["compile_ms"],
            "jit_disk_loads": jit1["disk_loads"] - jit0["disk_loads"]}


def traffic_for(tag, rows, kernel_name):
    """HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside the timed process)."""
    for rnd in PROFILE_ROUNDS:
        tpath = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("rows") == rows and tj.get("kernel") == kernel_name:
                return tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"], os.path.relpath(tpath, ROOT), bool(tj.get("per_average_launch"))
    return None, None, False


def roofline_of(r, rows, steps, tag, ceiling=None):
    achieved = r["k_bytes"] / (r["k_ms"] * 1e-3) / 1e9 if r["k_ms"] > 0 else 0.0
    launches = max(r["k_launches"], 1)
    traffic, src, per_avg = traffic_for(tag, rows, r["kernel"])
    # (a scan cut into several launches — the hash path's chunks — reports per-launch figures of the average launch)
    if traffic is not None and launches != steps and not per_avg:
        traffic, src = None, None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": r["kernel"], "avg_launch_ms": r["k_ms"] / launches,
            "launches_per_step": launches / max(steps, 1),
            "algorithmic_bytes_per_launch": r["k_bytes"] / launches, "bytes_per_row": r["k_bytes"] / max(rows * steps, 1),
            "whole_step_frac": (r["k_bytes"] / max(steps, 1)) / (r["elapsed"] / max(steps, 1)) / 1e9 / HBM_PEAK_GBS,
            "measured_read_ceiling": ceiling, "frac_of_measured_ceiling": achieved / ceiling if ceiling else None}


def jit_of(r):
    return {"first_step_ms": r["first_step_ms"], "kernels_compiled": r["jit_compiled"], "jit_compile_ms": r["jit_compile_ms"], "disk_cache_loads": r["jit_disk_loads"]}


_REAL_STDOUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout. RCCL prints a version banner with printf when a communicator is created (every
    rank, flushed whenever libc pleases — usually at exit, i.e. AFTER the JSON line), and other libraries may do the same: from
    here on file descriptor 1 is stderr for everybody, and the line goes to the saved descriptor."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def emit(text):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (text + "\n").encode())


# ---- the secondary measurements of the default N = 1 run (other_configs) --------------------------------------------------------
def oracle_aggregate(records, filt, aggs, groups):
    """oracle.OraclePlan.execute (the reference's chains → Synchronizer → final stage, restated on the CPU — the checker) over `records`."""
    import oracle  # noqa: F401
    from oracle import OracleBatch, OraclePlan
    threads, bs = os.cpu_count() or 1, 65536
    batches = [OracleBatch.from_arrow(r.slice(o, min(bs, r.num_rows - o))) for r in records for o in range(0, r.num_rows, bs)]
    plan = OraclePlan(filt, aggs, groups, nchains=threads)
    res = plan.execute(batches, threads)
    want = res.to_arrow()
    res.close(); plan.close()
    for b in batches:
        b.close()
    return want


class _PathKeyed:  # (what compare_with_oracle needs to know about a cfg 2-shaped result)
    config, ordered = 2, False


def oracle_filter_parity(rec, filt, got, max_rows=1 << 23):
    """The oracle's filter() (filter.go:276-354 restated) over the first rows of `rec` against the first rows of the device's compacted
    record `got` — every column, bit for bit (the device record is the filter of the WHOLE record: its first k rows are the filter of
    the first rows that hold k selected ones)."""
    import oracle  # noqa: F401
    from oracle import OraclePlan
    head = rec.slice(0, min(max_rows, rec.num_rows))
    plan = OraclePlan(filt)
    out, idx = plan.filter(head)
    plan.close()
    want = out.to_arrow() if out is not None else None
    if out is not None:
        out.close()
    k = len(idx)
    assert want is None or want.num_rows == k
    g = got.slice(0, k)
    for name in rec.schema.names:
        if k == 0:
            break
        a, b = g.column(g.schema.get_field_index(name)), want.column(want.schema.get_field_index(name))
        a = a.dictionary_decode() if hasattr(a, "dictionary_decode") else a
        b = b.dictionary_decode() if hasattr(b, "dictionary_decode") else b
        assert a.cast(b.type).equals(b) if a.type != b.type else a.equals(b), "oracle parity: filter() column %s differs" % name
    return {"rows": head.num_rows, "selected": k, "what": "oracle.OraclePlan.filter over the first %d rows of the first record vs. the first %d rows of the device's "
            "compacted record: every column equal, value for value" % (head.num_rows, k)}


def measure_select(wl, steps=10, warmup=2, ceiling=None):
    """`filter()` on the device (filter.go:276-354 ≙ fdb_plan_filter_batch): `value > 500` (50 % selectivity) over the first
    100 M resident rows, every column compacted. Algorithmic bytes = filter column once + every selected value / validity bit read
    once and written once; `min_traffic_frac` counts what a sector-granular memory must at least move (all input + the output)."""
    import torch
    from frostdb_amd import physicalplan as pp
    from frostdb_amd.logicalplan import Col
    recs = wl.resident[:len(wl.select_expected)]
    rows = sum(r.num_rows for r in recs)
    filt = Col("value") > SELECT_THRESHOLD

    def step(timing=False):
        plan = pp.HashAggregatePlan(filt, device=wl.device)
        if timing:
            plan.set_timing(True)
        outs = plan.FilterResidentMany(recs)
        st = plan.stats() if timing else None
        kernel = plan.last_kernel()
        plan.Close()
        return outs, st, kernel

    outs, _, _ = step()
    got = [o.num_rows for o in outs]
    assert got == wl.select_expected, (got, wl.select_expected)
    first = outs[0].to_arrow()
    v0 = wl.host_batches[0].column(wl.host_batches[0].schema.get_field_index("value")).to_numpy()
    assert first.column(first.schema.get_field_index("value")).to_numpy().tolist() == v0[v0 > SELECT_THRESHOLD].tolist()
    oracle_checked = oracle_filter_parity(wl.host_batches[0], filt, first) if not wl.args.no_oracle_parity else None
    for o in outs:
        o.close()
    for _ in range(warmup):
        for o in step()[0]:
            o.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_ms = k_bytes = k_launches = 0
    kernel = ""
    for _ in range(steps):
        outs, st, kernel = step(timing=True)
        k_ms += st["kernel_ms"]; k_bytes += st["algorithmic_bytes"]; k_launches += st["launches"]
        for o in outs:
            o.close()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sel = sum(wl.select_expected) / rows
    row_in = sum(r.device_bytes for r in recs) / rows  # ≈ 24.25: code 4 + path 4 + timestamp 8 + value 8 + two validity bits
    min_traffic = rows * (row_in + sel * row_in)
    ach = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, src, _ = traffic_for("select", rows, kernel)
    return {"workload": f"select: filter() compaction of every column, value > {SELECT_THRESHOLD:g} over {rows} resident rows ({len(recs)} records), selectivity {sel:.4f}",
            "value": rows * steps / el, "unit": "rows/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": src, "measured_read_ceiling": ceiling, "frac_of_measured_ceiling": ach / ceiling if ceiling else None,
                         "launches_per_step": k_launches / steps, "kernel_ms_per_step": k_ms / steps,
                         "algorithmic_bytes_per_row": k_bytes / (rows * steps), "min_traffic_bytes_per_row": min_traffic / rows,
                         "min_traffic_frac": (min_traffic * steps / (k_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if k_ms > 0 else 0.0,
                         "whole_step_frac": (k_bytes / steps) / (el / steps) / 1e9 / HBM_PEAK_GBS},
            "checked": {"selected_rows": sum(got), "first_record_values": "bit-identical to numpy's value[value > T]",
                        "against": ("oracle filter() (first record) + " if oracle_checked else "") + "numpy expectation (selected-row counts of every record; the first record's compacted values)" + ("" if oracle_checked else "; not the oracle"),
                        "oracle": oracle_checked}}


def h2d_rate_gbs(device, nbytes=1 << 30):
    """Pinned host → device copy rate of this box (the ceiling of anything that starts from host memory)."""
    import torch
    src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{device}")
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return 3 * nbytes / (time.perf_counter() - t0) / 1e9


def measure_host_records(wl, steps=3):
    """What `Callback(arrow.Record)` delivers (table.go:783-860): HOST Arrow records pushed through fdb_plan_push — copy of the
    referenced columns over PCIe + scan — as 25 M-row records and as 65 536-row records. PCIe-bound; never the headline."""
    import torch
    from frostdb_amd import physicalplan as pp
    h2d = h2d_rate_gbs(wl.device)
    big = wl.host_batches
    small = [big[0].slice(o, min(65536, big[0].num_rows - o)) for o in range(0, big[0].num_rows, 65536)]
    out = {"bound": "pcie", "measured_h2d_GBps": h2d, "bytes_per_row_copied": 16.25}
    for name, recs in (("records_25M_rows", big), ("records_65536_rows", small)):
        rows = sum(r.num_rows for r in recs)
        exported = [pp.ExportedBatch(r) for r in recs]  # pyarrow's C-data export is the caller's cost, not the library's

        def once():
            plan = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc)
            for ex in exported:
                plan.CallbackExported(ex)
            res = plan.Finish()
            plan.Close()
            return res
        res = once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        gbs = rows * 16.25 / dt / 1e9
        out[name] = {"records": len(recs), "rows": rows, "ms_per_pass": dt * 1e3, "value": rows / dt, "unit": "rows/s", "achieved_GBps": gbs,
                     "frac_of_h2d": gbs / h2d, "groups_out": res.num_rows}
        for ex in exported:
            ex.close()
    return out


def measure_host_records_chains(wl, rows_per_chain=2_097_152):
    """What the Go shim really drives: N chains pushing HOST records concurrently — one plan per chain on
    its own stream, one thread per chain like one goroutine per chain (physicalplan.go:337-347), records of 1 024 rows (the
    reference's batch floor, table.go:780), 8 192 and 65 536 rows, every chain its own rows. Wall time from the first Callback to
    the last chain's Finish; the per-chain results are merged (fdb_plan_merge) and checked against the numpy expectation."""
    import numpy as np
    import torch
    from frostdb_amd import physicalplan as pp
    h2d = h2d_rate_gbs(wl.device)
    src = wl.host_batches
    total_src = sum(b.num_rows for b in src)
    out = {"bound": "pcie / host", "measured_h2d_GBps": h2d, "bytes_per_row_copied": 16.25, "rows_per_chain": rows_per_chain}
    pinned = [True]
    for chains in (1, 8, 32):
        per_chain = min(rows_per_chain if chains <= 8 else rows_per_chain // 2, total_src // chains // 65536 * 65536)
        for rec_rows in (1024, 8192, 65536):
            # chain c owns rows [c × per_chain, (c + 1) × per_chain) of the host batches laid end to end
            exported, expected = [], None
            for c in range(chains):
                lo, hi, recs = c * per_chain, (c + 1) * per_chain, []
                base = 0
                for b in src:
                    a0, a1 = max(lo, base), min(hi, base + b.num_rows)
                    for o in range(a0, a1, rec_rows):
                        recs.append(b.slice(o - base, min(rec_rows, a1 - o)))
                    base += b.num_rows
                exported.append(pp.PreparedRun([pp.ExportedBatch(r) for r in recs]))  # (pointer tables built outside the timed region: see PreparedRun)
            # numpy expectation over the rows the chains own
            need = chains * per_chain
            base = 0
            for b in src:
                take = min(b.num_rows, max(0, need - base))
                if take > 0:
                    e = expected_cfg2(b.slice(0, take))
                    expected = list(e) if expected is None else fold_expected(2, expected, e)
                base += b.num_rows
            errors = []

            def run_all():
                plans = [pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc) for _ in range(chains)]
                bar = threading.Barrier(chains + 1)

                def work(c):
                    try:
                        # several chains: every chain's thread runs on the GPU's socket (physicalplan.local_cpus) — left to the scheduler the threads spread over
                        # both sockets and 32 chains of 65 536-row records stop at 0.56–0.69 of the link instead of 0.92. ONE chain stays where the scheduler
                        # put it, next to the records it reads: pinned to the GPU's socket it read them across the socket link and lost a quarter.
                        if chains > 1:
                            pinned[0] = pp.pin_thread_near(wl.device) and pinned[0]
                        bar.wait()
                        plans[c].CallbackPrepared(exported[c])
                        plans[c].last_kernel()  # (settle: the queued records are scanned — launched from this chain's thread, not waited for)
                    except BaseException as e:  # noqa: BLE001
                        errors.append(e)
                        bar.abort()
                ts = [threading.Thread(target=work, args=(c,)) for c in range(chains)]
                for t in ts:
                    t.start()
                bar.wait()
                t0 = time.perf_counter()
                for t in ts:
                    t.join()
                for p in plans[1:]:
                    plans[0].Merge(p)
                res = plans[0].Finish()
                dt = time.perf_counter() - t0
                for p in plans:
                    p.Close()
                if errors:
                    raise errors[0]
                return res, dt
            res, _ = run_all()
            got = dict(zip(res.column(0).to_pylist(), res.column(1).to_pylist()))
            from frostdb_amd import synth
            exp_sum, exp_cnt = expected
            for i, p in enumerate(synth.PATHS + [None]):
                if exp_cnt[i]:
                    assert math.isclose(got[p], exp_sum[i], rel_tol=1e-9), (chains, rec_rows, p)
            torch.cuda.synchronize()
            dts = [run_all()[1] for _ in range(3)]
            dt = sorted(dts)[1]
            rows = chains * per_chain
            gbs = rows * 16.25 / dt / 1e9
            out[f"chains_{chains}_records_{rec_rows}"] = {"chains": chains, "record_rows": rec_rows, "records": sum(len(e) for e in exported), "rows": rows, "ms": dt * 1e3,
                                                         "value": rows / dt, "unit": "rows/s", "achieved_GBps": gbs, "frac_of_h2d": gbs / h2d,
                                                         "us_per_record_per_chain": dt * 1e6 / max(len(exported[0]), 1)}
            for run in exported:
                for ex in run.keep:
                    ex.close()
    out["checked"] = {"against": "numpy expectation: every group's sum over the rows the chains own (bench.py expected_cfg2), chains merged with fdb_plan_merge; not the oracle"}
    if not wl.args.no_oracle_parity and wl.host_batches:
        # the oracle on the path's own input: the first 2^21 host rows cut into 65 536-row records, pushed by 8 chains (one plan each, fdb_plan_push_many), merged
        import threading as _th
        head = wl.host_batches[0].slice(0, min(1 << 21, wl.host_batches[0].num_rows))
        parts = [head.slice(o, min(65536, head.num_rows - o)) for o in range(0, head.num_rows, 65536)]
        plans = [pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc) for _ in range(8)]
        errs = []

        def push(c):
            try:
                for r in parts[c::8]:
                    plans[c].Callback(r)
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
        ts = [_th.Thread(target=push, args=(c,)) for c in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        try:
            if errs:
                raise errs[0]
            for p in plans[1:]:
                plans[0].Merge(p)
            got = plans[0].Finish()
        finally:
            for p in plans:
                p.Close()
        n_groups = compare_with_oracle(_PathKeyed, got, oracle_aggregate([head], wl.filt, wl.aggs, wl.groups))
        out["checked"]["oracle"] = {"rows": head.num_rows, "groups": n_groups, "what": "oracle.OraclePlan.execute over the first %d host rows vs. 8 chains pushing them as 65 536-row "
                                    "host records (fdb_plan_push), merged with fdb_plan_merge: group sets equal, float64 sums within 1e-9 relative" % head.num_rows}
        out["checked"]["against"] = "oracle (first %d host rows through 8 chains) + " % head.num_rows + out["checked"]["against"].replace("; not the oracle", "")
    out["cpu_quota_cpus"] = cpu_quota_cpus()  # (32 chain threads on a 16-CPU quota are throttled: profiles/round6_push_bench.txt)
    out["chain_threads"] = "8 and 32 chains: pinned to the cores of the GPU's NUMA node (physicalplan.pin_thread_near); 1 chain: where the scheduler put it" if pinned[0] else "not pinned (the GPU's local_cpulist could not be read or applied)"
    return out


def measure_parquet(device, rows=20_000_000, passes=7, use_oracle=True):
    """Parquet row groups (file bytes in pinned host memory) → columns decoded on the device
    (fdb_batch_from_parquet) → cfg 2's query. Bytes = file bytes read + column bytes produced; host / device split from
    fdb_parquet_stats. Two files: UNCOMPRESSED + PLAIN, and SNAPPY pages + DELTA_BINARY_PACKED timestamps."""
    import io
    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq
    import torch
    from frostdb_amd import physicalplan as pp
    from frostdb_amd import synth
    from frostdb_amd.logicalplan import Col, Sum
    from tests.parquet_util import row_group_chunks, write_parquet
    rec = synth.prometheus_chunk(0, 0, rows)
    exp_sum, exp_cnt = expected_cfg2(rec)
    t = pa.Table.from_batches([rec])
    t = t.set_column(0, "labels.code", t.column(0).cast(pa.binary())).set_column(1, "labels.path", t.column(1).cast(pa.binary()))
    q = (Col("labels.code") == "200", [Sum(Col("value"))], [Col("labels.path")])
    # the oracle over the record the files are written from (every row group of both files decodes to it)
    oracle_want = oracle_aggregate([rec], *q) if use_oracle else None
    n_oracle = 0
    h2d = h2d_rate_gbs(device, 1 << 28)
    out = {"rows": rows, "measured_h2d_GBps": h2d}
    for variant, kw in (("plain", {}), ("delta_snappy", dict(compression="SNAPPY", column_encoding={"timestamp": "DELTA_BINARY_PACKED"},
                                                             use_dictionary=["labels.code", "labels.path"]))):
        data = write_parquet(t, row_group_size=5_000_000, data_page_size=1 << 20, **kw)
        n_rg = pq.ParquetFile(io.BytesIO(data)).metadata.num_row_groups
        pinned = torch.empty(len(data), dtype=torch.uint8, pin_memory=True)
        pinned.numpy()[:] = np.frombuffer(data, dtype=np.uint8)
        groups = []
        for g in range(n_rg):
            ch, n = row_group_chunks(data, g)
            loc = []
            for nm, ty, opt, u8, b, cd in ch:
                off = data.find(b[:256])
                assert data[off:off + len(b)] == b
                loc.append((nm, ty, opt, u8, (pinned.data_ptr() + off, len(b)), cd))
            groups.append((loc, n))

        def once():
            plan = pp.HashAggregatePlan(*q, device=device)
            # every row group of the file in ONE call (fdb_batches_from_parquet: one copy queue, the host work of all row groups side by
            # side, a row group's kernels launched while later ones are still parsed). $FDB_BENCH_PQ_WORKERS=n: the former way — one
            # call per row group from n threads of the caller (A/B)
            workers = int(os.environ.get("FDB_BENCH_PQ_WORKERS", "0"))
            if workers > 0:
                with ThreadPoolExecutor(max_workers=workers) as ex:
                    keep = list(ex.map(lambda g: pp.ResidentBatch.from_parquet(g[0], g[1], device=device), groups))
            else:
                keep = pp.ResidentBatch.from_parquet_many(groups, device=device)
            plan.CallbackResident(keep)
            res = plan.Finish()
            plan.Close()
            for k in keep:
                k.close()
            return res
        res = once()
        got = dict(zip(res.column(0).to_pylist(), res.column(1).to_pylist()))
        for i, p in enumerate(synth.PATHS + [None]):
            if exp_cnt[i]:
                assert math.isclose(got[p], exp_sum[i], rel_tol=1e-9), (variant, p)
        if oracle_want is not None:
            n_oracle = compare_with_oracle(_PathKeyed, res, oracle_want)
        s0 = pp.parquet_stats()
        thr0 = cpu_throttled()
        torch.cuda.synchronize()
        per_pass = []
        for _ in range(passes):
            t0 = time.perf_counter()
            once()
            torch.cuda.synchronize()
            per_pass.append(time.perf_counter() - t0)
        # the MEDIAN pass: the bench boxes grant 16 CPUs of the 256 they show, and a pass that runs into the quota is held for the rest
        # of a 100 ms period (`cpu_throttled_periods` counts those over the passes; the mean is kept beside it)
        dt = sorted(per_pass)[len(per_pass) // 2]
        thr1 = cpu_throttled()
        s1 = pp.parquet_stats()
        fb = (s1["file_bytes"] - s0["file_bytes"]) / passes
        ob = (s1["out_bytes"] - s0["out_bytes"]) / passes
        out[variant] = {"file_bytes": int(fb), "decoded_column_bytes": int(ob), "row_groups": n_rg, "calls_per_pass": (s1["calls"] - s0["calls"]) / passes, "passes": passes, "ms_per_pass": dt * 1e3,
                        "ms_per_pass_mean": sum(per_pass) / passes * 1e3, "ms_per_pass_min": min(per_pass) * 1e3,
                        "cpu_throttled_periods": None if thr0[0] is None else thr1[0] - thr0[0], "value": rows / dt, "unit": "rows/s",
                        "file_GBps": fb / dt / 1e9, "file_plus_columns_GBps": (fb + ob) / dt / 1e9,
                        "host_part_ms": (s1["host_ms"] - s0["host_ms"]) / passes, "device_part_ms": (s1["device_ms"] - s0["device_ms"]) / passes,
                        "bound": "host" if (s1["host_ms"] - s0["host_ms"]) > (s1["device_ms"] - s0["device_ms"]) else "pcie", "frac_of_h2d": fb / dt / 1e9 / h2d,
                        "checked": {"groups_out": res.num_rows,
                                    "against": ("oracle (all %d rows of the record the file was written from: %d groups equal, sums within 1e-9 relative) + " % (rows, n_oracle) if oracle_want is not None else "")
                                    + "numpy expectation: every group's sum (bench.py expected_cfg2 on the same record)" + ("" if oracle_want is not None else "; not the oracle")}}
        del pinned
    return out


def other_configs(args, wl, rank, device, group, comm):
    """BASELINE.json's other single-GPU configurations and the secondary measurements, same run, same method (fewer steps)."""
    only = set(x for x in args.only_other.split(",") if x)
    want = lambda k: not only or k in only  # noqa: E731
    others = {}
    from frostdb_amd import physicalplan as _pp
    ceiling = _pp.read_ceiling(device, 2 << 30, 5)  # the plain read kernel on this box, this run: every roofline below quotes it
    if want("select") and wl.select_expected:
        others["select"] = measure_select(wl, ceiling=ceiling)
    if want("host_records") and wl.host_batches:
        others["host_records"] = measure_host_records(wl)
    if want("host_records_chains") and wl.host_batches:
        others["host_records_chains"] = measure_host_records_chains(wl)
    wl.release()
    for cfg, st, wu in ((3, max(5, args.steps // 2), 2), (5, 3, 1)):
        sub5 = {"cfg5_merge", "cfg5_sorted", "cfg5_sorted_sets", "cfg5_sorted_wide"}
        if not want(f"cfg{cfg}") and not (cfg == 5 and only and (only & sub5)):
            continue
        if cfg == 5 and not want("cfg5"):  # (--only-other with cfg5_* entries alone: the unsorted scan itself is not run)
            if "cfg5_merge" in only:
                w2 = Workload(args, cfg, 100_000_000, rank, device)
                if len(w2.resident) >= 2:
                    others["cfg5_merge"] = measure_cfg5_merge(args, w2, st, wu, ceiling)
                w2.release()
            only_sorted = True
        else:
            only_sorted = False
        if not only_sorted:
            w2 = Workload(args, cfg, 100_000_000, rank, device)
            r2 = run_workload(args, w2, st, wu, group, comm, 100_000_000)
            others[f"cfg{cfg}"] = {
                "workload": f"cfg{cfg}: Prometheus schema, 100000000 rows, {w2.qdesc}",
                "value": 100_000_000 * st / r2["elapsed"], "unit": "rows/s", "steps": st, "warmup": wu,
                "ms_per_step": r2["elapsed"] / st * 1e3, "roofline": roofline_of(r2, 100_000_000, st, f"cfg{cfg}", ceiling),
                "checked": r2["checked"], "jit": jit_of(r2), "setup": {"gen_and_upload_s": w2.t_gen, "hbm_resident_bytes": w2.hbm_bytes},
            }
            if cfg == 5 and want("cfg5_merge") and len(w2.resident) >= 2:
                others["cfg5_merge"] = measure_cfg5_merge(args, w2, st, wu, ceiling)
            if cfg == 5:  # the same scan finished for a consumer on the device (fdb_plan_finish_batch): no Arrow record crosses PCIe
                r3 = run_workload(args, w2, st, wu, group, comm, 100_000_000, resident_finish=True)
                others["cfg5_resident_finish"] = {
                    "workload": "cfg5 with the result left in HBM (fdb_plan_finish_batch) for a device-side consumer",
                    "value": 100_000_000 * st / r3["elapsed"], "unit": "rows/s", "steps": st, "warmup": wu, "ms_per_step": r3["elapsed"] / st * 1e3,
                    "roofline": roofline_of(r3, 100_000_000, st, "cfg5", ceiling), "checked": r3["checked"]}
            w2.release()
        if cfg == 5 and want("cfg5_sorted"):  # the same table SORTED by its label columns: the table-free OrderedAggregate (no hash kernel runs)
            w3 = Workload(args, 5, 100_000_000, rank, device, cfg5_sorted=True)
            r4 = run_workload(args, w3, st, wu, group, comm, 100_000_000)
            others["cfg5_sorted"] = {
                "workload": f"cfg5_sorted: Prometheus schema, 100000000 rows, {w3.qdesc}",
                "value": 100_000_000 * st / r4["elapsed"], "unit": "rows/s", "steps": st, "warmup": wu, "ms_per_step": r4["elapsed"] / st * 1e3,
                "roofline": roofline_of(r4, 100_000_000, st, "cfg5_sorted", ceiling), "checked": r4["checked"], "jit": jit_of(r4)}
            r5 = run_workload(args, w3, st, wu, group, comm, 100_000_000, resident_finish=True)
            others["cfg5_sorted_resident_finish"] = {
                "workload": "cfg5_sorted with the result left in HBM (fdb_plan_finish_batch)",
                "value": 100_000_000 * st / r5["elapsed"], "unit": "rows/s", "steps": st, "warmup": wu, "ms_per_step": r5["elapsed"] / st * 1e3,
                "roofline": roofline_of(r5, 100_000_000, st, "cfg5_sorted", ceiling), "checked": r5["checked"]}
            if want("cfg5_sorted_sets") and len(w3.resident) == 4:
                # the sorted table's four records pushed as FOUR ORDERED SETS (third, first, fourth, second quarter of the key range): Finish finds the
                # keys out of order and sorts the runs on the device (runs_sort_keys_kernel + a radix sort per 64 bits of key rank) — no hash kernel
                r8 = run_workload(args, w3, st, wu, group, comm, 100_000_000, push_order=[2, 0, 3, 1])
                others["cfg5_sorted_sets"] = {
                    "workload": "cfg5_sorted pushed as 4 ordered sets (records in the order 3, 1, 4, 2): the runs are sorted by key on the device at Finish",
                    "value": 100_000_000 * st / r8["elapsed"], "unit": "rows/s", "steps": st, "warmup": wu, "ms_per_step": r8["elapsed"] / st * 1e3,
                    "finish_ran": r8["after_finish"], "roofline": roofline_of(r8, 100_000_000, st, "cfg5_sorted", ceiling), "checked": r8["checked"]}
            w3.release()
        if cfg == 5 and want("cfg5_sorted_wide"):  # … with label dictionaries of 512 – 65 532 entries: the run kernel writes MEDIUM records, two bytes per key id (round 5)
            w4 = Workload(args, 5, 100_000_000, rank, device, cfg5_sorted=True, cfg5_wide=True)
            r6 = run_workload(args, w4, st, wu, group, comm, 100_000_000)
            others["cfg5_sorted_wide"] = {
                "workload": f"cfg5_sorted_wide: Prometheus schema, 100000000 rows, {w4.qdesc}",
                "value": 100_000_000 * st / r6["elapsed"], "unit": "rows/s", "steps": st, "warmup": wu, "ms_per_step": r6["elapsed"] / st * 1e3,
                "roofline": roofline_of(r6, 100_000_000, st, "cfg5_sorted_wide", ceiling), "checked": r6["checked"], "jit": jit_of(r6)}
            w4.release()
    if want("cfg5_1B"):
        # BASELINE.json's metric at cfg 5's shape: 1 B rows × 32 label columns over 10 M groups, all of it resident on ONE GPU (138.5 GB of
        # columns). At 100 M rows the scan creates its 10 M groups inside the timed step (one insert-heavy launch is a third of it); here ≈ 97 %
        # of the rows meet a group that exists — the steady state of the probing kernel, which is what the metric's scale measures.
        n1b = 1_000_000_000
        w6 = Workload(args, 5, n1b, rank, device)
        r9 = run_workload(args, w6, 2, 1, group, comm, n1b)
        others["cfg5_1B"] = {
            "workload": f"cfg5_1B: Prometheus schema, {n1b} rows resident on one GPU, {w6.qdesc}",
            "value": n1b * 2 / r9["elapsed"], "unit": "rows/s", "steps": 2, "warmup": 1, "ms_per_step": r9["elapsed"] / 2 * 1e3,
            "roofline": roofline_of(r9, n1b, 2, "cfg5_1B", ceiling), "checked": r9["checked"],
            "setup": {"gen_and_upload_s": w6.t_gen, "hbm_resident_bytes": w6.hbm_bytes}}
        w6.release()
    if want("cfg2_sorted"):  # the benchmark's own schema and query over a table sorted by labels.path: table-free OrderedAggregate, wide run records
        w5 = Workload(args, 2, 100_000_000, rank, device, cfg2_sorted=True)
        st2 = max(5, args.steps // 2)
        r7 = run_workload(args, w5, st2, 2, group, comm, 100_000_000)
        others["cfg2_sorted"] = {
            "workload": f"cfg2_sorted: Prometheus schema, 100000000 rows, {w5.qdesc}",
            "value": 100_000_000 * st2 / r7["elapsed"], "unit": "rows/s", "steps": st2, "warmup": 2, "ms_per_step": r7["elapsed"] / st2 * 1e3,
            "roofline": roofline_of(r7, 100_000_000, st2, "cfg2_sorted", ceiling), "checked": r7["checked"], "jit": jit_of(r7)}
        w5.release()
    if want("parquet"):
        others["parquet"] = measure_parquet(device, use_oracle=not args.no_oracle_parity)
    return others


def run_rank(args, group, device, comm):
    """Everything one rank does once it has its device, its control-plane group and (N > 1) its communicator endpoint.
    Returns the JSON line's dict on rank 0, None elsewhere."""
    from frostdb_amd import physicalplan as pp
    rank, world = group.rank, group.world
    headline = args.config == 0
    config = 2 if headline else args.config
    total = args.rows or (HEADLINE_ROWS if headline else 100_000_000)
    if world > 1 and not args.weak:
        rows, scaling, total_rows = shard_rows(total, world, rank), "strong", total
    else:
        rows, scaling, total_rows = total, "weak", total * world
    merging = world > 1 or args.force_merge
    secondary = headline and world == 1 and not args.no_other_configs and not args.host_records
    wl = Workload(args, config, rows, rank, device, keep_host=4 if secondary else 0,
                  gen_threads=max(1, min(16, (os.cpu_count() or 8) // world)) if isinstance(group, ThreadGroup) else None)
    shard_sizes = group.gather(rows)
    assert sum(shard_sizes) == total_rows, (shard_sizes, total_rows)
    r = run_workload(args, wl, args.steps, args.warmup, group, comm, total_rows, push_order=[int(x) for x in args.push_order.split(",") if x] or None)
    value = total_rows * args.steps / r["elapsed"]

    # ---- CPU baseline (rank 0, N = 1 only): the oracle's restatement on this box's host cores, bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl.sample, wl.filt, wl.aggs, wl.groups, args.cpu_sample_seconds)
    ceiling = None
    if rank == 0 and world == 1:
        ceiling = pp.read_ceiling(device, 2 << 30, 5)  # plain read kernel on this box, this run

    tag = ("cfg1B" if rows == HEADLINE_ROWS and config == 2 else f"cfg{config}")
    if world > 1 and config == 2:
        name = (f"cfg4: {total} rows sharded over {world} GPUs" if scaling == "strong" else f"cfg4-weak: {total} rows on each of {world} GPUs")
    else:
        name = "headline" if headline else f"cfg{config}"
    transport = None
    if merging:
        transport = ("torch.distributed (RCCL)" if args.torch_merge else "in-process peer-to-peer (fdb_comm_init_local)" if args.force_local
                     else "RCCL, one process (fdb_comm_init_all)" if args.one_process else "RCCL, one process per GPU (fdb_comm_init_rank)")
    line = {
        "metric": "rows/sec filter+group-by on 1B-row Prometheus Arrow (HBM-resident); achieved HBM GB/s vs peak" if not args.host_records
                  else "rows/sec filter+group-by on Prometheus Arrow (HOST records, PCIe-inclusive; secondary)",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{name}: Prometheus schema, {wl.qdesc}",
                   "total_rows": total_rows, "rows_per_gpu": shard_sizes if world > 1 else rows, "records_per_gpu": wl.n_chunks,
                   "groups": args.groups if config == 5 else 1025,
                   "parallelism": (f"parts sharded over {world} GPU(s), no data-path collective; " +
                                   ("all-to-all of hash-partitioned partial tables, result sharded" if config == 5
                                    else "in-place all-reduce of the partial tables (fdb_plan_allreduce)" if not args.torch_merge
                                    else "all-reduce of the partial tables" +
                                    ("; fallback: the C-ABI communicator could not be joined" if getattr(args, "merge_fallback", False) else "")) +
                                   f"; transport: {transport}") if merging else "1 GPU"},
        "roofline": roofline_of(r, rows, args.steps, tag, ceiling),
        "cpu_baseline": cpu,
        "checked": r["checked"],
        "jit": jit_of(r),
        "setup": {"gen_and_upload_s": wl.t_gen, "hbm_resident_bytes": wl.hbm_bytes},
    }
    if merging:
        line["merge_ms"] = r["merge_ms"]          # device time of the merge collectives per step (hipEvents on the plan's stream), max over ranks
        line["per_rank"] = r["per_rank"]
        line["rccl_ranks_seen"] = [p.get("rccl_ranks_seen") for p in r["per_rank"]]
    if getattr(args, "share_device", False):
        import torch
        line["devices_used"] = min(world, torch.cuda.device_count())
        line["note"] = "FDB_BENCH_TEST_SHARE_DEVICE: functional check of the one-process-per-rank path with ranks sharing devices; not an N-GPU measurement"
    if args.force_local:
        import torch
        line["devices_used"] = min(world, torch.cuda.device_count())
        line["note"] = "functional check of the N-rank path over the in-process transport; not an N-GPU measurement unless devices_used == n_gpus"

    if secondary:
        line["other_configs"] = other_configs(args, wl, rank, device, group, comm)
        line["jit_process_total"] = pp.jit_stats()
    else:
        wl.release()
    return line if rank == 0 else None


def main_one_process(args):
    """N ranks as N threads of this process (the reference's own model: N chains in one process)."""
    import torch
    from frostdb_amd import build as fb
    fb.build()
    from frostdb_amd import comm as fcomm
    n, n_dev = args.gpus, torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if args.force_local:
        devices = [r % n_dev for r in range(n)]
        comms = fcomm.Comm.init_local(devices)
    else:
        if n_dev < n:
            raise SystemExit(f"bench.py --gpus {n} --one-process: this box has {n_dev} GPU(s) (RCCL refuses two ranks on one device; --force-local runs the functional check)")
        devices = list(range(n))
        comms = fcomm.Comm.init_all(devices)
    shared = ThreadShared(n)
    results, errors = [None] * n, [None] * n

    def work(r):
        try:
            torch.cuda.set_device(devices[r])
            results[r] = run_rank(args, ThreadGroup(r, n, devices[r], shared), devices[r], comms[r])
        except BaseException as e:  # noqa: BLE001
            errors[r] = e
            shared.bar.abort()  # the other ranks leave their barriers instead of waiting for ever

    threads = [threading.Thread(target=work, args=(r,), name=f"rank{r}") for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in comms:
        c.close()
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real or any(errors):
        raise (real or [e for e in errors if e is not None])[0]
    emit(json.dumps(results[0]))


def main():
    args = parse_args()
    if args.force_local:
        args.one_process = True
    if args.one_process:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--one-process / --force-local is ONE process: do not start it under torch.distributed.run")
        claim_stdout()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return main_one_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # Test hook (tests/test_gpu_fake_rccl.py): the one-process-per-rank path on a box with fewer GPUs than ranks — the ranks share
    # devices, the control plane runs over gloo (torch's own RCCL refuses two ranks on a device) and the C-ABI communicator over
    # whatever $FDB_RCCL_LIB names. A functional check; the line says so.
    args.share_device = os.environ.get("FDB_BENCH_TEST_SHARE_DEVICE") == "1" and torch.cuda.device_count() < world
    if args.share_device:
        local_rank = local_rank % torch.cuda.device_count()
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this box ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    merging = world > 1 or args.force_merge
    if merging:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ctl_dev = "cpu" if args.share_device else "cuda"

    from frostdb_amd import build as fb
    if rank == 0:
        fb.build()
    if world > 1:
        dist.barrier()

    comm = None
    if merging and not args.torch_merge:
        # RCCL through the C ABI: rank 0's unique id travels over the already-initialised process group (as a Go host would
        # ship it over its own control plane), then every rank joins the communicator with fdb_comm_init_rank.
        from frostdb_amd import comm as fcomm
        uid = torch.zeros(fcomm.UNIQUE_ID_BYTES, dtype=torch.uint8, device=ctl_dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(fcomm.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(uid, src=0)
        # If joining the C-ABI communicator fails on this node, every rank agrees (over the process group) to merge through
        # torch.distributed instead — the line then says so — rather than losing the scaling measurement. (A failure that hits
        # only some ranks in the middle of a collective cannot be recovered from; this covers the symmetric ones: a missing
        # symbol, a refused init. Failures BEFORE a merge's first collective are voted inside the library, fdb_comm.cpp.)
        ok = 1
        try:
            comm = fcomm.Comm(bytes(uid.cpu().numpy().tobytes()), world, rank, local_rank)
        except Exception as e:  # noqa: BLE001
            ok, comm = 0, None
            print(f"[bench] rank {rank}: fdb_comm_init_rank failed ({e}); proposing the torch.distributed merge", file=sys.stderr)
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            if comm is not None:
                comm.close()
            comm = None
            args.torch_merge = True
            args.merge_fallback = True

    line = run_rank(args, ProcGroup(rank, world, local_rank, dist), local_rank, comm)
    if rank == 0:
        emit(json.dumps(line))
    if comm is not None:
        comm.close()
    if merging:
        dist.destroy_process_group()


def compare_with_oracle(wl, got, want):
    """`got`: the GPU path's record, `want`: the oracle's, for the SAME input rows — group by group: row counts and int64 MIN / MAX
    bit-exact, float64 sums within 1e-9 relative (BASELINE.json's contract). Returns the number of groups compared."""
    import numpy as np
    from frostdb_amd import synth
    if wl.config == 5:
        gi, wi = synth.cfg5_decode_group_ids(got), synth.cfg5_decode_group_ids(want)
        go, wo = np.argsort(gi, kind="stable"), np.argsort(wi, kind="stable")
        assert len(gi) == len(wi) and np.array_equal(gi[go], wi[wo]), "oracle parity: the group sets differ"
        gs = got.column(got.num_columns - 1).to_numpy()[go]   # ("sum(value)"; an ordered partial stage names it "value")
        ws = want.column(want.num_columns - 1).to_numpy()[wo]
        assert np.allclose(gs, ws, rtol=1e-9, atol=0.0), "oracle parity: a group's sum differs by more than 1e-9 relative"
        return int(len(gi))

    def rows(rec):
        key = rec.column(rec.schema.names.index("labels.path"))
        key = key.dictionary_decode() if hasattr(key, "dictionary_decode") else key
        cols = [rec.column(rec.schema.names.index(n)).to_pylist() for n in rec.schema.names if n != "labels.path"]
        names = [n for n in rec.schema.names if n != "labels.path"]
        return {k: dict(zip(names, v)) for k, v in zip([x if not isinstance(x, str) else x.encode() for x in key.to_pylist()], zip(*cols))}
    if wl.ordered and got.schema.names != want.schema.names and got.num_columns == want.num_columns:
        got = got.rename_columns(want.schema.names)  # (a partial-stage OrderedAggregate names its result after the column, ordered_aggregate.go:551-557)
    g, w = rows(got), rows(want)
    assert g.keys() == w.keys(), "oracle parity: the group sets differ"
    for k, wv in w.items():
        for n, x in wv.items():
            y = g[k][n]
            if isinstance(x, float):
                assert math.isclose(y, x, rel_tol=1e-9), (k, n, y, x)
            else:
                assert y == x, (k, n, y, x)
    return len(w)


def measure_cfg5_merge(args, wl, steps, warmup, ceiling):
    """Two chains of one GPU at cfg 5 (≙ Synchronizer + final-stage HashAggregate, synchronize.go:31-53, aggregate.go:340-348): each
    plan scans half of the resident records into its own hash table (≈ 5 – 10 M groups each, nearly all shared), fdb_plan_merge folds the
    second table into the first on the device (fdb_merge.hip: hash_merge_wave_kernel over the source's slots), Finish emits the record.
    Reports the merge alone (host clock around fdb_plan_merge, which returns with both streams idle) next to the whole step."""
    import numpy as np
    from frostdb_amd import physicalplan as pp
    half = len(wl.resident) // 2
    parts = (wl.resident[:half], wl.resident[half:])
    rows = sum(r.num_rows for r in wl.resident)

    def step():
        a = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc)
        b = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc)
        try:
            a.CallbackResident(parts[0]); b.CallbackResident(parts[1])
            na, nb = a.num_groups(), b.num_groups()  # (waits for both scans)
            t0 = time.perf_counter()
            a.Merge(b)
            t1 = time.perf_counter()
            kernel = a.last_kernel()
            out = a.Finish()
        finally:
            a.Close(); b.Close()
        return out, (t1 - t0) * 1e3, na, nb, kernel

    out, _, na, nb, kernel = step()
    expected = [np.asarray(e) for e in wl.expected]
    checked = wl.check(out, expected, rows)
    checked["against"] = "numpy expectation (row count and Σ value of every generated record) + group count bound; not the oracle"
    if not args.no_oracle_parity and wl.sample is not None:
        checked["oracle"] = oracle_parity_merged(wl, 1 << 18)
    n_out, n_key_cols = out.num_rows, out.num_columns - len(wl.aggs)
    del out
    for _ in range(warmup):
        step()
    merge_ms, step_s = [], []
    for _ in range(steps):
        t = time.perf_counter()
        merge_ms.append(step()[1])
        step_s.append(time.perf_counter() - t)
    # (the MEDIAN step: a step builds two 11 GB tables, and one step in a dozen meets a pool that has to grow — 60 ms once, then never)
    dt = float(np.median(step_s)) * steps
    # what one merge moves: every source group's entry and key tuple read once; a new group's tuple and entry written, a known group's entry folded
    kw_bytes, ew_bytes = 4 * ((2 + n_key_cols + 3) // 4 * 4), 32
    moved = nb * (kw_bytes + ew_bytes) + (n_out - na) * (kw_bytes + ew_bytes) + (nb - (n_out - na)) * ew_bytes
    ms = float(np.median(merge_ms))
    return {"workload": f"cfg5_merge: two plans of {rows // 2} rows each ({na} + {nb} groups) -> fdb_plan_merge -> Finish ({n_out} groups)",
            "merge_ms": ms, "merge_ms_all": [round(x, 3) for x in merge_ms], "merge_kernel": kernel, "groups_per_s": nb / (ms * 1e-3),
            "merge_bytes": moved, "merge_GBps": moved / (ms * 1e-3) / 1e9, "merge_frac_of_read_ceiling": moved / (ms * 1e-3) / 1e9 / ceiling if ceiling else None,
            "ms_per_step": dt / steps * 1e3, "ms_per_step_all": [round(x * 1e3, 2) for x in step_s], "steps": steps, "warmup": warmup, "value": rows * steps / dt, "unit": "rows/s", "checked": checked}


def oracle_parity_merged(wl, max_rows):
    """Two plans over two slices of the sample record, merged on the device, against the oracle over both slices."""
    from frostdb_amd import physicalplan as pp
    import oracle  # noqa: F401
    from oracle import OracleBatch, OraclePlan
    n = min(max_rows, wl.sample.num_rows // 2)
    s0, s1 = wl.sample.slice(0, n), wl.sample.slice(wl.sample.num_rows - n, n)
    r0, r1 = pp.ResidentBatch(s0, device=wl.device), pp.ResidentBatch(s1, device=wl.device)
    a = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc)
    b = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc)
    try:
        a.CallbackResident([r0]); b.CallbackResident([r1])
        a.Merge(b)
        got = a.Finish()
    finally:
        a.Close(); b.Close(); r0.close(); r1.close()
    threads, bs = os.cpu_count() or 1, 65536
    batches = [OracleBatch.from_arrow(s.slice(o, min(bs, n - o))) for s in (s0, s1) for o in range(0, n, bs)]
    oplan = OraclePlan(wl.filt, wl.aggs, wl.groups, nchains=threads)
    res = oplan.execute(batches, threads)
    want = res.to_arrow()
    res.close(); oplan.close()
    for x in batches:
        x.close()
    n_groups = compare_with_oracle(wl, got, want)
    return {"rows": 2 * n, "groups": n_groups, "what": "oracle.OraclePlan.execute over the first and the last %d rows of the first resident record vs. two GPU plans "
            "(one per slice) merged with fdb_plan_merge: group sets equal, counts bit-exact, float64 sums within 1e-9 relative" % n}


def oracle_parity(wl, max_rows=None):
    """oracle/ (the reference's algorithm restated on the CPU — the checker) and the timed GPU path on the SAME rows: the first
    resident record of the workload (`max_rows`: its first rows only — cfg 5's final stage walks the whole group map per group
    column like aggregate.go:586-590 and takes ≈ 20 µs per group). Returns what goes into the line's `checked.oracle`."""
    from frostdb_amd import physicalplan as pp
    import oracle  # noqa: F401
    from oracle import OracleBatch, OraclePlan
    sample = wl.sample if max_rows is None or max_rows >= wl.sample.num_rows else wl.sample.slice(0, max_rows)
    rb = wl.resident[0] if sample.num_rows == wl.sample.num_rows else pp.ResidentBatch(sample, device=wl.device)
    try:
        plan = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=wl.device, desc=wl.desc)
        plan.CallbackResident([rb])
        got = plan.Finish()
        plan.Close()
    finally:
        if rb is not wl.resident[0]:
            rb.close()
    threads, bs, n = os.cpu_count() or 1, 65536, sample.num_rows
    batches = [OracleBatch.from_arrow(sample.slice(o, min(bs, n - o))) for o in range(0, n, bs)]
    oplan = OraclePlan(wl.filt, wl.aggs, wl.groups, nchains=threads)
    t = time.perf_counter()
    res = oplan.execute(batches, threads)
    dt = time.perf_counter() - t
    want = res.to_arrow()
    res.close(); oplan.close()
    for b in batches:
        b.close()
    n_groups = compare_with_oracle(wl, got, want)
    return {"rows": n, "groups": n_groups, "oracle_s": round(dt, 3),
            "what": "oracle.OraclePlan.execute over the %s of the first resident record vs. the GPU path on the same rows: group sets equal, "
                    "counts / int64 MIN / MAX bit-exact, float64 sums within 1e-9 relative" % ("whole" if n == wl.sample.num_rows else "first %d rows" % n)}


def cpu_quota_cpus():
    """The container's CPU bandwidth limit in CPUs (cgroup v2 cpu.max / v1 cfs quota), or None: what the host-side numbers of this line
    (cpu_baseline, host_records_chains, the widening threads of a big Finish) were really given — the bench boxes report 256 CPUs and
    grant 16."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_throttled():
    """(periods in which the container ran into its CPU quota, µs it was held back) so far — cgroup v2 cpu.stat; (None, None) elsewhere."""
    try:
        kv = dict(l.split()[:2] for l in open("/sys/fs/cgroup/cpu.stat") if l.strip())
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except (OSError, ValueError):
        return None, None


def cpu_baseline(sample, filt, aggs, groups, target_seconds):
    """Times oracle.OraclePlan.execute (T chains → Synchronizer → final stage, the reference's algorithm restated
    in C++) on a bounded sample of the same workload; the Go reference itself cannot run here (no Go toolchain)."""
    import oracle  # noqa: F401
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
    return {"value": rate, "unit": "rows/s", "cores": threads, "cpu_quota_cpus": cpu_quota_cpus(), "kind": "port",
            "sample": f"{nrows} rows ({passes} passes over the first resident record) in {bs}-row records, {threads} chains, {dt:.2f} s wall"}


if __name__ == "__main__":
    main()
