#!/usr/bin/env python
"""bench.py — rows/sec of the fused filter + group-by aggregate over synthetic Prometheus Arrow data.

Contract: ``python bench.py --gpus N --steps K --warmup W``. N > 1 means one rank per GPU: under
``python -m torch.distributed.run`` the ranks read RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env; started plainly
(``python bench.py --gpus 8``) the script launches those N ranks itself and refuses loudly if the box has fewer GPUs.

Headline workload (BASELINE.json ``metric``): filter + group-by over **1 B Prometheus rows** resident in the HBM of
one GPU (cfg 2's query ``labels.code=='200' + SUM(value) GROUP BY labels.path``; 16.25 algorithmic B/row). At N > 1
every GPU holds its own 1 B-row shard (weak scaling; parts shard with no data-path collective) and the per-GPU partial
tables are merged with RCCL through the C ABI (``fdb_comm_*``). A *step* is one full pass of the hot path over this
rank's shard: create the operator chain (fdb_plan_create), scan every resident record with the fused HIP kernel
(fdb_plan_push_batches), merge (N > 1) and produce the final Arrow record (fdb_plan_finish). Inputs are resident in HBM
before the timed region; the result of the timed path is checked against a numpy restatement of the query.

Rank 0 prints ONE JSON line: the metric, the HBM roofline of the scan kernel (hipEvent-timed on the plan's own stream
over the timed steps), the CPU baseline (the oracle's restatement of the reference's algorithm on this box's host
cores, bounded sample) and — at N = 1 with the default workload — ``other_configs``: the same measurement for
BASELINE.json's cfg 3 (multi-predicate, 100 M rows) and cfg 5 (32 label columns, 10 M groups, 100 M rows).
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HEADLINE_ROWS = 1_000_000_000
PROFILE_ROUND = "round2"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 5],
                    help="0 (default): the headline — cfg 2's query over 1 B rows/GPU, plus cfg 3 / cfg 5 lines at N=1; 2 / 3 / 5: that BASELINE.json config alone (100 M rows)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: 1 B for the headline, 100 M for --config 2/3/5)")
    ap.add_argument("--batch-rows", type=int, default=25_000_000, help="rows per resident record (part)")
    ap.add_argument("--groups", type=int, default=10_000_000, help="cfg 5: distinct groups")
    ap.add_argument("--cfg5-sorted", action="store_true", help="cfg 5: every record's rows ordered by group (a scan of a table sorted by its label columns)")
    ap.add_argument("--rows-per-thread", type=int, default=0, help="0: slot kernel (default); 4/8: sequential kernel")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--host-records", action="store_true",
                    help="secondary measurement: push HOST Arrow records (fdb_plan_push: PCIe copy + scan per record) instead of HBM-resident parts; never the headline value")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant: 0 default (run-time specialised), 2: 256 threads, 3: 1024 threads, 4: interpreting kernel only")
    ap.add_argument("--per-record-launch", action="store_true", help="one kernel launch per resident record instead of one per scan")
    ap.add_argument("--force-merge", action="store_true", help="run the RCCL merge path even with one rank (functional check on a 1-GPU box)")
    ap.add_argument("--torch-merge", action="store_true", help="merge through torch.distributed (frostdb_amd.distributed) instead of the C ABI's fdb_comm_*")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg 3 / cfg 5 lines of the default run")
    ap.add_argument("--cpu-sample-seconds", type=float, default=3.0)
    ap.add_argument("--sweep", action="store_true", help="kernel geometry sweep first (tuning aid; table on stderr)")
    return ap.parse_args(argv)


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
    value = batch.column(batch.schema.get_field_index("value")).to_numpy()
    return batch.num_rows, float(value.sum())


def self_launch(args):
    """`python bench.py --gpus N` started plainly: become N ranks (one per GPU) under torch.distributed.run."""
    import socket
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    n_dev = int(out.stdout.strip() or 0) if out.returncode == 0 else 0
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {n_dev} GPU(s); refusing to report a {args.gpus}-GPU number from fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


class Workload:
    """One configuration made resident on this rank's GPU + the numpy expectation of its query."""

    def __init__(self, args, config, rows, rank, local_rank):
        from frostdb_amd import physicalplan as pp
        from frostdb_amd import synth
        from frostdb_amd.logicalplan import to_desc
        self.args, self.config, self.rows, self.rank, self.local_rank = args, config, rows, rank, local_rank
        self.filt, self.aggs, self.groups, self.qdesc = query(config)
        self.desc = to_desc(self.filt, self.aggs, self.groups)  # planned once; every step instantiates a fresh operator chain
        t0 = time.time()
        br = args.batch_rows
        n_chunks = (rows + br - 1) // br
        sizes = [min(br, rows - i * br) for i in range(n_chunks)]
        self.n_chunks = n_chunks

        def gen(i):
            if config == 5:
                b = synth.cfg5_chunk(rank, i, sizes[i], n_groups=args.groups, sorted_rows=args.cfg5_sorted)
                return b, expected_cfg5(b)
            b = synth.prometheus_chunk(rank, i, sizes[i], row_base=i * br, cfg3=(config == 3))
            return b, (expected_cfg2(b) if config == 2 else expected_cfg3(b))

        if config == 5:
            synth.cfg5_chunk(rank, 0, 8, n_groups=args.groups)  # builds the per-group digit tables once, before the thread pool
        self.resident, self.host_batches, self.sample = [], [], None
        self.expected = None
        workers = max(1, min(16, n_chunks, (os.cpu_count() or 8) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
        with ThreadPoolExecutor(max_workers=workers) as ex:
            for i, (b, e) in enumerate(ex.map(gen, range(n_chunks))):
                self._fold(e)
                if args.host_records:
                    self.host_batches.append(b)
                else:
                    self.resident.append(pp.ResidentBatch(b, device=local_rank))
                if i == 0:
                    self.sample = b
        self.t_gen = time.time() - t0
        self.hbm_bytes = sum(r.device_bytes for r in self.resident)

    def _fold(self, e):
        import numpy as np
        if self.expected is None:
            self.expected = list(e)
        elif self.config == 2 or self.config == 5:
            self.expected = [a + b for a, b in zip(self.expected, e)]
        else:
            c, mn, mx, s = self.expected
            self.expected = [c + e[0], np.minimum(mn, e[1]), np.maximum(mx, e[2]), s + e[3]]

    def release(self):
        for r in self.resident:
            r.close()
        self.resident, self.host_batches = [], []

    # ---- the result of the timed path against the numpy restatement (single rank's shard) -------------------------------
    def check(self, out):
        from frostdb_amd import synth
        names = out.schema.names
        col = lambda n: out.column(names.index(n))  # noqa: E731
        if self.config == 5:
            n_rows, total = self.expected
            n_out = out.num_rows
            s = col("sum(value)").to_numpy()
            assert n_out <= self.args.groups and (self.rows < 5 * self.args.groups or n_out > 0.99 * self.args.groups), n_out
            assert math.isclose(float(s.sum()), total, rel_tol=1e-9), (float(s.sum()), total)
            return {"groups_out": n_out, "sum_check": "Σ sum(value) == Σ value (1e-9 rel)"}
        paths = synth.PATHS + [None]
        key = col("labels.path")
        key = key.dictionary_decode() if hasattr(key, "dictionary_decode") else key
        got_keys = key.to_pylist()
        if self.config == 2:
            exp_sum, exp_cnt = self.expected
            got = dict(zip(got_keys, col("sum(value)").to_pylist()))
            for i, p in enumerate(paths):
                if exp_cnt[i] == 0:
                    assert p not in got, p
                else:
                    assert math.isclose(got[p], exp_sum[i], rel_tol=1e-9), (p, got[p], exp_sum[i])
            assert len(got) == int((exp_cnt > 0).sum())
            return {"groups_out": len(got)}
        cnt, mn, mx, s = self.expected
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
        return {"groups_out": len(rows)}


def run_workload(args, wl, steps, warmup, world, comm, dist):
    """Correctness step, warm-up, then the timed region (barrier + device sync on both sides, max over ranks)."""
    import torch
    from frostdb_amd import physicalplan as pp
    rank, local_rank = wl.rank, wl.local_rank
    merging = world > 1 or args.force_merge

    def step(timing=False, tuning=None):
        plan = pp.HashAggregatePlan(wl.filt, wl.aggs, wl.groups, device=local_rank, desc=wl.desc)
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
            for rb in wl.resident:
                plan.Callback(rb)
        else:
            plan.CallbackResident(wl.resident)
        out = None
        if merging and args.torch_merge:
            from frostdb_amd.distributed import layout_probe, merge_plan, merge_plan_alltoall
            if wl.config == 5:  # high cardinality: hash-partitioned all-to-all; every rank finishes its own shard of the groups
                shard = merge_plan_alltoall(plan, device=torch.device("cuda", local_rank))
                try:
                    out = shard.Finish()
                finally:
                    shard.Close()
            else:
                probe = layout_probe(plan, torch.device("cuda", local_rank))  # overlaps with the scan kernel
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
        else:
            out = plan.Finish()
        st = plan.stats() if timing else None
        if st is not None:
            st["kernel"] = plan.last_kernel()
        plan.Close()
        return out, st

    out, _ = step()
    checked = None
    if world == 1 and not args.host_records:
        checked = wl.check(out)
    elif wl.config == 5:
        t = torch.tensor([out.num_rows], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)  # shards of the all-to-all merge are disjoint
        n_out = int(t.item())
        assert n_out <= args.groups and n_out > 0, n_out
        checked = {"groups_out": n_out}
    del out

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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_ms, k_bytes, k_launches, kernel_name = 0.0, 0, 0, ""
    for _ in range(steps):
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
    return {"elapsed": elapsed, "k_ms": k_ms, "k_bytes": k_bytes, "k_launches": k_launches, "kernel": kernel_name, "checked": checked}


def traffic_for(tag, rows, kernel_name):
    """HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside the timed process)."""
    for rnd in (PROFILE_ROUND, "round1"):
        tpath = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("rows") == rows and tj.get("kernel") == kernel_name:
                return tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"], os.path.relpath(tpath, ROOT)
    return None, None


def traffic_is_per_average_launch(tag):
    """cfg 5's scan is cut into several launches: its traffic file holds the mean over all of them, like `avg_launch_ms`."""
    tpath = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{tag}_traffic.json")
    if not os.path.exists(tpath):
        return False
    with open(tpath) as fh:
        return bool(json.load(fh).get("per_average_launch"))


def roofline_of(r, rows, steps, tag, ceiling=None):
    achieved = r["k_bytes"] / (r["k_ms"] * 1e-3) / 1e9 if r["k_ms"] > 0 else 0.0
    launches = max(r["k_launches"], 1)
    traffic, src = traffic_for(tag, rows, r["kernel"])
    # (a scan cut into several launches — the hash path's ≤ 4 M-row chunks — reports per-launch figures of the average launch)
    if traffic is not None and launches != steps and not traffic_is_per_average_launch(tag):
        traffic, src = None, None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": r["kernel"], "avg_launch_ms": r["k_ms"] / launches,
            "launches_per_step": launches / max(steps, 1),
            "algorithmic_bytes_per_launch": r["k_bytes"] / launches, "bytes_per_row": r["k_bytes"] / max(rows * steps, 1),
            "whole_step_frac": (r["k_bytes"] / max(steps, 1)) / (r["elapsed"] / max(steps, 1)) / 1e9 / HBM_PEAK_GBS,
            "measured_read_ceiling": ceiling, "frac_of_measured_ceiling": achieved / ceiling if ceiling else None}


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


def main():
    args = parse_args()
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
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this box ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    merging = world > 1 or args.force_merge
    if merging:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from frostdb_amd import build as fb
    if rank == 0:
        fb.build()
    if world > 1:
        dist.barrier()
    from frostdb_amd import physicalplan as pp

    comm = None
    if merging and not args.torch_merge:
        # RCCL through the C ABI: rank 0's unique id travels over the already-initialised process group (as a Go host would
        # ship it over its own control plane), then every rank joins the communicator with fdb_comm_init_rank.
        from frostdb_amd import comm as fcomm
        uid = torch.zeros(fcomm.UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(fcomm.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(uid, src=0)
        # The C-ABI communicator has only ever run with one rank per device on the boxes this was developed on: if joining it
        # fails on this node, every rank agrees (over the process group) to merge through torch.distributed instead — the
        # line then says so — rather than losing the scaling measurement. (A failure that hits only some ranks in the
        # middle of a collective cannot be recovered from; this covers the symmetric ones: a missing symbol, a refused init.)
        ok = 1
        try:
            comm = fcomm.Comm(bytes(uid.cpu().numpy().tobytes()), world, rank, local_rank)
        except Exception as e:  # noqa: BLE001
            ok, comm = 0, None
            print(f"[bench] rank {rank}: fdb_comm_init_rank failed ({e}); proposing the torch.distributed merge", file=sys.stderr)
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            if comm is not None:
                comm.close()
            comm = None
            args.torch_merge = True
            args.merge_fallback = True

    headline = args.config == 0
    config = 2 if headline else args.config
    rows = args.rows or (HEADLINE_ROWS if headline else 100_000_000)

    wl = Workload(args, config, rows, rank, local_rank)
    r = run_workload(args, wl, args.steps, args.warmup, world, comm, dist)
    total_rows = rows * world * args.steps
    value = total_rows / r["elapsed"]

    # ---- CPU baseline (rank 0, N = 1 only): the oracle's restatement on this box's host cores, bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl.sample, wl.filt, wl.aggs, wl.groups, args.cpu_sample_seconds)
    ceiling = None
    if rank == 0 and world == 1:
        ceiling = pp.read_ceiling(local_rank, 2 << 30, 5)  # plain read kernel on this box, this run

    tag = ("cfg1B" if rows == HEADLINE_ROWS and config == 2 else f"cfg{config}")
    name = "headline" if headline else f"cfg{config}"
    if world > 1 and config == 2:
        name = "cfg4-style"
    line = {
        "metric": "rows/sec filter+group-by on 1B-row Prometheus Arrow (HBM-resident); achieved HBM GB/s vs peak" if not args.host_records
                  else "rows/sec filter+group-by on Prometheus Arrow (HOST records, PCIe-inclusive; secondary)",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{name}: Prometheus schema, {rows} rows/GPU × {world} GPU, {wl.qdesc}",
                   "rows_per_gpu": rows, "records_per_gpu": wl.n_chunks, "groups": args.groups if config == 5 else 1025,
                   "parallelism": (f"parts sharded over {world} GPU(s), no data-path collective; " +
                                   ("RCCL all-to-all of hash-partitioned partial tables, result sharded" if config == 5
                                    else "RCCL all-reduce of the partial tables (C ABI fdb_comm_*)" if not args.torch_merge
                                    else "RCCL all-reduce of the partial tables (torch.distributed" +
                                    ("; fallback: the C-ABI communicator could not be joined)" if getattr(args, "merge_fallback", False) else ")"))) if merging else "1 GPU"},
        "roofline": roofline_of(r, rows, args.steps, tag, ceiling),
        "cpu_baseline": cpu,
        "checked": r["checked"],
        "setup": {"gen_and_upload_s": wl.t_gen, "hbm_resident_bytes": wl.hbm_bytes},
    }

    # ---- the other single-GPU configurations of BASELINE.json, same run, same measurement (fewer steps) ---------------------
    if headline and world == 1 and not args.no_other_configs and not args.host_records:
        wl.release()
        others = {}
        for cfg, st, wu in ((3, max(5, args.steps // 2), 2), (5, 3, 1)):
            w2 = Workload(args, cfg, 100_000_000, rank, local_rank)
            r2 = run_workload(args, w2, st, wu, world, comm, dist)
            others[f"cfg{cfg}"] = {
                "workload": f"cfg{cfg}: Prometheus schema, 100000000 rows, {w2.qdesc}",
                "value": 100_000_000 * st / r2["elapsed"], "unit": "rows/s", "steps": st, "warmup": wu,
                "ms_per_step": r2["elapsed"] / st * 1e3, "roofline": roofline_of(r2, 100_000_000, st, f"cfg{cfg}"),
                "checked": r2["checked"], "setup": {"gen_and_upload_s": w2.t_gen, "hbm_resident_bytes": w2.hbm_bytes},
            }
            w2.release()
        line["other_configs"] = others

    if rank == 0:
        emit(json.dumps(line))
    if comm is not None:
        comm.close()
    if merging:
        dist.destroy_process_group()


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
    return {"value": rate, "unit": "rows/s", "cores": threads, "kind": "port",
            "sample": f"{nrows} rows ({passes} passes over the first resident record) in {bs}-row records, {threads} chains, {dt:.2f} s wall"}


if __name__ == "__main__":
    main()
