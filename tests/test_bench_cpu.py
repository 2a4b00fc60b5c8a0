"""bench.py's pieces that need no GPU: the queries it times are BASELINE.json's, and the cpu_baseline leg (the oracle timed on a
bounded sample) reports what the contract asks for."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_queries_are_the_configs_of_baseline_json():
    b = _bench()
    filt, aggs, groups, desc = b.query(2)
    assert str(filt) == "labels.code == 200" and [a.Name() for a in aggs] == ["sum(value)"] and [g.name for g in groups] == ["labels.path"]
    filt, aggs, groups, desc = b.query(3)
    assert sorted(a.Name() for a in aggs) == ["count(value)", "max(timestamp)", "min(timestamp)", "sum(value)"]
    assert "OR" in str(filt) and "AND" in str(filt) and "labels.instance" in str(filt)
    filt, aggs, groups, desc = b.query(5)
    assert filt is None and [g.name for g in groups] == ["labels"] and groups[0].dynamic
    assert b.HBM_PEAK_GBS == 8000.0


def test_cpu_baseline_leg_reports_the_contract_fields():
    from frostdb_amd import synth
    b = _bench()
    filt, aggs, groups, _ = b.query(2)
    sample = synth.prometheus_chunk(0, 0, 200_000)
    out = b.cpu_baseline(sample, filt, aggs, groups, target_seconds=0.2)
    assert set(out) == {"value", "unit", "cores", "cpu_quota_cpus", "kind", "sample"}
    assert out["unit"] == "rows/s" and out["kind"] == "port" and out["cores"] == (os.cpu_count() or 1)
    assert out["value"] > 1e5 and "rows" in out["sample"]


def test_expected_cfg2_is_an_independent_numpy_answer():
    from frostdb_amd import synth
    b = _bench()
    rec = synth.prometheus_chunk(0, 1, 50_000)
    s, c = b.expected_cfg2(rec)
    code = rec.column("labels.code").dictionary_decode().to_pylist()
    path = rec.column("labels.path").dictionary_decode().to_pylist()
    val = rec.column("value").to_numpy()
    want = {}
    for i in range(rec.num_rows):
        if code[i] == b"200":
            want[path[i]] = want.get(path[i], 0.0) + val[i]
    names = synth.PATHS + [None]
    got = {names[i]: s[i] for i in range(len(names)) if c[i] > 0}
    assert set(got) == set(want)
    for k in want:
        assert np.isclose(got[k], want[k], rtol=1e-9)


def test_stdout_carries_only_the_json_line():
    """Libraries print to file descriptor 1 behind Python's back (RCCL's version banner when a communicator is created): after
    claim_stdout() all of that lands on stderr and emit() alone reaches the real stdout."""
    import subprocess
    import sys
    code = ("import importlib.util, os, sys\n"
            "spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "b.claim_stdout()\n"
            "os.write(1, b'RCCL version : banner\\n')\n"
            "print('python noise')\n"
            "b.emit('{\"ok\": 1}')\n"
            "os.write(1, b'late noise\\n')\n") % os.path.join(ROOT, "bench.py")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stdout == '{"ok": 1}\n'
    assert "RCCL version : banner" in p.stderr and "python noise" in p.stderr and "late noise" in p.stderr


def test_shard_rule_of_cfg4_one_billion_rows_over_n_gpus():
    """BASELINE.json configs[3]: 1 B rows sharded across the GPUs — 125 M per GPU at N = 8 (SURVEY §8d cfg 4). The rule bench.py
    uses at N > 1 (strong scaling): equal shares, remainders to the first ranks, Σ = total for every N."""
    b = _bench()
    assert [b.shard_rows(1_000_000_000, 8, r) for r in range(8)] == [125_000_000] * 8
    assert [b.shard_rows(1_000_000_000, 1, 0)] == [1_000_000_000]
    for world in (1, 2, 3, 4, 5, 6, 7, 8, 16):
        shares = [b.shard_rows(1_000_000_000, world, r) for r in range(world)]
        assert sum(shares) == 1_000_000_000 and max(shares) - min(shares) <= 1 and shares == sorted(shares, reverse=True)
    assert [b.shard_rows(10, 3, r) for r in range(3)] == [4, 3, 3]
    import pytest
    with pytest.raises(ValueError):
        b.shard_rows(10, 2, 2)
    # the flags: N > 1 defaults to the sharded (strong) form; --weak keeps --rows on every GPU; --force-local implies one process
    a = b.parse_args(["--gpus", "8"])
    assert a.gpus == 8 and not a.weak and not a.one_process
    assert b.parse_args(["--gpus", "2", "--force-local"]).force_local
    assert b.HEADLINE_ROWS == 1_000_000_000


def _group_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = _bench()
        g = b.ProcGroup(rank, world, 0, dist)
        g.barrier()
        exp = [np.arange(5, dtype=np.float64) * (rank + 1), np.array([rank + 1, 10 - rank], dtype=np.int64)]
        summed = [g.reduce(exp[0], "sum"), g.reduce(exp[1], "sum")]
        mn, mx = g.reduce(exp[1], "min"), g.reduce(exp[1], "max")
        shards = g.gather(b.shard_rows(1_000_000_007, world, rank))
        slowest = float(g.reduce(np.array([0.5 + rank]), "max")[0])
        q.put((rank, [a.tolist() for a in summed], mn.tolist(), mx.tolist(), shards, slowest))
    finally:
        dist.destroy_process_group()


def test_control_plane_of_the_n_rank_bench_on_gloo_world_size_2():
    """bench.py's host-side control plane at N > 1 (barrier, max-over-ranks of the elapsed time, sums / mins / maxes of the
    numpy expectations, the shard sizes) over torch.distributed — gloo here, RCCL on the GPU box; the thread form used by
    --one-process gives the same answers."""
    import socket
    import threading
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, summed, mn, mx, shards, slowest in got:
        assert summed == [[0.0, 3.0, 6.0, 9.0, 12.0], [3, 19]] and mn == [1, 9] and mx == [2, 10]
        assert shards == [500_000_004, 500_000_003] and slowest == 1.5
    # the same through the thread form
    b = _bench()
    shared = b.ThreadShared(2)
    out = [None, None]

    def work(r):
        g = b.ThreadGroup(r, 2, 0, shared)
        g.barrier()
        out[r] = (g.reduce(np.array([r + 1, 10 - r], dtype=np.int64), "sum").tolist(), g.reduce(np.array([0.5 + r]), "max").tolist(),
                  g.gather(b.shard_rows(1_000_000_007, 2, r)))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert out[0] == out[1] == ([3, 19], [1.5], [500_000_004, 500_000_003])


def test_self_launch_of_gpus_8_is_the_command_torch_distributed_run_expects(monkeypatch):
    """`python bench.py --gpus 8 --steps K --warmup W` started plainly re-executes itself as the driver's own form: `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 …`. Checked
    against torch's own argument parser (torch.distributed.run.get_args_parser), without a GPU: the device count is faked, execv
    captured."""
    import subprocess
    import sys
    import types
    b = _bench()
    captured = {}
    monkeypatch.setattr(b.subprocess if hasattr(b, "subprocess") else subprocess, "run",
                        lambda *a, **k: types.SimpleNamespace(returncode=0, stdout="8\n", stderr=""))

    def fake_execv(exe, argv):
        captured["exe"], captured["argv"] = exe, list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(b.os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    args = b.parse_args()
    try:
        b.self_launch(args)
    except SystemExit:
        pass
    argv = captured["argv"]
    assert captured["exe"] == sys.executable and argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    from torch.distributed.run import get_args_parser
    ns = get_args_parser().parse_args(argv[3:])
    assert ns.nnodes == "1" and str(ns.nproc_per_node) == "8" and ns.master_addr == "127.0.0.1" and int(ns.master_port) > 0
    assert os.path.basename(ns.training_script) == "bench.py" and os.path.isabs(ns.training_script)
    assert ns.training_script_args == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    # fewer devices than ranks: refused (no N-GPU number from fewer GPUs) …
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout="1\n", stderr=""))
    import pytest
    with pytest.raises(SystemExit) as e:
        b.self_launch(args)
    assert "refusing" in str(e.value)
