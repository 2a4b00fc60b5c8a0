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
    assert set(out) == {"value", "unit", "cores", "kind", "sample"}
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
